"""Wavelet sets a transform object uses when its constructor is given none.

The names are part of the interface being reproduced (dtcwt/defaults.py:1-3): level 1 uses the
(5, 7)-tap near-symmetric biorthogonal pair, levels >= 2 the 10-tap quarter-sample-shift set --
the pair `BASELINE.json` quotes its metric on and the one the fused gfx950 kernels are tuned
for (dtcwt_amd/csrc/fused2d_table.hpp).  `dtcwt_amd.coeffs.biort` / `qshift` resolve them to
tap vectors.
"""
#: level-1 (biorthogonal) wavelet set
DEFAULT_BIORT = 'near_sym_a'
#: level >= 2 (q-shift) wavelet set
DEFAULT_QSHIFT = 'qshift_a'

__all__ = ['DEFAULT_BIORT', 'DEFAULT_QSHIFT']

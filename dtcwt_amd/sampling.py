"""``dtcwt.sampling`` of the reference (dtcwt/sampling.py), served by the ``hip`` backend.

The reference's module is backend-independent NumPy; here the same functions run on the
device (``dtcwt_amd/hip/sampling.py``) and accept device-resident pyramids directly.
"""
from dtcwt_amd.hip.sampling import (sample, sample_highpass, rescale, rescale_highpass,     # noqa: F401
                                    upsample, upsample_highpass, DTHETA_DX_2D, DTHETA_DY_2D)

__all__ = (
    'sample', 'sample_highpass',
    'rescale', 'rescale_highpass',
    'upsample', 'upsample_highpass',
)

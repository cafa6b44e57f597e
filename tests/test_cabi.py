"""CPU checks of the C-ABI boundary: the library loads and exports every symbol that
include/dtcwt_hip.h declares; the Python binding covers each of them; without a GPU the
product path fails loudly (no silent CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'dtcwt_hip.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dtcwt_hip_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported_and_bound():
    from dtcwt_amd.hip import _lib
    names = _declared()
    assert len(names) >= 30
    handle = _lib.load_library()
    for n in names:
        assert hasattr(handle, n), 'libdtcwt_hip.so does not export %s' % n
        assert n in _lib.SIGNATURES, 'binding lacks %s' % n
    assert sorted(_lib.SIGNATURES) == names
    # header, library and binding agree on the ABI version (a stale .so is refused by load_library)
    hv = int(re.search(r'#define\s+DTCWT_HIP_ABI_VERSION\s+(\d+)', open(HEADER).read()).group(1))
    assert handle.dtcwt_hip_abi_version() == hv == _lib.ABI_VERSION


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """A library whose ABI version differs from the binding's is reported as 'rebuild', not as an AttributeError
    on the first missing entry point."""
    from dtcwt_amd.hip import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, '_lib_error', None)
    monkeypatch.setattr(_lib, 'ABI_VERSION', _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.NoHIPPresentError, match='rebuild'):
        _lib.load_library()


def test_view_struct_layout():
    from dtcwt_amd.hip._lib import View
    assert ctypes.sizeof(View) == 9 * 8 + 4 * 4


def test_no_cpu_fallback_without_gpu():
    from dtcwt_amd.hip import _lib
    import dtcwt_amd
    if _lib.have_hip():
        pytest.skip('a GPU is present')
    with pytest.raises(_lib.NoHIPPresentError):
        dtcwt_amd.hip.Transform2d().forward(np.zeros((8, 8), np.float32))
    with pytest.raises(RuntimeError):
        dtcwt_amd.hip.Transform1d().forward(np.zeros(8))
    with pytest.raises(RuntimeError):
        dtcwt_amd.hip.Transform3d().forward(np.zeros((4, 4, 4)))


def test_product_does_not_import_oracle():
    """Nothing under dtcwt_amd/ may import or execute the oracle."""
    pkg = os.path.join(ROOT, 'dtcwt_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f
                assert 'dtcwt_oracle' not in src, f

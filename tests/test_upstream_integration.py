"""dtcwt.push_backend('hip') on the REAL upstream package (build container only: needs
/root/reference; skipped on the GPU box, where the reference does not exist)."""
import os
import sys

import numpy as np
import pytest

REF = os.environ.get('DTCWT_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'dtcwt')),
                                reason='reference checkout not present')


@pytest.fixture()
def upstream():
    added = []
    # the reference predates NumPy 2: re-create the three aliases it uses, in this process
    if not hasattr(np, 'int'):
        np.int = int; added.append('int')
    if not hasattr(np, 'asfarray'):
        np.asfarray = lambda a, dtype=None: np.asarray(
            a, dtype=dtype if (dtype is not None and np.issubdtype(np.dtype(dtype), np.inexact)) else np.float64)
        added.append('asfarray')
    if not hasattr(np, 'issubsctype'):
        np.issubsctype = lambda a, b: np.issubdtype(a if isinstance(a, type) else np.dtype(a).type, b)
        added.append('issubsctype')
    sys.path.insert(0, REF)
    old_flag = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        import dtcwt
        yield dtcwt
    finally:
        sys.dont_write_bytecode = old_flag
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == 'dtcwt' or k.startswith('dtcwt.')]:
            del sys.modules[k]
        for a in added:
            delattr(np, a)


def test_push_backend_hip_on_upstream(upstream):
    import dtcwt_amd
    dtcwt = dtcwt_amd.install(upstream)
    assert dtcwt.backend_name == 'numpy'
    with dtcwt.preserve_backend_stack():
        dtcwt.push_backend('hip')
        assert dtcwt.backend_name == 'hip'
        assert dtcwt.Transform2d is dtcwt_amd.hip.Transform2d
        assert dtcwt.Transform1d is dtcwt_amd.hip.Transform1d
        assert dtcwt.Transform3d is dtcwt_amd.hip.Transform3d
        assert dtcwt.Pyramid is dtcwt_amd.hip.Pyramid
        t = dtcwt.Transform2d(biort='near_sym_b', qshift='qshift_b')
        assert len(t.biort) == 4 and len(t.qshift) == 8
        dtcwt.pop_backend()
        assert dtcwt.Transform2d is dtcwt.numpy.Transform2d
    assert dtcwt.backend_name == 'numpy'


def test_numpy_name_resolves_to_upstream(upstream):
    import dtcwt_amd
    with dtcwt_amd.preserve_backend_stack():
        dtcwt_amd.push_backend('numpy')
        assert dtcwt_amd.Transform2d is upstream.numpy.Transform2d
    dtcwt_amd._AVAILABLE_BACKENDS.pop('numpy', None)
    assert dtcwt_amd.backend_name == 'hip'


def test_wavelet_tables_equal_upstream(upstream):
    from dtcwt.coeffs import biort as rb, qshift as rq
    from dtcwt_amd.coeffs import biort, qshift
    for n in ('antonini', 'legall', 'near_sym_a', 'near_sym_b', 'near_sym_b_bp'):
        for a, b in zip(biort(n), rb(n)):
            assert a.shape == b.shape and np.array_equal(a, b)
    for n in ('qshift_06', 'qshift_a', 'qshift_b', 'qshift_c', 'qshift_d', 'qshift_b_bp'):
        for a, b in zip(qshift(n), rq(n)):
            assert a.shape == b.shape and np.array_equal(a, b)

"""Backend dispatch surface (pattern of the reference's tests/test_switchbackends.py:26-93,
with 'hip' as the accelerated backend and a test-registered stand-in for the second one)."""
import pytest

import dtcwt_amd
import dtcwt_amd.hip as hip
from oracle import dtcwt_oracle as o


class _P(object):
    pass


@pytest.fixture(autouse=True)
def _clean():
    with dtcwt_amd.preserve_backend_stack():
        dtcwt_amd.register_backend('oracle-test', {'Transform1d': o.Transform1d, 'Transform2d': o.Transform2d,
                                                   'Transform3d': o.Transform3d, 'Pyramid': o.Pyramid})
        yield
    dtcwt_amd._AVAILABLE_BACKENDS.pop('oracle-test', None)


def test_default_backend():
    assert dtcwt_amd.backend_name == 'hip'
    assert dtcwt_amd.Transform2d is hip.Transform2d
    assert dtcwt_amd.Pyramid is hip.Pyramid
    assert dtcwt_amd.Transform1d is hip.Transform1d
    assert dtcwt_amd.Transform3d is hip.Transform3d


def test_switch_and_pop():
    dtcwt_amd.push_backend('oracle-test')
    assert dtcwt_amd.backend_name == 'oracle-test'
    assert dtcwt_amd.Transform2d is o.Transform2d
    assert dtcwt_amd.Pyramid is o.Pyramid
    dtcwt_amd.push_backend('hip')
    assert dtcwt_amd.Transform2d is hip.Transform2d
    dtcwt_amd.pop_backend()
    assert dtcwt_amd.Transform2d is o.Transform2d
    dtcwt_amd.pop_backend()
    assert dtcwt_amd.backend_name == 'hip'
    assert dtcwt_amd.Transform2d is hip.Transform2d


def test_no_such_backend():
    with pytest.raises(ValueError):
        dtcwt_amd.push_backend('does-not-exist')


def test_cannot_pop_default():
    with pytest.raises(IndexError):
        dtcwt_amd.pop_backend()


def test_backend_with_guard():
    with dtcwt_amd.preserve_backend_stack():
        dtcwt_amd.push_backend('oracle-test')
        assert dtcwt_amd.Transform2d is o.Transform2d
    assert dtcwt_amd.backend_name == 'hip'
    assert dtcwt_amd.Transform2d is hip.Transform2d


def test_backend_guard_restores_on_exception():
    with pytest.raises(RuntimeError):
        with dtcwt_amd.preserve_backend_stack():
            dtcwt_amd.push_backend('oracle-test')
            raise RuntimeError('boom')
    assert dtcwt_amd.backend_name == 'hip'


def test_register_backend_validates_table():
    with pytest.raises(ValueError):
        dtcwt_amd.register_backend('broken', {'Transform2d': o.Transform2d})


def test_install_into_upstream_like_module():
    class Fake(object):
        _AVAILABLE_BACKENDS = {'numpy': {}}
    m = dtcwt_amd.install(Fake)
    assert m._AVAILABLE_BACKENDS['hip']['Transform2d'] is hip.Transform2d


def test_coeffs_interface():
    from dtcwt_amd.coeffs import biort, qshift
    assert [len(x) for x in biort('near_sym_a')] == [5, 7, 7, 5]
    assert len(biort('near_sym_b_bp')) == 6
    assert len(qshift('qshift_b_bp')) == 12
    assert all(x.shape == (10, 1) for x in qshift('qshift_a'))
    with pytest.raises(IOError):
        biort('no_such_wavelet')
    with pytest.raises(ValueError):
        biort('qshift_a')
    with pytest.raises(ValueError):
        qshift('near_sym_a')


def test_unpack_numpy_pyramid_parts():
    """utils.unpack (dtcwt/utils.py:9-42) on a host-constructed pyramid: no device needed."""
    import numpy as np
    from dtcwt_amd.utils import unpack
    from dtcwt_amd.hip import Pyramid
    lo, hi = np.zeros((4, 4)), (np.zeros((2, 2, 6), complex),)
    assert len(list(unpack(Pyramid(lo, hi)))) == 2
    yl, yh, ys = unpack(Pyramid(lo, hi, (lo,)))
    assert yl is not None and len(yh) == 1 and len(ys) == 1
    assert list(unpack(Pyramid(lo, hi), 'hip'))[0] is None      # host arrays: no device handles
    import pytest
    with pytest.raises(ValueError):
        list(unpack(Pyramid(lo, hi), 'tf'))

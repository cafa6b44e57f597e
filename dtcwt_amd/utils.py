"""Host-side helpers on the hot path (dtype coercion, index reflection).

Interface mirrors the reference's dtcwt/utils.py:98-153; written for NumPy >= 2
(the reference uses aliases NumPy removed)."""
import numpy as np


def asfarray(X):
    """float32 and float64 arrays pass through uncopied, everything else becomes float64
    (dtcwt/utils.py:98-105)."""
    X = np.asanyarray(X)
    if X.dtype == np.float32 or X.dtype == np.float64:
        return X
    if np.issubdtype(X.dtype, np.complexfloating):
        return X
    return X.astype(np.float64)


def appropriate_complex_type_for(X):
    """complex64 for float32, complex128 for float64 (dtcwt/utils.py:107-124)."""
    dt = asfarray(X).dtype
    if np.issubdtype(dt, np.complexfloating):
        return dt.type
    return np.complex64 if dt == np.float32 else np.complex128


def as_column_vector(v):
    """(N,) or (1, N) -> (N, 1) (dtcwt/utils.py:126-134)."""
    v = np.atleast_2d(v)
    return v.T if v.shape[0] == 1 else v


def reflect(x, minx, maxx):
    """Reflect values of *x* about *minx* and *maxx* (triangle wave).  With integer x
    and half-integer bounds the end samples repeat: half-sample symmetric extension
    (dtcwt/utils.py:136-153)."""
    x = np.asanyarray(x)
    rng = maxx - minx
    rng2 = 2 * rng
    mod = np.fmod(x - minx, rng2)
    mod = np.where(mod < 0, mod + rng2, mod)
    out = np.where(mod >= rng, rng2 - mod, mod) + minx
    return np.array(out, dtype=x.dtype)


def flat_taps(h):
    """Filter as a flat float64 vector."""
    return np.ascontiguousarray(np.asarray(h, dtype=np.float64).reshape(-1))


def unpack(pyramid, backend='numpy'):
    """Unpack a pyramid into its constituent parts: a generator of ``Yl, Yh[, Yscale]`` (the
    third only if the pyramid was created with ``include_scale``), as dtcwt/utils.py:9-42.
    ``backend='numpy'`` gives NumPy arrays whatever backend produced the pyramid;
    ``backend='hip'`` gives the device-resident buffers (``hip_lowpass`` ...), the analogue of
    the reference's ``'opencl'`` / ``'tf'`` selections."""
    backend = backend.lower()
    if backend == 'numpy':
        yield pyramid.lowpass
        yield pyramid.highpasses
        if pyramid.scales is not None:
            yield pyramid.scales
    elif backend == 'hip':
        yield pyramid.hip_lowpass
        yield pyramid.hip_highpasses
        if pyramid.hip_scales is not None:
            yield pyramid.hip_scales
    else:
        raise ValueError('unknown backend "%s": use "numpy" or "hip"' % backend)

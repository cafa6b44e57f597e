"""ctypes binding of libdtcwt_hip.so (the C ABI declared in include/dtcwt_hip.h).

Plays the role dtcwt/opencl/lowlevel.py:1-21,150-181 plays for PyOpenCL in the reference:
import never fails; the first *use* without a usable runtime raises
:class:`NoHIPPresentError` (a ``RuntimeError``), never a silent CPU fallback.
"""
import ctypes
import importlib.util
import math
import os
import sys
import threading
import weakref

import numpy as np

__all__ = ['NoHIPPresentError', 'HipError', 'lib', 'have_hip', 'Context', 'DeviceArray',
           'default_context', 'View', 'F32', 'F64', 'ACCUMULATE', 'LIB_PATH']

LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        'libdtcwt_hip.so')

F32, F64 = 0, 1
ACCUMULATE = 1
MAX_TAPS = 40


class NoHIPPresentError(RuntimeError):
    """libdtcwt_hip.so is missing/unloadable or no HIP device is present."""


class HipError(RuntimeError):
    """A call into libdtcwt_hip.so failed."""


class View(ctypes.Structure):
    """struct dtcwt_hip_view"""
    _fields_ = [('outer', ctypes.c_int64), ('n', ctypes.c_int64), ('inner', ctypes.c_int64),
                ('xso', ctypes.c_int64), ('xsn', ctypes.c_int64), ('xsi', ctypes.c_int64),
                ('yso', ctypes.c_int64), ('ysn', ctypes.c_int64), ('ysi', ctypes.c_int64),
                ('pad_lo', ctypes.c_int32), ('pad_hi', ctypes.c_int32),
                ('crop_lo', ctypes.c_int32), ('crop_hi', ctypes.c_int32)]


_lib = None
_lib_error = None
_lock = threading.Lock()

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_sz = ctypes.c_size_t
_dbl = ctypes.c_double
_pd = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes); every symbol include/dtcwt_hip.h declares
ABI_VERSION = 6          # == DTCWT_HIP_ABI_VERSION of include/dtcwt_hip.h (checked in load_library)

SIGNATURES = {
    'dtcwt_hip_abi_version': (_i, []),
    'dtcwt_hip_last_error': (ctypes.c_char_p, []),
    'dtcwt_hip_device_count': (_i, [ctypes.POINTER(_i)]),
    'dtcwt_hip_device_info': (_i, [_i, ctypes.c_char_p, ctypes.POINTER(_i), ctypes.POINTER(_sz)]),
    'dtcwt_hip_ctx_create': (_i, [_i, _vp, ctypes.POINTER(_vp)]),
    'dtcwt_hip_ctx_create_partition': (_i, [_i, _i, _i, ctypes.POINTER(_vp)]),
    'dtcwt_hip_ctx_destroy': (_i, [_vp]),
    'dtcwt_hip_sync': (_i, [_vp]),
    'dtcwt_hip_ctx_stream': (_vp, [_vp]),
    'dtcwt_hip_device_sync': (_i, [_vp]),
    'dtcwt_hip_malloc': (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    'dtcwt_hip_free': (_i, [_vp, _vp]),
    'dtcwt_hip_trim': (_i, [_vp]),
    'dtcwt_hip_memcpy_h2d': (_i, [_vp, _vp, _vp, _sz]),
    'dtcwt_hip_memcpy_d2h': (_i, [_vp, _vp, _vp, _sz]),
    'dtcwt_hip_memcpy_d2d': (_i, [_vp, _vp, _vp, _sz]),
    'dtcwt_hip_to_float': (_i, [_vp, _i, _vp, _i, _vp, _i64]),
    'dtcwt_hip_host_alloc': (_i, [_sz, ctypes.POINTER(_vp)]),
    'dtcwt_hip_host_free': (_i, [_vp]),
    'dtcwt_hip_memcpy_h2d_async': (_i, [_vp, _vp, _vp, _sz]),
    'dtcwt_hip_memcpy_d2h_overlapped': (_i, [_vp, _vp, _vp, _sz]),
    'dtcwt_hip_copy_sync': (_i, [_vp]),
    'dtcwt_hip_memset': (_i, [_vp, _vp, _i, _sz]),
    'dtcwt_hip_event_create': (_i, [_vp, ctypes.POINTER(_vp)]),
    'dtcwt_hip_event_record': (_i, [_vp, _vp]),
    'dtcwt_hip_event_elapsed_ms': (_i, [_vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    'dtcwt_hip_event_destroy': (_i, [_vp]),
    'dtcwt_hip_colfilter': (_i, [_vp, _i, _vp, _vp, ctypes.POINTER(View), _pd, _i, _i]),
    'dtcwt_hip_coldfilt': (_i, [_vp, _i, _vp, _vp, ctypes.POINTER(View), _pd, _pd, _i, _i]),
    'dtcwt_hip_colifilt': (_i, [_vp, _i, _vp, _vp, ctypes.POINTER(View), _pd, _pd, _i, _i]),
    'dtcwt_hip_colfilter2': (_i, [_vp, _i, _vp, _vp, _vp, ctypes.POINTER(View), _pd, _i, _pd, _i]),
    'dtcwt_hip_colfilter_sum2': (_i, [_vp, _i, _vp, _vp, _vp, ctypes.POINTER(View), _pd, _i, _pd, _i]),
    'dtcwt_hip_coldfilt2': (_i, [_vp, _i, _vp, _vp, _vp, ctypes.POINTER(View), _pd, _pd, _pd, _pd, _i]),
    'dtcwt_hip_colifilt_sum2': (_i, [_vp, _i, _vp, _vp, _vp, ctypes.POINTER(View), _pd, _pd, _pd, _pd, _i]),
    'dtcwt_hip_q2c': (_i, [_vp, _i, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _i, _i]),
    'dtcwt_hip_fwd3_axis0_cube2c': (_i, [_vp, _i, _vp, _i64, _i64, _i64, _pd, _i, _pd, _i, _i, _i, _vp, _vp]),
    'dtcwt_hip_inv3_axis1_c2cube': (_i, [_vp, _i, _vp, _vp, _i64, _i64, _i64, _pd, _i, _pd, _i, _i, _i, _vp]),
    'dtcwt_hip_level1d_forward': (_i, [_vp, _i, _i, _vp, _i64, _i64, _i, _i, _pd, _pd, _pd, _pd, _i, _i, _vp, _vp]),
    'dtcwt_hip_level1d_inverse': (_i, [_vp, _i, _i, _vp, _vp, _i64, _i64, _dbl, _i, _pd, _pd, _pd, _pd, _i, _i, _vp]),
    'dtcwt_hip_level2d_forward': (_i, [_vp, _i, _i, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _pd, _pd, _pd, _pd,
                                       _i, _i, _vp, _vp, _vp, _vp]),
    'dtcwt_hip_level2d_inverse': (_i, [_vp, _i, _i, _vp, _vp, _i64, _i64, _i64, _pd, _i, _i, _pd, _pd, _pd,
                                       _pd, _i, _i, _vp, _vp, _vp]),
    'dtcwt_hip_c2q': (_i, [_vp, _i, _vp, _i64, _i64, _i64, _i, _i, _dbl, _dbl, _vp, _i64, _i64]),
    'dtcwt_hip_fwd3_level1': (_i, [_vp, _vp, _i64, _i64, _i64, _pd, _i, _pd, _i, _vp, _vp]),
    'dtcwt_hip_fwd3_level2': (_i, [_vp, _vp, _i64, _i64, _i64, _i, _i, _i, _pd, _pd, _pd, _pd, _i, _vp, _vp]),
    'dtcwt_hip_inv3_level1': (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _pd, _i, _pd, _i, _vp]),
    'dtcwt_hip_inv3_level2': (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _pd, _pd, _pd, _pd, _i, _vp]),
    'dtcwt_hip_sample': (_i, [_vp, _i, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i, _vp]),
    'dtcwt_hip_rescale': (_i, [_vp, _i, _vp, _i64, _i64, _i64, _i64, _i64, _i, _vp]),
    'dtcwt_hip_upsample2': (_i, [_vp, _i, _vp, _i64, _i64, _i64, _i, ctypes.POINTER(ctypes.c_int), _pd, _pd, _vp]),
    'dtcwt_hip_phase_roll_grid': (_i, [_vp, _i, _vp, _i64, _i64, _i64, _i, ctypes.POINTER(ctypes.c_int), _pd, _pd, _dbl, _dbl, _dbl, _vp]),
    'dtcwt_hip_phase_roll_points': (_i, [_vp, _i, _vp, _i64, _i64, _i, ctypes.POINTER(ctypes.c_int), _pd, _pd, _vp, _vp, _dbl, _vp]),
    'dtcwt_hip_qtilde': (_i, [_vp, _i, _vp, _vp, _i64, _i64, _dbl, _vp]),
    'dtcwt_hip_solve6': (_i, [_vp, _vp, _i64, _vp]),
    'dtcwt_hip_boxfilter': (_i, [_vp, _vp, _i64, _i64, _i64, _i, _vp]),
    'dtcwt_hip_colsum': (_i, [_vp, _vp, _i64, _i64, _vp]),
    'dtcwt_hip_affine_velocity': (_i, [_vp, _vp, _i64, _i64, _vp, _vp]),
    'dtcwt_hip_warp_coords': (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    'dtcwt_hip_axpy': (_i, [_vp, _i64, _dbl, _vp, _vp]),
    'dtcwt_hip_fill_rows': (_i, [_vp, _i64, _i, _pd, _vp]),
    'dtcwt_hip_estimatereg': (_i, [_vp, _i, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64), _i64, _i64,
                                   _i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), _vp]),
    'dtcwt_hip_cube2c': (_i, [_vp, _i, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _i]),
    'dtcwt_hip_c2cube': (_i, [_vp, _i, _vp, _i64, _i64, _i64, _i, _vp, _i64, _i64]),
    'dtcwt_hip_pack1d': (_i, [_vp, _i, _vp, _i64, _i64, _vp]),
    'dtcwt_hip_unpack1d': (_i, [_vp, _i, _vp, _i64, _i64, _dbl, _vp]),
    'dtcwt_hip_scale': (_i, [_vp, _i, _vp, _i64, _dbl]),
    'dtcwt_hip_plan2d_set_bandpass': (_i, [_vp, _pd, _pd, _i, _pd, _pd, _pd, _pd, _i]),
    'dtcwt_hip_plan2d_create': (_i, [_vp, _i, _i, _i, _i, ctypes.POINTER(_pd), ctypes.POINTER(_i),
                                     ctypes.POINTER(_pd), ctypes.POINTER(_i), ctypes.POINTER(_vp)]),
    'dtcwt_hip_plan2d_destroy': (_i, [_vp]),
    'dtcwt_hip_plan2d_shapes': (_i, [_vp, ctypes.POINTER(_i)]),
    'dtcwt_hip_plan2d_forward': (_i, [_vp, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    'dtcwt_hip_plan2d_inverse': (_i, [_vp, _vp, ctypes.POINTER(_vp), _pd, _vp]),
    'dtcwt_hip_plan2d_set_profiling': (_i, [_vp, _i]),
    'dtcwt_hip_plan2d_capture': (_i, [_vp, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _pd, _vp,
                                      ctypes.POINTER(_vp)]),
    'dtcwt_hip_graph_launch': (_i, [_vp]),
    'dtcwt_hip_graph_destroy': (_i, [_vp]),
    'dtcwt_hip_plan2d_kernel_ms': (_i, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
    'dtcwt_hip_plan2d_launches': (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    'dtcwt_hip_plan2d_set_concurrency': (_i, [_vp, _i]),
    'dtcwt_hip_plan2d_set_program': (_i, [_vp, _i]),
    'dtcwt_hip_plan2d_level1_march': (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    'dtcwt_hip_plan2d_describe': (_i, [_vp, _i, ctypes.c_char_p, ctypes.c_size_t]),
    'dtcwt_hip_plan3d_create': (_i, [_vp, _i64, _i64, _i64, _i, _i, ctypes.POINTER(_pd), ctypes.POINTER(_i),
                                     ctypes.POINTER(_pd), ctypes.POINTER(_i), ctypes.POINTER(_vp)]),
    'dtcwt_hip_plan3d_destroy': (_i, [_vp]),
    'dtcwt_hip_plan3d_shapes': (_i, [_vp, ctypes.POINTER(_i64)]),
    'dtcwt_hip_plan3d_forward': (_i, [_vp, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i]),
    'dtcwt_hip_plan3d_inverse': (_i, [_vp, _vp, ctypes.POINTER(_vp), _vp, _i]),
    'dtcwt_hip_plan1d_create': (_i, [_vp, _i, _i64, _i64, _i, ctypes.POINTER(_pd), ctypes.POINTER(_i),
                                     ctypes.POINTER(_pd), ctypes.POINTER(_i), ctypes.POINTER(_vp)]),
    'dtcwt_hip_plan1d_destroy': (_i, [_vp]),
    'dtcwt_hip_plan1d_shapes': (_i, [_vp, ctypes.POINTER(_i64)]),
    'dtcwt_hip_plan1d_forward': (_i, [_vp, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    'dtcwt_hip_plan1d_inverse': (_i, [_vp, _vp, ctypes.POINTER(_vp), _pd, _vp]),
    'dtcwt_hip_mgpu_create': (_i, [_i, ctypes.POINTER(_i), _i, _i, _i, _i, ctypes.POINTER(_pd), ctypes.POINTER(_i),
                                   ctypes.POINTER(_pd), ctypes.POINTER(_i), _i, ctypes.POINTER(_vp)]),
    'dtcwt_hip_mgpu_create_lane': (_i, [_i, ctypes.POINTER(_i), _i, _i, _i, _i, ctypes.POINTER(_pd), ctypes.POINTER(_i),
                                   ctypes.POINTER(_pd), ctypes.POINTER(_i), _i, _i, _i, ctypes.POINTER(_vp)]),
    'dtcwt_hip_mgpu_shares': (_i, [_vp]),
    'dtcwt_hip_mgpu_destroy': (_i, [_vp]),
    'dtcwt_hip_mgpu_ndev': (_i, [_vp]),
    'dtcwt_hip_mgpu_taps_broadcast': (_i, [_vp]),
    'dtcwt_hip_mgpu_shard': (_i, [_vp, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    'dtcwt_hip_mgpu_ctx': (_vp, [_vp, _i]),
    'dtcwt_hip_mgpu_shapes': (_i, [_vp, ctypes.POINTER(_i)]),
    'dtcwt_hip_mgpu_forward2d': (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    'dtcwt_hip_mgpu_forward2d_scales': (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    'dtcwt_hip_mgpu_inverse2d': (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _pd, ctypes.POINTER(_vp)]),
    'dtcwt_hip_mgpu_sync': (_i, [_vp]),
    'dtcwt_hip_mgpu_scatter': (_i, [_vp, _vp, _sz, ctypes.POINTER(_vp)]),
    'dtcwt_hip_mgpu_gather': (_i, [_vp, ctypes.POINTER(_vp), _sz, _vp]),
    'dtcwt_hip_mgpu_scatter_async': (_i, [_vp, _vp, _sz, ctypes.POINTER(_vp)]),
    'dtcwt_hip_mgpu_gather_async': (_i, [_vp, ctypes.POINTER(_vp), _sz, _vp]),
}


def _preload_runtime():
    """If PyTorch-ROCm is installed it ships its own libamdhip64.so; a process that uses
    both torch and this library must share ONE HIP runtime.  Loading torch's copy first
    (same SONAME as the system one) makes the dynamic linker bind both to it, whichever
    is imported first.  Set DTCWT_HIP_RUNTIME=system to skip."""
    if os.environ.get('DTCWT_HIP_RUNTIME', '') == 'system':
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(path=None):
    """Load (once) and return the ctypes handle; raises NoHIPPresentError."""
    global _lib, _lib_error
    with _lock:
        if _lib is not None:
            return _lib
        if _lib_error is not None:
            raise NoHIPPresentError(_lib_error)
        path = path or os.environ.get('DTCWT_HIP_LIBRARY', LIB_PATH)
        if not os.path.exists(path):
            _lib_error = ('%s not found: build it with `make -C dtcwt_amd/csrc` (or '
                          '`python -c "import __graft_entry__ as g; g.build()"`)' % path)
            raise NoHIPPresentError(_lib_error)
        _preload_runtime()
        try:
            handle = ctypes.CDLL(path)
        except OSError as e:
            _lib_error = 'cannot load %s: %s' % (path, e)
            raise NoHIPPresentError(_lib_error)
        # a stale build would otherwise surface as an AttributeError on the first missing entry point
        try:
            handle.dtcwt_hip_abi_version.restype = _i
            have = handle.dtcwt_hip_abi_version()
        except AttributeError:
            have = None
        if have != ABI_VERSION:
            _lib_error = ('%s has ABI version %r, this package needs %d: rebuild it with `make -C dtcwt_amd/csrc`'
                          % (path, have, ABI_VERSION))
            raise NoHIPPresentError(_lib_error)
        missing = [name for name in SIGNATURES if not hasattr(handle, name)]
        if missing:
            _lib_error = '%s lacks %s: rebuild it with `make -C dtcwt_amd/csrc`' % (path, ', '.join(missing[:4]))
            raise NoHIPPresentError(_lib_error)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
        return _lib


def lib():
    return load_library()


def check(rc):
    if rc != 0:
        msg = lib().dtcwt_hip_last_error()
        raise HipError('libdtcwt_hip: %s (code %d)' % (msg.decode('utf-8', 'replace') if msg else '?', rc))


def device_count():
    n = ctypes.c_int(0)
    try:
        rc = lib().dtcwt_hip_device_count(ctypes.byref(n))
    except NoHIPPresentError:
        return 0
    return n.value if rc == 0 else 0


def have_hip():
    """True when the library loads and at least one HIP device is visible."""
    return device_count() > 0


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float32 or dt == np.complex64:
        return F32
    if dt == np.float64 or dt == np.complex128:
        return F64
    raise TypeError('unsupported dtype %s' % dt)


def taps_arg(h):
    """(keep-alive array, double*, length) for a filter vector."""
    a = np.ascontiguousarray(np.asarray(h, dtype=np.float64).reshape(-1))
    if a.shape[0] > MAX_TAPS:
        raise ValueError('filters longer than %d taps are not supported by the hip backend' % MAX_TAPS)
    return a, a.ctypes.data_as(_pd), int(a.shape[0])


class Context(object):
    """A device + stream (``dtcwt_hip_ctx``).  ``stream`` may be a raw ``hipStream_t``
    value (e.g. ``torch.cuda.current_stream().cuda_stream``) to run on a caller's stream;
    the analogue of the OpenCL backend's ``queue`` (dtcwt/opencl/lowlevel.py:154-167).
    ``partition=(part, nparts)``: the context's own stream runs on one of ``nparts`` equal shares of the compute units
    (``dtcwt_hip_ctx_create_partition``) -- for ``nparts`` independent transforms in flight, one context each.
    The stream of a partition context is a *blocking* stream (``hipExtStreamCreateWithCUMask`` takes no flags): it
    synchronises implicitly with the legacy NULL stream, unlike the non-blocking stream of a plain context -- keep other
    device work (a framework's default stream) off the NULL stream while partition contexts are busy, or the transforms
    in flight serialise."""

    def __init__(self, device=0, stream=None, partition=None):
        L = lib()
        if device_count() < 1:
            raise NoHIPPresentError('no HIP device visible to libdtcwt_hip.so')
        h = _vp()
        if partition is not None:
            if stream:
                raise ValueError('Context: a caller\'s stream cannot be confined to a partition')
            part, nparts = partition
            check(L.dtcwt_hip_ctx_create_partition(int(device), int(part), int(nparts), ctypes.byref(h)))
        else:
            check(L.dtcwt_hip_ctx_create(int(device), _vp(stream) if stream else None, ctypes.byref(h)))
        self.partition = None if partition is None else (int(partition[0]), int(partition[1]))
        self._h = h
        self.device = int(device)
        self._lib = L

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        """The raw ``hipStream_t`` of this context as an integer (``__cuda_array_interface__`` v3 ``stream``)."""
        return int(self._lib.dtcwt_hip_ctx_stream(self._h) or 0)

    def sync(self):
        check(self._lib.dtcwt_hip_sync(self._h))

    def trim(self):
        """Return cached (freed) device buffers to the driver."""
        check(self._lib.dtcwt_hip_trim(self._h))

    def copy_sync(self):
        """Wait for the downloads enqueued with :meth:`DeviceArray.get_async`."""
        check(self._lib.dtcwt_hip_copy_sync(self._h))

    def device_sync(self):
        """hipDeviceSynchronize(): every stream of this device."""
        check(self._lib.dtcwt_hip_device_sync(self._h))

    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype):
        a = DeviceArray(self, shape, dtype)
        check(self._lib.dtcwt_hip_memset(self._h, a.ptr, 0, a.nbytes))
        return a

    def to_device(self, X, dtype=None):
        X = np.ascontiguousarray(X, dtype=dtype)
        a = DeviceArray(self, X.shape, X.dtype)
        check(self._lib.dtcwt_hip_memcpy_h2d(self._h, a.ptr, X.ctypes.data_as(_vp), X.nbytes))
        return a

    _INT_KINDS = {'u1': 0, 'i1': 1, 'u2': 2, 'i2': 3, 'u4': 4, 'i4': 5, 'u8': 6, 'i8': 7, 'b1': 8}

    def to_device_float(self, X):
        """Upload a real array as floating point the way the reference's ``asfarray`` decides it
        (dtcwt/utils.py:98-105): float32 / float64 as they are, everything else float64 -- integer
        and bool arrays travel in their own width and are widened on the device
        (dtcwt_hip_to_float; a uint8 image is 1/8 of the float64 bytes over the host link)."""
        X = np.asanyarray(X)
        if X.dtype == np.float32 or X.dtype == np.float64:
            return self.to_device(X)
        if np.issubdtype(X.dtype, np.complexfloating):
            # the reference keeps complex input complex (np.asfarray(X, dtype=X.dtype), dtcwt/utils.py:98-105) and
            # filters it as such; the device kernels are real: refuse instead of silently dropping the imaginary part
            raise TypeError('complex input is not supported by the hip backend: transform the real and the '
                            'imaginary part separately (the transform is linear)')
        kind = self._INT_KINDS.get(X.dtype.str[1:]) if X.dtype.isnative or X.dtype.itemsize == 1 else None
        if kind is None or X.size == 0:
            return self.to_device(X.astype(np.float64))
        raw = self.to_device(X)
        out = DeviceArray(self, X.shape, np.float64)
        check(self._lib.dtcwt_hip_to_float(self._h, kind, raw.ptr, F64, out.ptr, X.size))
        return out

    def convert(self, a, dtype):
        """A device array in another floating-point precision (float32 <-> float64, complex64 <-> complex128), converted
        on the device (dtcwt_hip_to_float kinds 9 / 10)."""
        dtype = np.dtype(dtype)
        if a.dtype == dtype:
            return a
        pairs = {(np.dtype(np.float32), np.dtype(np.float64)): (9, F64, 1), (np.dtype(np.float64), np.dtype(np.float32)): (10, F32, 1),
                 (np.dtype(np.complex64), np.dtype(np.complex128)): (9, F64, 2), (np.dtype(np.complex128), np.dtype(np.complex64)): (10, F32, 2)}
        kind, dst, mul = pairs[(np.dtype(a.dtype), dtype)]
        out = DeviceArray(self, a.shape, dtype)
        if a.size:
            check(self._lib.dtcwt_hip_to_float(self._h, kind, a.ptr, dst, out.ptr, a.size * mul))
        return out

    def event(self):
        return Event(self)

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self._lib.dtcwt_hip_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Event(object):
    def __init__(self, ctx):
        self.ctx = ctx
        h = _vp()
        check(ctx._lib.dtcwt_hip_event_create(ctx.handle, ctypes.byref(h)))
        self._h = h

    def record(self):
        check(self.ctx._lib.dtcwt_hip_event_record(self.ctx.handle, self._h))
        return self

    def elapsed_ms(self, later):
        ms = ctypes.c_float(0)
        check(self.ctx._lib.dtcwt_hip_event_elapsed_ms(self._h, later._h, ctypes.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if getattr(self, '_h', None) and getattr(self.ctx, '_h', None):
                self.ctx._lib.dtcwt_hip_event_destroy(self._h)
            self._h = None
        except Exception:
            pass


class _HostPool(object):
    """Recycles the host memory of downloaded arrays.

    A device-to-host copy into a fresh ``np.empty`` spends 5x longer on first-touch page faults
    than on the copy itself (4096^2 pyramid: 31 ms vs 6 ms, tools/bench_host_api.py).  Arrays
    returned by :meth:`DeviceArray.get` are views of a byte buffer; when such an array is
    garbage collected AND nothing else (a slice, a reshaped view ...) still refers to the buffer,
    the buffer goes back to the pool for the next download of the same size.  Buffers under
    1 MiB are not pooled; ``DTCWT_HIP_HOST_POOL_MB`` bounds the idle memory (default 2048, 0
    disables).

    The buffers are PAGE-LOCKED (``dtcwt_hip_host_alloc`` = hipHostMalloc; ``DTCWT_HIP_PINNED_HOST=0``
    or a failed allocation falls back to pageable ``np.empty``): a download into them is one DMA
    transfer at the link rate with no staging copy, and it can be enqueued asynchronously
    (:meth:`DeviceArray.get_async`, ``Pyramid.prefetch``) -- the lazy host copies of the reference's
    OpenCL pyramid (dtcwt/opencl/transform2d.py:30-84), overlapped."""

    MIN_BYTES = 1 << 20

    def __init__(self):
        self._free = {}
        self._idle = 0
        # re-entrant: dropping the last reference to a page-locked buffer runs its finaliser (_unpin, which takes the
        # lock for the byte count) wherever that happens -- also inside trim() / give() while they hold it
        self._lock = threading.RLock()
        self.limit = int(os.environ.get('DTCWT_HIP_HOST_POOL_MB', '2048')) << 20
        self.pinned = os.environ.get('DTCWT_HIP_PINNED_HOST', '1') != '0'
        self.pinned_bytes = 0           # page-locked bytes currently allocated through this pool
        self._quiet_refs = None
        self._quiet_refs = self._calibrate()

    def _calibrate(self):
        # references to a buffer seen inside give() when no user view exists; the buffer is built the way
        # _new_base builds a page-locked one (np.frombuffer over a ctypes array), which is also an ndarray that
        # does not own its data -- like every pooled base once it has been handed out as a view
        seen = []
        raw = (ctypes.c_uint8 * 8)()
        base = np.frombuffer(raw, dtype=np.uint8)
        arr = base.view(np.float32).reshape(2)
        weakref.finalize(arr, self.give, base, seen)
        del base, arr
        return seen[0] if seen else -1

    def _unpin(self, ptr, nbytes):
        try:
            if _lib is not None:
                _lib.dtcwt_hip_host_free(ptr)
        finally:
            with self._lock:            # finalisers run on whichever thread drops the last view
                self.pinned_bytes -= nbytes

    def _new_base(self, nbytes):
        """A byte buffer for downloads: page-locked when the library is there, pageable otherwise."""
        if self.pinned and _lib is not None:
            p = _vp()
            if _lib.dtcwt_hip_host_alloc(nbytes, ctypes.byref(p)) == 0 and p.value:
                raw = (ctypes.c_uint8 * nbytes).from_address(p.value)
                base = np.frombuffer(raw, dtype=np.uint8)
                with self._lock:
                    self.pinned_bytes += nbytes
                weakref.finalize(base, self._unpin, p.value, nbytes)      # when the last view of it is gone
                return base
        # pageable: over the memory of an uninitialised array (a bytearray would be zero-filled first: one more pass
        # over hundreds of MB); like the page-locked base it is an ndarray that does not own its data and is the
        # .base of every view handed out, which is what give() counts on
        return np.frombuffer(np.empty(nbytes, dtype=np.uint8).data, dtype=np.uint8)

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = math.prod(int(v) for v in shape) * dtype.itemsize
        if nbytes < self.MIN_BYTES or self.limit <= 0 or self._quiet_refs < 0:
            return np.empty(shape, dtype=dtype)
        base = None
        with self._lock:
            lst = self._free.get(nbytes)
            if lst:
                base = lst.pop()
                self._idle -= nbytes
        if base is None:
            base = self._new_base(nbytes)
        arr = base.view(dtype).reshape(shape)
        weakref.finalize(arr, self.give, base)
        return arr

    def give(self, base, probe=None):
        refs = sys.getrefcount(base)
        if probe is not None:
            probe.append(refs)
            return
        if refs != self._quiet_refs:        # some view of the array outlived it: not ours to reuse
            return
        with self._lock:
            if self._idle + base.nbytes <= self.limit:
                self._free.setdefault(base.nbytes, []).append(base)
                self._idle += base.nbytes

    def trim(self):
        with self._lock:
            old, self._free = self._free, {}
            self._idle = 0
        old.clear()             # the buffers' finalisers run here, outside the lock


host_pool = _HostPool()


class DeviceArray(object):
    """A C-contiguous array in HBM: pointer + shape + dtype.  Owns its memory unless
    wrapping a foreign pointer (``DeviceArray.wrap``)."""

    def __init__(self, ctx, shape, dtype, ptr=None, owner=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.size = math.prod(self.shape)         # (np.prod costs 2 us a call: the top host cost of a transform)
        self.nbytes = self.size * self.dtype.itemsize
        self._owner = owner
        if ptr is None:
            p = _vp()
            check(ctx._lib.dtcwt_hip_malloc(ctx.handle, self.nbytes, ctypes.byref(p)))
            self.ptr = p.value
            self._owned = True
        else:
            self.ptr = int(ptr)
            self._owned = False

    @classmethod
    def wrap(cls, ctx, obj):
        """Zero-copy view of an on-device tensor (anything with ``data_ptr()``, ``shape``
        and ``dtype`` such as a contiguous torch CUDA/ROCm tensor), by analogy with the
        OpenCL backend accepting pyopencl arrays (dtcwt/opencl/transform2d.py:133-135)."""
        if hasattr(obj, 'is_contiguous') and not obj.is_contiguous():
            raise ValueError('device tensors must be contiguous')
        dt = np.dtype(str(obj.dtype).replace('torch.', ''))
        if dt not in (np.dtype(np.float32), np.dtype(np.float64), np.dtype(np.complex64), np.dtype(np.complex128)):
            raise TypeError('device tensors must be float32 / float64 (complex64 / complex128 for subbands), not %s' % dt)
        # The producer's stream (torch's current stream) and this context's stream are unrelated: nothing
        # orders our kernels after the work that filled the tensor.  Wait for the device once, here.
        check(ctx._lib.dtcwt_hip_device_sync(ctx.handle))
        return cls(ctx, tuple(obj.shape), dt, ptr=obj.data_ptr(), owner=obj)

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def __cuda_array_interface__(self):
        # version 3: consumers (torch, cupy) synchronise with the stream our kernels run on before reading
        st = getattr(self.ctx, 'stream', 0)
        return {'shape': self.shape, 'typestr': self.dtype.str, 'data': (self.ptr, False),
                'version': 3, 'strides': None, 'stream': st if st else 1}

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        assert math.prod(int(v) for v in shape) == self.size
        return DeviceArray(self.ctx, shape, self.dtype, ptr=self.ptr, owner=self)

    def view(self, dtype):
        dtype = np.dtype(dtype)
        shape = list(self.shape)
        shape[-1] = shape[-1] * self.dtype.itemsize // dtype.itemsize
        return DeviceArray(self.ctx, shape, dtype, ptr=self.ptr, owner=self)

    def get(self):
        """Copy to a NumPy array (synchronises the stream)."""
        out = host_pool.empty(self.shape, self.dtype)
        check(self.ctx._lib.dtcwt_hip_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(_vp), self.ptr,
                                                 self.nbytes))
        return out

    def get_async(self):
        """Enqueue the download on the context's copy stream (ordered after everything issued so far, not holding
        up what is issued next) and return the destination array; its contents are valid after
        ``ctx.copy_sync()``."""
        out = host_pool.empty(self.shape, self.dtype)
        check(self.ctx._lib.dtcwt_hip_memcpy_d2h_overlapped(self.ctx.handle, out.ctypes.data_as(_vp), self.ptr,
                                                            self.nbytes))
        return out

    def set(self, X):
        X = np.ascontiguousarray(X, dtype=self.dtype)
        assert X.shape == self.shape
        check(self.ctx._lib.dtcwt_hip_memcpy_h2d(self.ctx.handle, self.ptr, X.ctypes.data_as(_vp), X.nbytes))
        return self

    def copy(self):
        out = DeviceArray(self.ctx, self.shape, self.dtype)
        check(self.ctx._lib.dtcwt_hip_memcpy_d2d(self.ctx.handle, out.ptr, self.ptr, self.nbytes))
        return out

    def __del__(self):
        try:
            if getattr(self, '_owned', False) and self.ptr and getattr(self.ctx, '_h', None):
                self.ctx._lib.dtcwt_hip_free(self.ctx.handle, self.ptr)
                self.ptr = 0
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    """Memoised context on ``device`` (default: env DTCWT_HIP_DEVICE, LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get('DTCWT_HIP_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        n = device_count()
        if n:
            device %= n
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def is_device_array(X):
    return isinstance(X, DeviceArray)


def is_foreign_device_tensor(X):
    """torch-like tensor living on a GPU."""
    return hasattr(X, 'data_ptr') and getattr(X, 'is_cuda', False)

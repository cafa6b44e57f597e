"""dtcwt_amd: an MI355X-native ``hip`` backend for the Dual-Tree Complex Wavelet Transform.

This package reproduces the backend-dispatch surface of rjw57/dtcwt
(dtcwt/__init__.py:24-143): ``push_backend`` / ``pop_backend`` /
``preserve_backend_stack`` / ``backend_name`` and the module attributes ``Transform1d``,
``Transform2d``, ``Transform3d``, ``Pyramid`` that they rebind.  It ships exactly one
backend of its own, ``'hip'`` (hand-written gfx950 kernels behind a C ABI, see
include/dtcwt_hip.h); the name ``'numpy'`` resolves to the upstream ``dtcwt.numpy`` classes
when the upstream package is installed next to this one, and ``install()`` registers
``'hip'`` in an installed upstream ``dtcwt`` so that ``dtcwt.push_backend('hip')`` works
there verbatim.  There is no CPU fallback inside the ``hip`` backend.
"""
import os
import sys

__all__ = [
    '__version__', 'Transform1d', 'Transform2d', 'Transform3d', 'Pyramid', 'backend_name',
    'push_backend', 'pop_backend', 'preserve_backend_stack', 'register_backend', 'install',
]
__version__ = '0.1.0'

import dtcwt_amd.hip

_THIS = sys.modules[__name__]

# An array of (name, table) pairs: the backend stack (dtcwt/__init__.py:24-26).
_BACKEND_STACK = []

# Possible backends keyed by name (dtcwt/__init__.py:28-48).
_AVAILABLE_BACKENDS = {
    'hip': {
        'Transform1d': dtcwt_amd.hip.Transform1d,
        'Transform2d': dtcwt_amd.hip.Transform2d,
        'Transform3d': dtcwt_amd.hip.Transform3d,
        'Pyramid': dtcwt_amd.hip.Pyramid,
    },
}


def _upstream_numpy_table():
    """The reference's own NumPy backend, if rjw57/dtcwt is importable."""
    try:
        import dtcwt.numpy as up
    except Exception:
        return None
    return {'Transform1d': up.Transform1d, 'Transform2d': up.Transform2d,
            'Transform3d': up.Transform3d, 'Pyramid': up.Pyramid}


def register_backend(name, table):
    """Make a backend selectable by name.  *table* maps 'Transform1d', 'Transform2d',
    'Transform3d', 'Pyramid' to classes (the shape of dtcwt/__init__.py:29-48)."""
    missing = [k for k in ('Transform1d', 'Transform2d', 'Transform3d', 'Pyramid') if k not in table]
    if missing:
        raise ValueError('backend table lacks {0}'.format(', '.join(missing)))
    _AVAILABLE_BACKENDS[name] = dict(table)


def _update_from_current_backend():
    for k, v in _BACKEND_STACK[-1][1].items():
        setattr(_THIS, k, v)
    _THIS.backend_name = _BACKEND_STACK[-1][0]


class _BackendGuard(object):
    """Context manager restoring the backend stack (dtcwt/__init__.py:56-76)."""

    def __init__(self, stack):
        self._stack = list(stack)

    def __enter__(self):
        return _BACKEND_STACK

    def __exit__(self, exc_type, exc_value, exc_tb):
        _BACKEND_STACK[:] = self._stack
        _update_from_current_backend()
        return False


def preserve_backend_stack():
    """``with preserve_backend_stack(): push_backend(...)`` restores the stack afterwards,
    exception or not (dtcwt/__init__.py:78-95)."""
    return _BackendGuard(_BACKEND_STACK)


def push_backend(name):
    """Switch backend implementation to *name*, remembering the previous one.

    :raises ValueError: if *name* does not correspond to a known backend
        (dtcwt/__init__.py:97-116)."""
    if name == 'numpy' and 'numpy' not in _AVAILABLE_BACKENDS:
        table = _upstream_numpy_table()
        if table is not None:
            _AVAILABLE_BACKENDS['numpy'] = table
    try:
        _BACKEND_STACK.append((name, _AVAILABLE_BACKENDS[name]))
    except KeyError:
        raise ValueError('No such backend: {0}'.format(name))
    _update_from_current_backend()


def pop_backend():
    """Undo the most recent :func:`push_backend`.

    :raises IndexError: when only the default backend is left (dtcwt/__init__.py:118-131)."""
    if len(_BACKEND_STACK) <= 1:
        raise IndexError('Cannot pop default backend')
    _BACKEND_STACK.pop()
    _update_from_current_backend()


def install(dtcwt_module=None):
    """Register the ``'hip'`` backend in an installed upstream ``dtcwt`` package so that
    ``dtcwt.push_backend('hip')`` selects it (the two-line patch INTEGRATION.md shows, done
    at run time).  Returns the module."""
    if dtcwt_module is None:
        import dtcwt as dtcwt_module
    dtcwt_module._AVAILABLE_BACKENDS['hip'] = dict(_AVAILABLE_BACKENDS['hip'])
    return dtcwt_module


backend_name = None
"""Name of the backend currently bound to the module attributes.  The default is
``'hip'``; override with the DTCWT_BACKEND environment variable (dtcwt/__init__.py:133-143)."""

_default = os.getenv('DTCWT_BACKEND', 'hip')
try:
    push_backend(_default)
except ValueError:
    push_backend('hip')

"""The ``hip`` backend: DT-CWT on AMD Instinct MI355X (gfx950).

Exposes what a reference backend package exposes (cf. dtcwt/numpy/__init__.py,
dtcwt/opencl/__init__.py): ``Transform1d``, ``Transform2d``, ``Transform3d``, ``Pyramid``,
plus the device plumbing (``Context``, ``DeviceArray``) and the low-level filters.
Importing never touches the GPU; the first use without libdtcwt_hip.so or a device raises
:class:`NoHIPPresentError`.
"""
from dtcwt_amd.hip._lib import (NoHIPPresentError, HipError, Context, DeviceArray, default_context,
                                have_hip, device_count)
from dtcwt_amd.hip.common import Pyramid
from dtcwt_amd.hip.transform1d import Transform1d
from dtcwt_amd.hip.transform2d import Transform2d
from dtcwt_amd.hip.transform3d import Transform3d

__all__ = ['Pyramid', 'Transform1d', 'Transform2d', 'Transform3d', 'Context', 'DeviceArray',
           'default_context', 'have_hip', 'device_count', 'NoHIPPresentError', 'HipError']

"""``dtcwt.registration`` of the reference (dtcwt/registration.py), served by the ``hip``
backend: the pyramids stay on the device, every per-pixel loop is a kernel
(``dtcwt_amd/hip/registration.py``)."""
from dtcwt_amd.hip.registration import (estimatereg, velocityfield, warp, warptransform,       # noqa: F401
                                        qtildematrices, solvetransform, warphighpass, normsample,
                                        normsamplehighpass, EXPECTED_SHIFTS)

__all__ = [
    'estimatereg',
    'velocityfield',
    'warp',
    'warptransform',
]

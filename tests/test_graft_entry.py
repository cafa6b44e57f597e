"""The driver's entry points: build() must succeed on a box without a GPU (it is the round's "does it build" check)
and must not pin a stale constant (round 3: it asserted ABI version 1 after the header had moved to 2)."""
import re
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_checks_the_abi_version_symbolically():
    src = open(os.path.join(ROOT, '__graft_entry__.py')).read()
    assert re.search(r'dtcwt_hip_abi_version\(\)\s*==\s*_lib\.ABI_VERSION', src)
    assert not re.search(r'dtcwt_hip_abi_version\(\)\s*==\s*\d', src)


def test_entry_points_exist():
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionfinish(session, exitstatus):
    """The worst relative error each tolerance class came to in this session (tests/_hip.py: assert_close), written
    next to the other GPU-box outputs: the bounds say what must hold, this says what did."""
    try:
        from tests import _hip
        if not _hip.WORST:
            return
        import json
        out = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_worst.json'), 'w') as f:
            json.dump(_hip.WORST, f, indent=1, sort_keys=True)
    except Exception:
        pass

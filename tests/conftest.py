import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def golden():
    import json
    return json.loads((ROOT / "tests" / "golden" / "golden_v1.json").read_text())


@pytest.fixture(scope="session")
def engine(pkg):
    """The HIP engine on device 0.  No fallback: a missing library or device is a hard failure."""
    eng = pkg.GPEngine(0)
    yield eng
    eng.close()


def to_tuple(t):
    """JSON lists -> nested tuples of the oracle tree form."""
    if isinstance(t, list):
        return tuple(to_tuple(x) for x in t)
    return t

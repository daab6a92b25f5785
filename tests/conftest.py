import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One library context on GPU 0 for the whole session (GPU tests only)."""
    from vectorwave_amd import _native as N
    c = N.Context(0)
    yield c
    c.close()


@pytest.fixture
def plan(ctx):
    """Set plan options (vw_set_plan_option) for one test; all of them are reset afterwards."""
    def set_option(name, value):
        ctx.set_plan_option(name, value)
    yield set_option
    ctx.set_plan_option("reset", 0)

import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.lib()
    return oracle_lib


def apply_test_options(dev):
    """TRINITY_TEST_OPTIONS="planes=0,plane_div=1073741824": planner options (tri_dev_set_option) for every batch of a test run — the
    GPU suite is run again under the option sets that force one execution path (tools/gpu_round.sh `testopt:` stage)."""
    spec = os.environ.get("TRINITY_TEST_OPTIONS", "")
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=", 1)
        dev.set_option(k.strip(), int(v))
    return dev

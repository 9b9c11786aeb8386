import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def oracle_parity():
    from lane_slam_amd.config import default_config
    from oracle.oracle import Oracle
    return Oracle(default_config("parity"))


@pytest.fixture(scope="session")
def oracle_fullres():
    from lane_slam_amd.config import default_config
    from oracle.oracle import Oracle
    return Oracle(default_config("fullres"))

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL_LITE0 = os.path.join(ROOT, "models", "efficientdet_lite0_synth.vbtm")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model_path():
    return MODEL_LITE0


@pytest.fixture(scope="session")
def oracle_lib():
    """Build (if needed) and load the CPU oracle. Test infrastructure only."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libvbt_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    from oracle import detector_ref
    return detector_ref

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle.load_api()


@pytest.fixture(scope="session")
def hip_api():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dynslam_amd.engine import load_hip_api
    return load_hip_api()  # raises loudly when libdsr_hip.so is missing: no fallback

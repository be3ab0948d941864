import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu_oracle
    cpu_oracle.lib()
    return cpu_oracle


@pytest.fixture(scope="session")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import lcrec_amd
    lcrec_amd._lib.load()   # raises if the extension is missing: GPU tests never fall back
    return lcrec_amd

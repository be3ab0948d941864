import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a MI355X (run on the GPU box via gpurun)")


def pytest_sessionstart(session):
    """A fresh checkout has no binaries (they are git-ignored): build the C-ABI library and the oracle once,
    the way __graft_entry__.build() does, before any test looks at them.  hipcc cross-compiles without a GPU."""
    lib = os.path.join(ROOT, "lc-rec_amd", "csrc", "liblcrec_hip.so")
    ora = os.path.join(ROOT, "oracle", "liblcrec_oracle.so")
    if os.path.exists(lib) and os.path.exists(ora):
        return
    import subprocess
    for d in (os.path.join(ROOT, "lc-rec_amd", "csrc"), os.path.join(ROOT, "oracle")):
        subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)


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

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle (CPU) side of the parity tests: torch's default thread count on the GPU box is that host's hardware threads (256), which is several times
    # SLOWER for these small GEMMs than 16-32 threads (bench.py's cpu_baseline probe: 7 s per step at 32 threads, 33 s at 128, 319 s at 256).
    # CLIBD_TEST_THREADS overrides; an 8-CPU container keeps its 8.
    import torch
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    want = int(os.environ.get("CLIBD_TEST_THREADS", "0")) or min(32, ncpu)
    if torch.get_num_threads() > want:
        torch.set_num_threads(want)


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")

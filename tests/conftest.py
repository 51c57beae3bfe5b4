import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The CPU oracle runs on torch's intra-op pool, which sizes itself by the machine's logical CPUs.  A GPU box shows all of the
# host's CPUs and grants 16 of them: left alone, the pool oversubscribes its share many times over and the oracle passes of the
# model parity tests take 85 s instead of 9 (round 4: half of the GPU suite's time).  Cap the pool at the share before torch
# starts it - environment for OpenMP / MKL, and torch.set_num_threads once torch is imported (pytest_sessionstart).
try:
    _CPUS = len(os.sched_getaffinity(0))
except (AttributeError, OSError):
    _CPUS = os.cpu_count() or 1
_CPUS = max(1, min(_CPUS, 16))
for _v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(_CPUS))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    try:
        import torch
        torch.set_num_threads(_CPUS)
    except Exception:   # torch is imported by the tests that need it; a missing torch fails there, not here
        pass


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from building_detection_amd.ops import get_engine
    return get_engine(0)

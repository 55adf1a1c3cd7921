import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The engine launches its 64 x 144 latency tiles whenever a GEMM has few rows (batch 1-4) -- i.e. in nearly every engine test here.  The suite keeps them
# OFF by default so that the small-batch oracle tests go on exercising the 256-row kernels the headline configuration runs; the latency tiles have their own
# per-tile parity cases (test_gpu_ops.py, kernel "latency64") and engine-level tests that switch them on (test_gpu_engine.py::test_latency_tiles_*), where
# they must reproduce the 256-row kernels bit for bit.
os.environ.setdefault("ECAD_LATENCY_TILES", "0")
# ... except in the modules that hold the engine / plug-in / generator / GA oracle tests: every test there runs TWICE, once per setting (`rows256`: the 256-row
# kernels of the headline batch; `latency64`: the product default for batch 1-4, which is what these small-batch tests then launch), so the oracle sees both.
# The variable is read at ecad_create, and these tests create their engines inside the test.
BOTH_TILE_MODES = ("test_gpu_engine", "test_gpu_plugins", "test_gpu_generator", "test_gpu_genetic")


def _usable_cores() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.rsplit(".", 1)[-1] in BOTH_TILE_MODES:
        metafunc.parametrize("few_rows_tiles", ["rows256", "latency64"], indirect=True)


@pytest.fixture(autouse=True)
def few_rows_tiles(request, monkeypatch):
    """ECAD_LATENCY_TILES for this test (parametrised over both settings in BOTH_TILE_MODES modules; the session default elsewhere)."""
    mode = getattr(request, "param", None)
    if mode is not None:
        monkeypatch.setenv("ECAD_LATENCY_TILES", "1" if mode == "latency64" else "0")
    return mode


def pytest_configure(config):
    import torch

    torch.set_num_threads(_usable_cores())  # a 256-thread pool on a 16-CPU quota crawls
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def repo_root() -> Path:
    return ROOT


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return ROOT / "tests" / "golden"

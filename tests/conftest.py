import os
import sys

import pytest
import torch  # noqa: F401  (before anything loads libpandora_mi355x.so: one HIP runtime per process)

# Eager PyTorch on the many small ops of the CPU oracle scales NEGATIVELY past ~16 threads (bench.py cpu_baseline, measured on
# the GPU box's 2 x 64-core host: 16 threads 3.8 s, 128 threads 25.7 s for the same U-Net forward): cap the pool for the whole
# session - the oracle sides of the GPU parity tests are a large part of the suite's wall time.
torch.set_num_threads(min(16, os.cpu_count() or 16))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU (full-width oracle / live-reference runs); "
                                       'casual runs: -m "not gpu and not slow"')


# Measured margins, not just "<= threshold": every `[parity]` / `[budget]` / `[scaling]` line a test prints is collected from
# the captured output and repeated in the terminal summary, so the tail of a `pytest -q` log (the driver's GPUTEST record)
# carries the numbers themselves.
_MEASURED = []


def pytest_runtest_logreport(report):
    if report.when != "call":
        return
    for line in (report.capstdout or "").splitlines():
        line = line.strip()
        if line.startswith(("[parity]", "[budget]", "[scaling]")):
            _MEASURED.append(line)


def pytest_terminal_summary(terminalreporter):
    if not _MEASURED:
        return
    terminalreporter.section("measured margins ([parity] / [budget] lines of the tests that ran)")
    for line in _MEASURED:
        terminalreporter.write_line(line)


@pytest.fixture(scope="session")
def hip_ops_factory():
    import torch
    from open_pandora_amd.ops_hip import HipOps

    cache = {}

    def make(dtype, diag=False):
        if (dtype, diag) not in cache:
            cache[(dtype, diag)] = HipOps(dtype=dtype, device="cuda:0", diag=diag)
        return cache[(dtype, diag)]

    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return make

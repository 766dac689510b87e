import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # The oracle (torch-CPU, oracle/) is the slow side of every parity test, and torch's intra-op pool does not help its op mix
    # beyond a few threads: on the GPU box's 2 x 64 cores the headline image runs at 552 detections/s with 16 threads and at 108
    # with all 128 (bench.py cpu_baseline).  16 threads for the whole test session (results do not depend on the thread count: the
    # oracle's reductions are per-row / per-segment).
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:      # noqa: BLE001 -- torch is a test dependency; its absence is reported by the tests themselves
        pass

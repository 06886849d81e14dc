import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the hipGraph-replayed training step (tests/test_train_step_gpu.py) needs ROCm 7.2's graph packet capture switched off before
    # the HIP runtime starts; the library no longer does that on import -- the test harness opts in here, before any GPU call
    from temporalstereo_amd import train
    train.enable_graph_replay()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

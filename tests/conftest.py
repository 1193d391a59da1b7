import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # The f32 oracle is the yardstick of several tolerances, and its rounding depends on torch's intra-op thread count
    # (summation order of scatter / GEMM): pin it, so that a test's verdict does not depend on which tests ran before
    # (GraphLoader lowers the count) or on the host's core count.  8-32 threads are also the oracle's fastest setting.
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))

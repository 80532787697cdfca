import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The boundary-merge test hands torch tensors to the library, so two users of the HIP
    runtime share the test process.  torch goes first, as in bench.py: brought up after the
    library had used the device it has been seen to report "No HIP GPUs are available"."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
            torch.cuda.synchronize()
    except Exception:  # no torch, no device: the CPU suite does not need either
        pass

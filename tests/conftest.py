import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    import torch
    from depth_image_captioning_pub_amd.hostinfo import host_cores
    torch.set_num_threads(host_cores())      # the GPU box exposes 256 CPUs but grants a 16-CPU quota
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The C-ABI library; GPU tests call the product through it."""
    from depth_image_captioning_pub_amd import _lib
    return _lib.load()

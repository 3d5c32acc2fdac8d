import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import codlad_amd  # noqa: E402,F401  (as every entry point does: before the first HIP call, see codlad_amd/__init__.py)

torch.set_grad_enabled(False)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # device_count() does not initialise the HIP runtime in this process (is_available() does): the multi-rank GPU
    # test starts its rank processes before this process has touched the GPU
    if torch.cuda.device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

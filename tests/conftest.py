import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def engine():
    from genima_amd.engine import Engine

    return Engine("cuda:0")


@pytest.fixture(autouse=True)
def _collector_sees_everything_between_tests():
    """ControlNetTrainer(gc_freeze=True) parks every live object in the permanent generation after its second step (training.py);
    between tests they go back under the collector, so a finished test's trainer (and its device buffers) is reclaimed."""
    yield
    import gc

    gc.unfreeze()
    gc.collect()

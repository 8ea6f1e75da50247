import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-omni_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch

    return torch.device("cuda:0")


@pytest.fixture
def plan_registry():
    """An empty tile-plan registry (and an unloaded tuned table) for tests that compare entry points of the decode-batch
    linear kernel bit for bit: the tuned table may give ``linear`` and ``linear_partials`` different split-K factors for one
    shape, and the whole table is registered the first time any test routes a layer through ``layers.py``."""
    from vllm_omni_amd import _C
    from vllm_omni_amd.tuned import linear_plans

    linear_plans.reset()
    yield _C
    linear_plans.reset()

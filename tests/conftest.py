import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_ops():
    from oracle import ops
    ops.build()
    return ops


@pytest.fixture(scope="session")
def oracle_graph(oracle_ops):
    from oracle import graph
    return graph


@pytest.fixture(scope="session")
def ssd():
    import ssd_amd
    return ssd_amd


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test collected without a GPU (the HIP path has no CPU fallback)")
    return torch


TINY_PARAMS = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80,
               "score_threshold": 0.15, "iou_threshold": 0.6, "max_boxes_per_class": 25,
               "min_dimension": 128}

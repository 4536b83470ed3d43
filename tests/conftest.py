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


@pytest.fixture
def libopt(ssd):
    """libopt(key=value, ...): process-wide library options (ssd_set_option with a NULL handle, include/ssd_hip.h) for one
    test -- kernel / schedule selectors that must not change a result bit; the previous values come back afterwards."""
    saved = []

    def set_options(**kw):
        for k, v in kw.items():
            saved.append((k, ssd.get_option(k)))
            ssd.set_option(k, int(v))
    yield set_options
    for k, old in reversed(saved):
        ssd.set_option(k, old)          # INT32_MIN = "unset"


TINY_PARAMS = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80,
               "score_threshold": 0.15, "iou_threshold": 0.6, "max_boxes_per_class": 25,
               "min_dimension": 128}

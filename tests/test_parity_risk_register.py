"""The oracle-side alternates of the parity risk register (oracle/ssd_oracle.c, tests/parity_risk.py): every switch changes what
it says it changes, nothing else, and is off by default -- the default reading is the one every parity test pins."""
import numpy as np
import pytest


@pytest.fixture
def alt(oracle_ops):
    def set_alt(name, v):
        oracle_ops.set_alternate(name, v)
    yield set_alt
    for name in oracle_ops.ALTERNATES:
        oracle_ops.set_alternate(name, 0)


def test_alternates_default_to_off(oracle_ops):
    assert all(oracle_ops.get_alternate(n) == 0 for n in oracle_ops.ALTERNATES)
    with pytest.raises(KeyError):
        oracle_ops.set_alternate("no_such_reading", 1)


def test_nms_tie_order(oracle_ops, alt):
    # two overlapping boxes with EQUAL scores: the default keeps the lower index, the alternate the higher one
    boxes = np.array([[0.1, 0.1, 0.5, 0.5], [0.12, 0.1, 0.5, 0.5]], np.float32)
    scores = np.array([0.9, 0.9], np.float32)
    assert oracle_ops.nms(boxes, scores, 10, 0.5, 0.1).tolist() == [0]
    alt("nms_tie", 1)
    assert oracle_ops.nms(boxes, scores, 10, 0.5, 0.1).tolist() == [1]


def test_fast_exp_is_within_an_ulp_or_two(oracle_ops, alt):
    xs = np.linspace(-12, 12, 4001).astype(np.float32)
    ref = oracle_ops.sigmoid(xs)
    alt("fast_exp", 1)
    got = oracle_ops.sigmoid(xs)
    assert (ref != got).any()                      # it IS another function ...
    assert np.abs(ref - got).max() <= 2.5e-7       # ... a couple of ulp of a value in (0, 1) away


def test_round_half_up_only_differs_on_exact_halves(oracle_ops, alt):
    d0, s0 = oracle_ops.resize_dims(256, 257, 640, 128)      # 257 * 2.5 = 642.5
    d1, _ = oracle_ops.resize_dims(427, 640, 640, 128)
    alt("round", 1)
    e0, t0 = oracle_ops.resize_dims(256, 257, 640, 128)
    e1, _ = oracle_ops.resize_dims(427, 640, 640, 128)
    assert list(d0)[:2] == [640, 642] and list(e0)[:2] == [640, 643] and list(d1) == list(e1)


def test_resize_index_rules(oracle_ops, alt):
    img = np.arange(10, dtype=np.float32).reshape(1, 1, 10, 1).repeat(3, axis=3)      # one row, 10 columns -> 4 columns
    dims = (1, 4, 0, 0)
    assert oracle_ops.resize_pad(img, dims)[0, 0, :, 0].tolist() == [0.0, 2.0, 5.0, 7.0]          # floor(x * 2.5)
    alt("resize", 1)
    assert oracle_ops.resize_pad(img, dims)[0, 0, :, 0].tolist() == [1.0, 3.0, 6.0, 8.0]          # floor((x + 0.5) * 2.5)
    alt("resize", 2)
    assert oracle_ops.resize_pad(img, dims)[0, 0, :, 0].tolist() == [0.0, 3.0, 6.0, 9.0]          # round(x * 3)


def test_bn_form_agrees_to_rounding(oracle_ops, alt):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 5, 5, 16)).astype(np.float32)
    g, b, m, v = [rng.random(16).astype(np.float32) + 0.5 for _ in range(4)]
    ref = oracle_ops.bn_act(x.copy(), g, b, m, v, "relu")
    alt("bn_form", 1)
    got = oracle_ops.bn_act(x.copy(), g, b, m, v, "relu")
    assert np.abs(ref - got).max() <= 1e-6 and (ref != got).any()

"""-m gpu: precision mode f16x3 (include/ssd_hip.h SSD_PRECISION_F16X3) -- the FPN / head
convolutions on split-fp16 operands (x = h + l, products xh*wh + xh*wl + xl*wh on the fp16
matrix cores, fp32 accumulation).  The summation order differs from the oracle's fmaf chain,
so these tests bound the difference instead of asserting bit equality: the north-star
tolerance (1e-4) for the graph outputs with identical labels / num_boxes, and a much tighter
bound (2e-5 of the tensor's scale, the size of fp32 accumulation noise at K = 2304) per
convolution."""
import os
import sys

import numpy as np
import pytest

from test_gpu_stages import CONV_CASES, bn_params, dev
from test_gpu_forward import STAGES, compare_outputs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-4          # BASELINE.json north_star: fp32 scores / coords within 1e-4
CONV_TOL = 2e-5     # one convolution, relative to max(1, max |ref|)


@pytest.mark.parametrize("tile", ["128", "64"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(i) for i in range(len(CONV_CASES))])
def test_conv2d_f16x3(cuda, ssd, oracle_ops, case, tile, monkeypatch):
    monkeypatch.setenv("SSD_IGEMM_TILE", tile)
    B, H, W, Cin, Cout, k, stride, mode, use_bn, use_bias, act, use_up = case
    rng = np.random.default_rng(100 + CONV_CASES.index(case))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float32)
    ref = oracle_ops.conv2d(x, w, stride, mode)
    bn = None
    if use_bn:
        g, b, m, v = bn_params(rng, Cout)
        ref = oracle_ops.bn_act(ref, g, b, m, v, None)
        bn = (m, oracle_ops.bn_scale(g, v), b)
    bias = None
    if use_bias:
        bias = rng.standard_normal(Cout).astype(np.float32)
        ref = oracle_ops.bias_add(ref, bias)
    up = None
    if use_up:
        coarse = rng.standard_normal((B, ref.shape[1] // 2, ref.shape[2] // 2, Cout)).astype(np.float32)
        ref = oracle_ops.upsample2_add(coarse, ref)
        up = dev(cuda, coarse)
    if act == "relu":
        ref = np.maximum(ref, 0)
    elif act == "relu6":
        ref = np.clip(ref, 0, 6)
    got = ssd.ssd.conv2d(dev(cuda, x), w, stride, mode, bn=bn, bias=bias, up=up, act=act,
                         precision="f16x3").cpu().numpy()
    assert got.shape == ref.shape
    err = float(np.abs(got - ref).max())
    scale = max(1.0, float(np.abs(ref).max()))
    print("conv2d f16x3 %s: max abs err %.3g (scale %.3g)" % (case, err, scale))
    assert err <= CONV_TOL * scale


def test_conv2d_f16x3_small_and_large_magnitudes(cuda, ssd, oracle_ops):
    """Operands far from 1: weights ~1e-6 (the power-of-two weight scale keeps their low halves
    normal), activations ~1e-4 (low halves subnormal: honoured by the matrix cores) and ~1e3.
    Activation rows carry an ABSOLUTE resolution of 2^-25 (3e-8) and a range of +-65504."""
    rng = np.random.default_rng(3)
    for xs, ws in ((1.0, 1e-6), (1e-4, 1.0), (1e3, 1e-2), (1e-3, 1e3)):
        x = (rng.standard_normal((1, 9, 11, 64)) * xs).astype(np.float32)
        w = (rng.standard_normal((3, 3, 64, 40)) * ws / 24.0).astype(np.float32)
        ref = oracle_ops.conv2d(x, w, 1, "SAME")
        got = ssd.ssd.conv2d(dev(cuda, x), w, 1, "SAME", precision="f16x3").cpu().numpy()
        err = float(np.abs(got - ref).max())
        scale = float(np.abs(ref).max())
        print("x~%g w~%g: max abs err %.3g of scale %.3g" % (xs, ws, err, scale))
        # split-fp16 rows resolve 2^-25 absolutely (the low half's subnormal step is 2^-24): for tiny
        # activations that input rounding, times the weights' norm (~1 here), is what is left
        assert err <= CONV_TOL * scale + 4 * 2.0 ** -25


def test_conv2d_f16x3_overflow_is_reported(cuda, ssd):
    """Values beyond the fp16 range cannot be carried as h + l: the call fails loudly."""
    x = cuda.full((1, 4, 4, 32), 300.0, dtype=cuda.float32, device="cuda")
    w = np.full((1, 1, 32, 32), 10.0, np.float32)       # outputs 96 000 > 65 504
    with pytest.raises(ssd.SsdError, match="fp16 range"):
        ssd.ssd.conv2d(x, w, 1, "SAME", precision="f16x3")
    x = cuda.full((1, 4, 4, 32), 1e5, dtype=cuda.float32, device="cuda")   # input itself out of range: clamped silently
    w = np.full((1, 1, 32, 32), 1e-3, np.float32)                           # by the test conversion, outputs fine
    ssd.ssd.conv2d(x, w, 1, "SAME", precision="f16x3")


def stage_errors(engine, keep, what):
    worst = 0.0
    for n in STAGES:
        got = engine.get_tensor(n)
        ref = keep[n].reshape(got.shape)
        err = float(np.abs(got - ref).max())
        scale = max(1.0, float(np.abs(ref).max()))
        print("%s %s: max err %.3g scale %.3g" % (what, n, err, scale))
        assert err <= TOL * scale, (what, n)
        worst = max(worst, err / scale)
    return worst


@pytest.mark.parametrize("backbone,H,W,B", [("mobilenet", 128, 256, 9), ("shufflenet", 128, 128, 2)])
def test_forward_f16x3_small(cuda, ssd, oracle_graph, backbone, H, W, B):
    params = {"backbone": backbone, "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 128}
    Wt = ssd.synthetic_weights(params, seed=11, logits_bias=-4.0)
    img = np.random.default_rng(5).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    keep = {}
    ref = oracle_graph.forward(img, Wt, params, keep)
    eng = ssd.Engine(params, Wt, precision="f16x3")
    assert eng.precision == "f16x3"
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    stage_errors(eng, keep, backbone + " f16x3")
    compare_outputs(out, ref, backbone + " small f16x3")
    assert ref["num_boxes"].min() > 0
    assert eng.status() == 0
    # the same engine switched to f32 gives the oracle's bits, and back
    eng.set_precision("f32")
    out32 = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    assert np.array_equal(out32[2], ref["scores"]) and np.array_equal(out32[0], ref["boxes"])
    eng.set_precision("f16x3")
    again = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    for a, b in zip(out, again):
        assert np.array_equal(a, b)
    eng.close()


@pytest.mark.parametrize("cfg,H,W", [("config_mobilenet.json", 640, 896), ("config_shufflenet.json", 640, 640)])
def test_forward_f16x3_full_size(cuda, ssd, oracle_graph, cfg, H, W):
    params = ssd.load_config(os.path.join(HERE, "golden", cfg))
    Wt = ssd.synthetic_weights(params, seed=0, logits_bias=-4.0 if "mobile" in cfg else -9.0)
    img = np.random.default_rng(0).integers(0, 256, (1, H, W, 3), dtype=np.uint8)
    keep = {}
    ref = oracle_graph.forward(img, Wt, params, keep)
    eng = ssd.Engine(params, Wt, precision="f16x3")
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    stage_errors(eng, keep, "full f16x3")
    compare_outputs(out, ref, "full size f16x3")
    assert ref["num_boxes"][0] > 50
    assert eng.status() == 0
    eng.close()


def test_golden_tiny_f16x3(cuda, ssd):
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as mg
    Wt, img = mg.inputs()
    z = np.load(os.path.join(HERE, "golden", "tiny_mobilenet_128.npz"))
    eng = ssd.Engine(mg.TINY, Wt, precision="f16x3")
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    compare_outputs(out, z, "golden tiny f16x3")
    assert np.abs(eng.get_tensor("encoded_boxes").reshape(2, -1, 4) - z["encoded_boxes"]).max() <= TOL
    eng.close()

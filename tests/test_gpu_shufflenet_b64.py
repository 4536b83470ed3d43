"""-m gpu: BASELINE config 4 -- ShuffleNet-v2 + FPN (config_shufflenet.json), 640x640, batch 64, one MI355X --
at its full size, in both precision modes.  (Round 1 found a memory fault at exactly this size with a scratch
script; this is the committed form.)  Size-independent properties at B = 64 (permutation, batch independence),
equality with the CPU oracle on two images of the batch, and a saturated-logits run (every score a candidate)."""
import os

import numpy as np
import pytest

from test_gpu_forward import STAGES, compare_outputs as compare_exact
from test_gpu_f16x3 import _same_within_tolerance, compare_outputs as compare_tol, TOL

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
B, H, W = 64, 640, 640


def _setup(ssd, bias):
    params = ssd.load_config(os.path.join(HERE, "golden", "config_shufflenet.json"))
    Wt = ssd.synthetic_weights(params, seed=0, logits_bias=bias)
    imgs = np.random.default_rng(64).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    return params, Wt, imgs


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_shufflenet_640_batch64(cuda, ssd, oracle_graph, precision):
    params, Wt, imgs = _setup(ssd, -4.0)
    eng = ssd.Engine(params, Wt, precision=precision)
    d = cuda.from_numpy(imgs).cuda()
    full = [t.cpu().numpy() for t in eng.forward(d)]
    assert full[0].shape == (B, 2000, 4) and full[3].min() > 0
    # permuting the batch permutes the outputs: no cross-image term, and no dependence on which tile holds an image
    perm = np.random.default_rng(1).permutation(B)
    permd = [t.cpu().numpy() for t in eng.forward(d[cuda.from_numpy(perm).cuda()].contiguous())]
    for a, b in zip(full, permd):
        assert np.array_equal(a[perm], b)
    # the CPU oracle on eight images of the batch in mode f32 (four per half-batch backbone chain, first and last of each), two in
    # the opt-in mode
    pick = [0, 9, 20, 31, 32, 41, 52, B - 1] if precision == "f32" else [0, B - 1]
    keep = {}
    ref = oracle_graph.forward(imgs[pick], Wt, params, keep)
    got = [t[pick] for t in full]
    if precision == "f32":
        compare_exact(got, ref, "shufflenet B=64 images 0/9/20/31/32/41/52/63 (f32)")
        assert np.array_equal(got[0], ref["boxes"]) and np.array_equal(got[2], ref["scores"])       # bit-identical
        # every image of the batch = its own batch-1 run (other tile shapes, other kernels' launch geometry)
        one = [t.cpu().numpy() for t in eng.forward(d[17:18].contiguous())]
        for a, b in zip(full, one):
            assert np.array_equal(a[17:18], b)
    else:
        compare_tol(got, ref, "shufflenet B=64 images 0/63 (f16x3)", keep, oracle_graph.ops, (H, W))
        assert eng.status() == 0
        one = [t.cpu().numpy() for t in eng.forward(d[17:18].contiguous())]
        _same_within_tolerance(one, [t[17:18] for t in full], "shufflenet image 17 alone vs in the batch (f16x3)")
    eng.close()


def test_shufflenet_640_batch64_stages_vs_oracle(cuda, ssd, oracle_graph):
    """Every retained stage of the B = 64 forward, on the eight images the oracle also ran: bit-identical in mode f32."""
    params, Wt, imgs = _setup(ssd, -4.0)
    eng = ssd.Engine(params, Wt, precision="f32")
    eng.forward(cuda.from_numpy(imgs).cuda())
    keep = {}
    pick = [0, 9, 20, 31, 32, 41, 52, B - 1]        # four per half-batch backbone chain
    oracle_graph.forward(imgs[pick], Wt, params, keep)
    for n in STAGES:
        got = eng.get_tensor(n)[pick]
        ref = keep[n].reshape(got.shape)
        assert np.array_equal(got, ref), n
    eng.close()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_shufflenet_640_batch64_saturated_logits(cuda, ssd, precision):
    """Logits bias +1: every (anchor, class) is a candidate (51 150 x 80 per image), every NMS list is far longer
    than the register capacity, the fused scan's bitmap is all ones -- at config 4's full size (the fault fixed in
    b6501d5 was a queue overrun in exactly this regime).  f16x3 against f32; f32 against its own permuted run."""
    params, Wt, imgs = _setup(ssd, 1.0)
    d = cuda.from_numpy(imgs).cuda()
    e32 = ssd.Engine(params, Wt, precision="f32")
    ref = [t.cpu().numpy() for t in e32.forward(d)]
    assert (ref[3] == 2000).all()
    if precision == "f32":
        perm = np.random.default_rng(2).permutation(B)
        permd = [t.cpu().numpy() for t in e32.forward(d[cuda.from_numpy(perm).cuda()].contiguous())]
        for a, b in zip(ref, permd):
            assert np.array_equal(a[perm], b)
        e32.close()
        return
    e32.close()
    e16 = ssd.Engine(params, Wt, precision="f16x3")
    out = [t.cpu().numpy() for t in e16.forward(d)]
    assert e16.status() == 0
    _same_within_tolerance(out, ref, "shufflenet B=64 saturated logits")
    e16.close()


def test_shufflenet_forward_has_no_shuffle_or_concat_launches(cuda, ssd):
    """SURVEY 8(f) row 1, second half: concat_shuffle_split (shufflenet_v2.py:94-115) runs as no kernel of its own: every
    producer of a stage stores its channels dense and conv1x1_before gathers its input rows through a source table
    (sn_pw.hip); the stage output is kept in two-part rows, its x half written by the last unit, its y half by ONE row
    gather per stage.  Profile class `other` at this batch: the FPN's top-down merge + those three gathers -- no launch per
    unit -- and every depthwise layer runs inside the fused kernel (class `depthwise` empty: 19 depthwise layers in 19
    fused launches)."""
    params = ssd.load_config(os.path.join(HERE, "golden", "config_shufflenet.json"))
    Wt = ssd.synthetic_weights(params, seed=0, logits_bias=-4.0)
    eng = ssd.Engine(params, Wt)
    img = cuda.from_numpy(np.random.default_rng(3).integers(0, 256, (2, 256, 256, 3), dtype=np.uint8)).cuda()
    eng.forward(img)
    eng.profile_reset()
    eng.profile_enable(True)
    eng.forward(img)
    eng.profile_enable(False)
    prof = eng.profile_read()
    assert prof["other"]["launches"] == 1 + 3, prof["other"]
    assert prof["depthwise"]["launches"] == 0 and prof["depthwise_pointwise_fused"]["launches"] == 19, prof
    eng.close()


def _scaled(Wt, name, factor):
    out = dict(Wt)
    out[name] = Wt[name] * np.float32(factor)
    return out


def test_f16x3_lateral_input_overflow_is_reported(cuda, ssd):
    """ShuffleNet's c3 is an unbounded ReLU output that the FPN lateral reads as fp32 rows and splits into
    halves while staging (in_fmt 2).  A value beyond +-65504 there used to be clamped silently.  c3 is blown up by
    1e6 through the batch-norm scale of the stage's last 1x1; Stage3's entry convolutions and lateral3 are scaled down
    by the same factor so that NOTHING else leaves the range: only the staging check can set the status bit."""
    params = {"backbone": "shufflenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 256}
    Wt = ssd.synthetic_weights(params, seed=5, logits_bias=-6.0)
    img = cuda.from_numpy(np.random.default_rng(3).integers(0, 256, (2, 256, 256, 3), dtype=np.uint8)).cuda()
    eng = ssd.Engine(params, Wt, precision="f16x3")
    eng.forward(img)
    assert eng.status() == 0
    c3max = float(np.abs(eng.get_tensor("c3")).max())
    eng.close()
    k = np.float32(2.0 ** 20)
    big = dict(Wt)
    for u in ("unit_4/conv1x1_after", ):
        big["ShuffleNetV2/Stage2/%s/batch_norm/gamma" % u] = Wt["ShuffleNetV2/Stage2/%s/batch_norm/gamma" % u] * k
        big["ShuffleNetV2/Stage2/%s/batch_norm/beta" % u] = Wt["ShuffleNetV2/Stage2/%s/batch_norm/beta" % u] * k
    for n in ("ShuffleNetV2/Stage3/unit_1/conv1x1_before/weights", "ShuffleNetV2/Stage3/unit_1/second_branch/depthwise/depthwise_weights",
              "fpn/lateral3/kernel"):
        big[n] = Wt[n] / k
    eng = ssd.Engine(params, big, precision="f16x3")
    eng.forward(img)
    c3 = eng.get_tensor("c3")
    assert float(np.abs(c3).max()) > 65504.0 > c3max
    for n in ("c4", "c5", "p3", "p4", "p5"):
        assert float(np.abs(eng.get_tensor(n)).max()) < 65504.0, n
    assert eng.status() & 1, "a staged fp32 activation beyond the fp16 range was clamped without setting the status bit"
    eng.close()


def test_f16x3_nan_is_reported_by_the_large_tile_kernel(cuda, ssd, libopt):
    """igemm16's range check is a running fmax, which drops NaN operands; a NaN result must still fail
    loudly (batch-norm beta = NaN, no activation: the value reaches the S16 store as NaN or as -inf after the clamps)."""
    libopt(igemm16=1)
    x = cuda.ones((1, 8, 8, 256), dtype=cuda.float32, device="cuda")
    w = np.full((1, 1, 256, 256), 0.01, np.float32)
    ones, zeros = np.ones(256, np.float32), np.zeros(256, np.float32)
    beta = zeros.copy()
    beta[7] = np.nan
    with pytest.raises(ssd.SsdError, match="fp16 range"):
        ssd.ssd.conv2d(x, w, 1, "SAME", bn=(zeros, ones, beta), act=None, precision="f16x3")
    libopt(igemm16=0)          # the 128x128 kernel's S16 epilogue
    with pytest.raises(ssd.SsdError, match="fp16 range"):
        ssd.ssd.conv2d(x, w, 1, "SAME", bn=(zeros, ones, beta), act=None, precision="f16x3")

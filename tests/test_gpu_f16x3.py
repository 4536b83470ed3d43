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
from test_gpu_forward import STAGES

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-4          # BASELINE.json north_star: fp32 scores / coords within 1e-4
CONV_TOL = 2e-5     # one convolution, relative to max(1, max |ref|)


@pytest.mark.parametrize("tile", ["128", "64", "256"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(i) for i in range(len(CONV_CASES))])
def test_conv2d_f16x3(cuda, ssd, oracle_ops, case, tile, libopt):
    # "256": the one-block-per-CU 256x256-tile kernel (igemm16.hip) wherever its form applies (batch norm,
    # output width a multiple of 256); the library otherwise keeps it for launches with >= 512 tiles
    if tile == "256":
        libopt(igemm16=1)
    else:
        libopt(igemm16=0)
        libopt(igemm_tile=int(tile, 0))
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


@pytest.mark.parametrize("shape", [(3, 40, 56, 256, 256, 3, 1), (2, 20, 28, 1024, 256, 3, 2), (1, 80, 112, 256, 512, 1, 1),
                                   (5, 17, 13, 96, 256, 1, 1)])
def test_conv2d_f16x3_large_tiles(cuda, ssd, oracle_ops, shape, libopt):
    """igemm16.hip on shapes with several 256-row tiles, ragged last tiles, two column tiles, the
    explicit-pad stride-2 form (fpn p6) and the shortest K loop it accepts (3 K-steps)."""
    libopt(igemm16=1)
    B, H, W, Cin, Cout, k, stride = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float32)
    g, b, m, v = bn_params(rng, Cout)
    mode = "SAME" if stride == 1 else "EXPLICIT"
    ref = np.maximum(oracle_ops.bn_act(oracle_ops.conv2d(x, w, stride, mode), g, b, m, v, None), 0)
    got = ssd.ssd.conv2d(dev(cuda, x), w, stride, mode, bn=(m, oracle_ops.bn_scale(g, v), b), act="relu",
                         precision="f16x3").cpu().numpy()
    err = float(np.abs(got - ref).max())
    scale = max(1.0, float(np.abs(ref).max()))
    print("igemm16 %s: max abs err %.3g (scale %.3g)" % (shape, err, scale))
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


def compare_outputs(got, ref, what, keep=None, ops=None, hw=None):
    """Graph outputs against the oracle: num_boxes and labels identical (class-major layout, so the
    per-class counts are identical too), scores and boxes within the north-star tolerance.
    Candidates of one class whose ORACLE scores are closer than the tolerance (exact fp32 ties occur:
    random-init heads give pairs of anchors with identical scores) are ordered -- and, where they overlap,
    one of them suppressed -- by the oracle's tie rule "lower anchor index first"; any arithmetic that is
    not bit-identical decides such ties by its own last bits, and TF 1.12 itself by a priority-queue
    artefact.  So a slot whose box differs must hold (i) another oracle detection of the
    same class out of the same run of near-equal scores, or (ii) the decoded box of another candidate
    anchor of that class whose oracle score is the score the slot reports (needs keep / ops / hw): a tie
    decided the other way can change which later candidates of the class survive, and with the per-class cap
    (25) reached the count stays the same.  The exact statement about the NMS itself is
    check_postprocess_of_own_heads."""
    gb, gl, gs, gn = got
    rb, rl, rs, rn = ref["boxes"], ref["labels"], ref["scores"], ref["num_boxes"]
    assert np.array_equal(gn, rn), (what, gn, rn)
    assert np.array_equal(gl, rl), what + ": labels"
    assert np.abs(gs - rs).max() <= TOL, (what, float(np.abs(gs - rs).max()))
    swapped = other = 0
    anc = ops.anchors(*hw) if ops is not None else None
    for b in range(gb.shape[0]):
        n = int(rn[b])
        d = np.abs(gb[b, :n] - rb[b, :n]).max(axis=1) if n else np.zeros(0)
        dec = None
        for i in np.nonzero(d > TOL)[0]:
            cand = np.nonzero((rl[b, :n] == rl[b, i]) & (np.abs(rs[b, :n] - rs[b, i]) <= TOL))[0]
            if any(np.abs(gb[b, i] - rb[b, j]).max() <= TOL for j in cand):
                swapped += 1
                continue
            assert keep is not None, (what, "image %d slot %d matches no oracle detection of its class and score" % (b, i))
            if dec is None:
                dec = ops.decode_clip(keep["encoded_boxes"].reshape(gb.shape[0], -1, 4)[b], anc)
            logit = keep["class_predictions"].reshape(gb.shape[0], dec.shape[0], -1)[b][:, rl[b, i]]
            near = np.nonzero(np.abs(logit - np.log(gs[b, i] / (1.0 - gs[b, i]))) <= 1e-2)[0]
            sc = ops.sigmoid(logit[near])
            tied = near[np.abs(sc - gs[b, i]) <= TOL]      # oracle score of the anchor ~ the score this slot reports
            if any(np.abs(gb[b, i] - dec[a]).max() <= TOL for a in tied):
                other += 1
                continue
            # (iii) the oracle's own anchor, but an ill-conditioned decode: h = exp(th / 5) * ha amplifies the
            # code's rounding noise when the unclipped box is many times the image (random-init ShuffleNet heads
            # produce such codes): allow the tolerance times the unclipped extent
            a = int(np.argmin(np.abs(dec - rb[b, i]).max(axis=1)))
            codes = keep["encoded_boxes"].reshape(gb.shape[0], -1, 4)[b][a].astype(np.float64)
            ha, wa = float(anc[a, 2] - anc[a, 0]), float(anc[a, 3] - anc[a, 1])
            extent = max(1.0, np.exp(codes[2] / 5.0) * ha, np.exp(codes[3] / 5.0) * wa, abs(codes[0]) / 10.0 * ha, abs(codes[1]) / 10.0 * wa)
            assert np.abs(dec[a] - rb[b, i]).max() == 0.0 and np.abs(gb[b, i] - rb[b, i]).max() <= TOL * extent, \
                (what, "image %d slot %d: box differs by %.3g (unclipped extent %.3g)" % (b, i, float(d[i]), extent))
            other += 1
        assert np.abs(gb[b, n:]).max(initial=0.0) == 0.0
    total = int(rn.sum())
    print(what, "num", gn.tolist(), "max score err %.3g;" % float(np.abs(gs - rs).max()),
          "of %d slots %d hold a tying detection in another order, %d another tying candidate or an ill-conditioned decode" % (total, swapped, other))
    assert swapped + other <= max(8, total // 500), (what, swapped, other)


def check_postprocess_of_own_heads(engine, got, params, hw, ops):
    """Exact, order-sensitive half of the parity argument: the oracle's decode + per-class NMS applied to
    the heads THIS forward produced reproduces its detections bit for bit (post-processing is the same
    exact code in both precision modes; only the convolutions' summation order differs)."""
    gb, gl, gs, gn = got
    B = gb.shape[0]
    C = params["num_classes"]
    logits = engine.get_tensor("class_predictions").reshape(B, -1, C)
    codes = engine.get_tensor("encoded_boxes").reshape(B, -1, 4)
    b, l, s, n = ops.postprocess(logits, codes, ops.anchors(*hw), params["score_threshold"], params["iou_threshold"],
                                 params["max_boxes_per_class"])
    assert np.array_equal(n, gn) and np.array_equal(l, gl)
    assert np.array_equal(s, gs) and np.array_equal(b, gb)


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
    check_postprocess_of_own_heads(eng, out, params, (H, W), oracle_graph.ops)
    compare_outputs(out, ref, backbone + " small f16x3", keep, oracle_graph.ops, (H, W))
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
    Wt = ssd.synthetic_weights(params, seed=0, logits_bias=-4.0)
    img = np.random.default_rng(0).integers(0, 256, (1, H, W, 3), dtype=np.uint8)
    keep = {}
    ref = oracle_graph.forward(img, Wt, params, keep)
    eng = ssd.Engine(params, Wt, precision="f16x3")
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    stage_errors(eng, keep, "full f16x3")
    check_postprocess_of_own_heads(eng, out, params, (H, W), oracle_graph.ops)
    compare_outputs(out, ref, "full size f16x3", keep, oracle_graph.ops, (H, W))
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


def test_detector_f16x3_and_overflow_fallback(cuda, ssd, oracle_graph):
    """Detector(precision="f16x3"): the drop-in API in the fast mode; and the safety net -- weights that push
    an FPN activation beyond the fp16 range make the status word non-zero, the Detector warns, switches to
    f32 and returns exactly what an f32 detector returns."""
    params = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 128}
    Wt = ssd.synthetic_weights(params, seed=3, logits_bias=-4.0)
    img = np.random.default_rng(9).integers(0, 256, (128, 128, 3), dtype=np.uint8)
    det = ssd.Detector(Wt, config=params, precision="f16x3")
    boxes, labels, scores = det(img, score_threshold=0.2)
    rb, rl, rs = oracle_graph.detector_call(oracle_graph.forward(img[None], Wt, ssd.load_config(params)), 0.2)
    assert np.array_equal(labels, rl) and len(labels) > 0
    assert np.abs(scores - rs).max() <= TOL and np.abs(boxes - rb).max() <= TOL
    assert det.engine.precision == "f16x3"
    # x5 = lateral5(c5) blown up by 1e6: |x5| > 65504
    big = dict(Wt)
    big["fpn/lateral5/kernel"] = Wt["fpn/lateral5/kernel"] * np.float32(1e6)
    det2 = ssd.Detector(big, config=params, precision="f16x3")
    with pytest.warns(UserWarning, match="fp16 range"):
        b2, l2, s2 = det2(img, score_threshold=0.2)
    assert det2.engine.precision == "f32"
    det3 = ssd.Detector(big, config=params, precision="f32")
    b3, l3, s3 = det3(img, score_threshold=0.2)
    assert np.array_equal(b2, b3) and np.array_equal(l2, l3) and np.array_equal(s2, s3)
    # the engine API reports the same condition without acting on it
    eng = ssd.Engine(params, big, precision="f16x3")
    eng.forward(cuda.from_numpy(img[None].copy()).cuda())
    assert eng.status() == 1 and eng.status() == 0      # read-and-clear
    eng.close()


def _same_within_tolerance(a, b, what):
    """f16x3 engine outputs against the f32 engine's (which other tests pin bit for bit to the oracle):
    identical num_boxes / labels, scores within TOL; boxes slot by slot, except a few slots where exactly
    tying candidates were ordered the other way (see compare_outputs)."""
    ab, al, as_, an = a
    bb, bl, bs, bn = b
    assert np.array_equal(an, bn), (what, an, bn)
    assert np.array_equal(al, bl), what
    assert np.abs(as_ - bs).max() <= TOL, (what, float(np.abs(as_ - bs).max()))
    d = np.abs(ab - bb).max(axis=2)
    bad = int((d > TOL).sum())
    total = int(bn.sum())
    print("%s: %d detections, max score diff %.3g, slots with another box %d" % (what, total, float(np.abs(as_ - bs).max()), bad))
    assert bad <= max(8, total // 500), (what, bad, total)


@pytest.mark.parametrize("backbone,B,H,W,env", [
    ("mobilenet", 40, 256, 384, {}),                       # 40 images: the 256x256-tile kernel carries towers, p3 and logits
    ("mobilenet", 40, 256, 384, {"igemm16": 0}),     # the same on the 128x128 kernel's S16 path
    ("mobilenet", 5, 256, 128, {"nsub": 3}),         # consecutive sub-batch plans
    ("mobilenet", 2, 300, 500, {}),                        # resize_keeping_aspect_ratio path (min_dimension 256): reduced in width, two launches
    ("mobilenet", 5, 200, 300, {}),                        # ... enlarged: the fused first-layers launch with the gather (front.hip), two chains
    ("shufflenet", 2, 200, 300, {}),
    ("mobilenet", 1, 256, 256, {"streams": 1}),      # every launch on the caller's stream
    ("mobilenet", 8, 256, 256, {"backbone_split": 4}),   # four backbone chains on four streams
    ("shufflenet", 6, 256, 256, {}),
])
def test_f16x3_against_f32_engine(cuda, ssd, libopt, backbone, B, H, W, env):
    libopt(**env)
    params = {"backbone": backbone, "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 256}
    Wt = ssd.synthetic_weights(params, seed=17, logits_bias=-6.0)
    img = np.random.default_rng(B + H).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    e32 = ssd.Engine(params, Wt, precision="f32")
    ref = [t.cpu().numpy() for t in e32.forward(cuda.from_numpy(img).cuda())]
    e32.close()
    e16 = ssd.Engine(params, Wt, precision="f16x3")
    out = None
    for rep in range(2):
        out = [t.cpu().numpy() for t in e16.forward_cached(img)]
    assert e16.status() == 0
    assert ref[3].sum() > 0
    _same_within_tolerance(out, ref, "%s B=%d %dx%d %s" % (backbone, B, H, W, env))
    e16.close()


def test_igemm16_repeatable(cuda, ssd, libopt):
    """Race screen for the LDS-DMA pipeline of igemm16.hip (the DMA of a stage is ordered for its readers only
    by the issuing wave's vmcnt wait plus a barrier): the same launch repeated must give the same bits, on a
    shape with many 256-row tiles per CU so that blocks start and finish at different phases."""
    libopt(igemm16=1)
    rng = np.random.default_rng(77)
    x = dev(cuda, rng.standard_normal((24, 40, 56, 256)).astype(np.float32))
    w = (rng.standard_normal((3, 3, 256, 256)) * np.sqrt(2.0 / 2304)).astype(np.float32)
    g, b, m, v = bn_params(rng, 256)
    bn = (m, g / np.sqrt(v + 1e-3), b)
    first = ssd.ssd.conv2d(x, w, 1, "SAME", bn=bn, act="relu", precision="f16x3")
    for _ in range(25):
        again = ssd.ssd.conv2d(x, w, 1, "SAME", bn=bn, act="relu", precision="f16x3")
        assert cuda.equal(first, again)
    # and the class-logits form (bias, fp32 rows, 480 of 512 columns)
    wl = (rng.standard_normal((3, 3, 256, 480)) * np.sqrt(2.0 / 2304)).astype(np.float32)
    bias = rng.standard_normal(480).astype(np.float32)
    first = ssd.ssd.conv2d(x, wl, 1, "SAME", bias=bias, precision="f16x3")
    for _ in range(10):
        assert cuda.equal(first, ssd.ssd.conv2d(x, wl, 1, "SAME", bias=bias, precision="f16x3"))


@pytest.mark.parametrize("backbone,dm,classes,H,W", [("mobilenet", 0.5, 20, 128, 256), ("mobilenet", 0.75, 3, 256, 128),
                                                     ("shufflenet", 0.5, 20, 128, 128), ("shufflenet", 1.5, 80, 128, 256)])
def test_f16x3_other_widths_and_class_counts(cuda, ssd, backbone, dm, classes, H, W):
    """depth_multiplier / num_classes variants in mode f16x3: padded channel counts (24 -> 32, 88 -> 96 ...), head
    widths 18 / 120 / 480 (rows that are not 16-byte aligned take the per-element store path) against the f32 engine."""
    params = {"backbone": backbone, "depth_multiplier": dm, "num_classes": classes, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 128}
    Wt = ssd.synthetic_weights(params, seed=21, logits_bias=-4.0)
    img = np.random.default_rng(7).integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    e32 = ssd.Engine(params, Wt, precision="f32")
    ref = [t.cpu().numpy() for t in e32.forward(cuda.from_numpy(img).cuda())]
    e32.close()
    e16 = ssd.Engine(params, Wt, precision="f16x3")
    out = [t.cpu().numpy() for t in e16.forward(cuda.from_numpy(img).cuda())]
    assert e16.status() == 0 and out[0].shape == (3, classes * 25, 4)
    _same_within_tolerance(out, ref, "%s x%.2f C=%d" % (backbone, dm, classes))
    e16.close()


def test_f16x3_dense_candidates(cuda, ssd):
    """Every logit above the threshold (bias +1): the candidate-octet bitmap of the fused score filter is all ones,
    the scan's LDS queue must be drained between rounds, and every (image, class) list is longer than the register
    capacity of the NMS kernels.  Against the f32 engine (whose scan reads the logits themselves)."""
    params = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 256}
    Wt = ssd.synthetic_weights(params, seed=23, logits_bias=1.0)
    img = np.random.default_rng(40).integers(0, 256, (40, 256, 384, 3), dtype=np.uint8)
    e32 = ssd.Engine(params, Wt, precision="f32")
    ref = [t.cpu().numpy() for t in e32.forward(cuda.from_numpy(img).cuda())]
    e32.close()
    e16 = ssd.Engine(params, Wt, precision="f16x3")
    out = [t.cpu().numpy() for t in e16.forward(cuda.from_numpy(img).cuda())]
    assert e16.status() == 0 and (ref[3] == 2000).all()
    _same_within_tolerance(out, ref, "dense candidates")
    e16.close()


def test_f16x3_batch_independence_full_size(cuda, ssd):
    """Images are independent end to end (nms.py:96-101), and an output element's arithmetic does not depend on which
    tile computes it: the first 32 images of a 70-image batch (two sub-batch plans: the 2 GiB rule) and of a 33-image
    batch give the bits of the 32-image batch, at the bench's size, in mode f16x3."""
    import bench
    Wt = ssd.synthetic_weights(bench.PARAMS, seed=0, logits_bias=bench.LOGITS_BIAS["mobilenet"])
    eng = ssd.Engine(bench.PARAMS, Wt, precision="f16x3")
    g = cuda.Generator().manual_seed(7)
    frames = cuda.randint(0, 256, (70, bench.H, bench.W, 3), dtype=cuda.uint8, generator=g).cuda()
    ref = [t.clone() for t in eng.forward(frames[:32].contiguous())]
    for B in (70, 33):
        out = eng.forward(frames[:B].contiguous())
        assert all(cuda.equal(a[:32], b) for a, b in zip(out, ref)), B
    assert eng.status() == 0 and float(ref[3].float().mean()) > 50
    eng.close()


def test_igemm16_overflow_is_reported(cuda, ssd, libopt):
    """The 256x256-tile kernel's own range check (a running maximum instead of per-value clamps)."""
    libopt(igemm16=1)
    x = cuda.full((1, 8, 8, 256), 100.0, dtype=cuda.float32, device="cuda")
    w = np.full((1, 1, 256, 256), 1.0, np.float32)                    # sums of 25 600
    ones, zeros = np.ones(256, np.float32), np.zeros(256, np.float32)
    ok = ssd.ssd.conv2d(x, w, 1, "SAME", bn=(zeros, ones, zeros), act="relu", precision="f16x3")
    assert float(ok.max()) == 25600.0
    with pytest.raises(ssd.SsdError, match="fp16 range"):
        ssd.ssd.conv2d(x, w, 1, "SAME", bn=(zeros, ones * 3.0, zeros), act="relu", precision="f16x3")   # 76 800

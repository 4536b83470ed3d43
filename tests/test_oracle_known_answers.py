"""Pins the CPU oracle (oracle/) against every known answer derivable from the reference's
source and against an independent convolution library (torch CPU).
The reference ships no tests or golden vectors, so this is all the pinning there is:
parity with TensorFlow itself stays UNPINNED (oracle/ssd_oracle.c header)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def tconv(x, w, stride=1, pad=(0, 0, 0, 0)):
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    wt = torch.from_numpy(w).permute(3, 2, 0, 1)
    return nhwc(F.conv2d(F.pad(xt, pad), wt, stride=stride))


# --------------------------------------------------------------------------- anchors
def test_anchor_counts(oracle_ops):
    # anchor_generator.py:59-62; SURVEY 8c "Counts"
    assert oracle_ops.anchors(640, 896).shape == (71610, 4)
    assert oracle_ops.anchors(640, 640).shape == (51150, 4)
    assert oracle_ops.anchors(128, 128).shape == (2046, 4)


def test_anchor_values_640x896(oracle_ops):
    a = oracle_ops.anchors(640, 896)
    exp = {
        0: [-0.01875, -0.01339286, 0.03125, 0.02232143],
        1: [-0.01142767, -0.02078953, 0.02392767, 0.02971810],
        2: [-0.02910534, -0.00816262, 0.04160534, 0.01709119],
        3: [-0.02910500, -0.02078928, 0.04160500, 0.02971786],
        6: [-0.01875, -0.00446429, 0.03125, 0.03125],
        53760: [-0.0375, -0.02678571, 0.0625, 0.04464286],
        71609: [0.10000768, 0.64285988, 1.69999242, 1.21428299],
    }
    for i, v in exp.items():
        np.testing.assert_allclose(a[i], v, rtol=0, atol=1e-6)
    assert abs(a.min() - (-0.699992)) < 1e-5 and abs(a.max() - 1.699992) < 1e-5
    # not clipped (anchor_generator.py:116-118): 10 444 anchors leave the unit square
    assert int(((a < 0) | (a > 1)).any(axis=1).sum()) == 10444


def test_anchor_layout_offsets(oracle_ops):
    # order [y][x][a], levels concatenated 3..7 (anchor_generator.py:157-169, :110)
    a = oracle_ops.anchors(640, 896) * np.array([640, 896, 640, 896], np.float32)
    offs = [0, 53760, 67200, 70560, 71400]
    for lvl, (off, s) in enumerate(zip(offs, [8, 16, 32, 64, 128])):
        w = 896 // s
        for (y, x) in [(0, 0), (1, 2), (640 // s - 1, w - 1)]:
            i = off + (y * w + x) * 6
            cy = (a[i, 0] + a[i, 2]) / 2
            cx = (a[i, 1] + a[i, 3]) / 2
            assert abs(cy - (y * s + s / 2)) < 1e-3 and abs(cx - (x * s + s / 2)) < 1e-3
            assert abs((a[i, 2] - a[i, 0]) - 32 * 2 ** lvl) < 1e-2     # pair (1.0, 1.0)


def _tile_anchors_numpy(H, W, strides, scales, mults, ars):
    """anchor_generator.py:40-170 written as the TF graph is -- whole-tensor fp32 operations, tile_anchors' meshgrid /
    stack / tile / concat / reshape -- in numpy: a restatement that shares no loop with the oracle's."""
    import itertools
    f = np.float32
    ih, iw = f(H), f(W)
    pairs = list(itertools.product(mults, ars))
    ratios = np.array([a for _, a in pairs], f)
    out = []
    for i, s in enumerate(strides):
        st = f(s)
        h, w = int(np.ceil(ih / st)), int(np.ceil(iw / st))
        sc = np.array([m * scales[i] for m, _ in pairs], f)
        oy = f(0.5) * (ih - (f(h) - f(1.0)) * st)
        ox = f(0.5) * (iw - (f(w) - f(1.0)) * st)
        rs = np.sqrt(ratios)
        heights, widths = sc / rs, sc * rs
        yc = np.arange(h).astype(f) * st + oy
        xc = np.arange(w).astype(f) * st + ox
        xg, yg = np.meshgrid(xc, yc)
        centers = np.tile(np.stack([yg, xg], axis=2)[:, :, None, :], (1, 1, len(pairs), 1))
        sizes = np.tile(np.stack([heights, widths], axis=1)[None, None], (h, w, 1, 1))
        out.append(np.concatenate([centers - f(0.5) * sizes, centers + f(0.5) * sizes], axis=3).reshape(-1, 4))
    return (np.concatenate(out, 0) / np.array([ih, iw, ih, iw], f)).astype(f)


def test_anchor_generator_any_hyper_parameters(oracle_ops):
    # AnchorGenerator.__init__ takes any strides / scales / multipliers / ratios (anchor_generator.py:13-38)
    default = ([8, 16, 32, 64, 128], [32, 64, 128, 256, 512], [1.0, 1.4142], [1.0, 2.0, 0.5])
    assert np.array_equal(oracle_ops.anchors_ex(640, 896, *default), oracle_ops.anchors(640, 896))
    assert np.array_equal(oracle_ops.anchors_ex(256, 384, *default), _tile_anchors_numpy(256, 384, *default))
    other = ([16, 32, 64], [40, 96.5, 200], [1.0, 1.26, 1.5874], [1.0, 3.0, 1.0 / 3.0, 0.5])
    for H, W in [(256, 384), (250, 330)]:          # the second: sizes the strides do not divide (ceil, offsets < stride / 2)
        a = oracle_ops.anchors_ex(H, W, *other)
        assert a.shape == (sum(-(-H // s) * -(-W // s) for s in other[0]) * 12, 4)
        assert np.array_equal(a, _tile_anchors_numpy(H, W, *other))
    # known answers by hand, 256x384, level 0 (stride 16, scale 40), cell (0,0): centre (8, 8)
    a = oracle_ops.anchors_ex(256, 384, *other) * np.array([256, 384, 256, 384], np.float32)
    np.testing.assert_allclose(a[0], [8 - 20, 8 - 20, 8 + 20, 8 + 20], atol=1e-4)                     # (1.0, 1.0)
    r3 = math.sqrt(3.0)
    np.testing.assert_allclose(a[1], [8 - 20 / r3, 8 - 20 * r3, 8 + 20 / r3, 8 + 20 * r3], atol=1e-4)  # (1.0, 3.0): w/h = 3
    np.testing.assert_allclose(a[4], [8 - 25.2, 8 - 25.2, 8 + 25.2, 8 + 25.2], atol=1e-4)              # (1.26, 1.0)
    np.testing.assert_allclose(a[12], [8 - 20, 24 - 20, 8 + 20, 24 + 20], atol=1e-4)                   # cell (0,1)


# --------------------------------------------------------------------------- decode
def test_decode_known_answers(oracle_ops):
    anc = np.array([[0.2, 0.3, 0.4, 0.7]], np.float32)
    z = np.zeros((1, 4), np.float32)
    np.testing.assert_allclose(oracle_ops.decode_clip(z, anc), anc, atol=1e-7)
    # ty = 10 shifts cy by exactly ha (SCALE_FACTORS, constants.py:15)
    d = oracle_ops.decode_clip(np.array([[10, 0, 0, 0]], np.float32), anc)
    np.testing.assert_allclose(d, [[0.4, 0.3, 0.6, 0.7]], atol=1e-6)
    # th = 5 ln 2 doubles h
    d = oracle_ops.decode_clip(np.array([[0, 0, 5 * math.log(2), 0]], np.float32), anc)
    np.testing.assert_allclose(d, [[0.1, 0.3, 0.5, 0.7]], atol=1e-6)
    # clip to [0,1] (nms.py:77)
    d = oracle_ops.decode_clip(np.array([[0, 0, 5 * math.log(8), 0]], np.float32), anc)
    np.testing.assert_allclose(d, [[0.0, 0.3, 1.0, 0.7]], atol=1e-6)


# --------------------------------------------------------------------------- NMS (TF r1.12)
def test_nms_semantics(oracle_ops):
    nms = oracle_ops.nms
    b = np.array([[0, 0, 1, 1], [0, 0, 1, 1]], np.float32)
    assert list(nms(b, np.array([0.9, 0.8], np.float32), 25, 0.6, 0.15)) == [0]   # identical -> one
    # IoU exactly at the threshold is NOT suppressed (strict >)
    sel = nms(np.array([[0, 0, 1, 1], [0, 0, 0.5, 1]], np.float32), np.array([0.9, 0.8], np.float32), 25, 0.5, 0.15)
    assert list(sel) == [0, 1]                                   # IoU == 0.5 exactly, thr 0.5
    # score exactly at the threshold is never selected (strict >)
    sel = nms(np.array([[0, 0, 1, 1]], np.float32), np.array([0.15], np.float32), 25, 0.6, np.float32(0.15))
    assert len(sel) == 0
    # max_output_size
    boxes = np.stack([np.array([i, 0, i + 0.5, 1], np.float32) for i in range(40)])
    sel = nms(boxes, np.linspace(0.9, 0.5, 40).astype(np.float32), 25, 0.6, 0.15)
    assert list(sel) == list(range(25))
    # zero-area boxes neither suppress nor are suppressed
    zb = np.zeros((30, 4), np.float32)
    sel = nms(zb, np.linspace(0.9, 0.5, 30).astype(np.float32), 25, 0.6, 0.15)
    assert list(sel) == list(range(25))
    # selection order is descending score; ties -> lower index first (our fixed tie rule)
    b = np.array([[0, 0, 1, 1], [2, 2, 3, 3], [4, 4, 5, 5]], np.float32)
    assert list(nms(b, np.array([0.5, 0.9, 0.5], np.float32), 25, 0.6, 0.15)) == [1, 0, 2]
    # flipped corners are normalised
    assert oracle_ops.iou_greater(np.array([1, 1, 0, 0], np.float32), np.array([0, 0, 1, 1], np.float32), 0.9)


def test_postprocess_layout(oracle_ops):
    # class-major, score-descending inside a class, zero padded to C*m, num_boxes (nms.py:42-44,83-93)
    N, C = 12, 3
    anc = np.zeros((N, 4), np.float32)
    for i in range(N):
        anc[i] = [0.05 * i, 0.05 * i, 0.05 * i + 0.04, 0.05 * i + 0.04]      # disjoint boxes
    codes = np.zeros((1, N, 4), np.float32)
    logits = np.full((1, N, C), -10.0, np.float32)
    logits[0, 3, 2] = 2.0
    logits[0, 5, 0] = 1.0
    logits[0, 7, 0] = 3.0
    logits[0, 9, 2] = 0.5
    boxes, labels, scores, num = oracle_ops.postprocess(logits, codes, anc, 0.15, 0.6, 4)
    assert boxes.shape == (1, 12, 4) and labels.dtype == np.int32 and num[0] == 4
    assert list(labels[0][:4]) == [0, 0, 2, 2]
    np.testing.assert_allclose(boxes[0][:4], anc[[7, 5, 3, 9]], atol=1e-6)
    assert scores[0][0] > scores[0][1] and scores[0][2] > scores[0][3]
    assert not boxes[0][4:].any() and not scores[0][4:].any() and not labels[0][4:].any()


def test_sigmoid_bias_init(oracle_ops):
    # box_predictor.py:121-127: bias -log(99) -> sigma = 0.01
    assert abs(oracle_ops.lib().orc_sigmoid(-math.log(99.0)) - 0.01) < 1e-7


# --------------------------------------------------------------------------- data movement
def test_shuffle_known_answer(oracle_ops):
    # D=4: x' = [x0,y0,x1,y1], y' = [x2,y2,x3,y3]  (shufflenet_v2.py:94-115)
    x = np.array([[10, 11, 12, 13]], np.float32)
    y = np.array([[20, 21, 22, 23]], np.float32)
    xo, yo = oracle_ops.concat_shuffle_split(x, y)
    assert list(xo[0]) == [10, 20, 11, 21] and list(yo[0]) == [12, 22, 13, 23]


def test_upsample_add(oracle_ops):
    c = np.arange(2 * 3 * 1, dtype=np.float32).reshape(1, 2, 3, 1)
    lat = np.zeros((1, 4, 6, 1), np.float32)
    up = oracle_ops.upsample2_add(c, lat)
    for i in range(2):
        for j in range(3):
            assert (up[0, 2 * i:2 * i + 2, 2 * j:2 * j + 2, 0] == c[0, i, j, 0]).all()


def test_resize_keeping_aspect_ratio(oracle_ops):
    # identity at 896x640, box_scaler = 1
    dims, bs = oracle_ops.resize_dims(640, 896, 640, 128)
    assert dims == (640, 896, 0, 0) and (bs == 1).all()
    # 480x640 (HxW): scale 640/480, long side round(640*1.3333334)=853 -> padded to 896
    dims, bs = oracle_ops.resize_dims(480, 640, 640, 128)
    assert dims == (640, 853, 0, 43) and abs(bs[1] - 853 / 896) < 1e-7 and bs[0] == 1
    dims, bs = oracle_ops.resize_dims(500, 375, 640, 128)       # portrait: height is the long side
    assert dims == (853, 640, 43, 0) and abs(bs[0] - 853 / 896) < 1e-7
    # half-to-even rounding of the long side (tf.round): 3 * 0.5 = 1.5 -> 2, 5 * 0.5 = 2.5 -> 2
    # nearest neighbour: src = min(floor(dst * in/out), in-1); pad band is zero
    img = np.arange(2 * 3, dtype=np.float32).reshape(1, 2, 3, 1)
    out = oracle_ops.resize_pad(img, (4, 6, 1, 2))
    assert out.shape == (1, 5, 8, 1)
    assert (out[0, :4, :6, 0] == np.repeat(np.repeat(img[0, :, :, 0], 2, 0), 2, 1)).all()
    assert not out[0, 4:].any() and not out[0, :, 6:].any()
    up = oracle_ops.resize_pad(np.arange(5, dtype=np.float32).reshape(1, 1, 5, 1), (1, 3, 0, 0))
    assert list(up[0, 0, :, 0]) == [0.0, 1.0, 3.0]             # floor(0*5/3), floor(1*1.667), floor(2*1.667)


def test_preprocess(oracle_ops):
    img = np.array([[[[0, 255, 128]]]], np.uint8)
    out = oracle_ops.preprocess(img)
    assert out[0, 0, 0, 0] == -1.0 and out[0, 0, 0, 1] == 1.0
    assert abs(out[0, 0, 0, 2] - (2 * 128 / 255 - 1)) < 1e-6


# --------------------------------------------------------------------------- convs vs torch
@pytest.mark.parametrize("cin,cout,k", [(16, 32, 3), (24, 40, 3), (8, 24, 1), (3, 32, 3)])
def test_conv_same_stride1(oracle_ops, cin, cout, k):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 9, 11, cin)).astype(np.float32)
    w = rng.standard_normal((k, k, cin, cout)).astype(np.float32)
    p = (k - 1) // 2
    ref = tconv(x, w, 1, (p, p, p, p))
    for scalar in (False, True):
        np.testing.assert_allclose(oracle_ops.conv2d(x, w, 1, "SAME", scalar=scalar), ref, rtol=1e-5, atol=1e-4)
    assert np.array_equal(oracle_ops.conv2d(x, w, 1, "SAME"), oracle_ops.conv2d(x, w, 1, "SAME", scalar=True))


def test_conv_padding_asymmetries(oracle_ops):
    """TF 'SAME' stride 2 on even sizes pads bottom/right only; conv2d_same stride 2
    (layer_utils.py:26-43) pads 1 on both sides then VALID."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, 8, 12, 16)).astype(np.float32)
    w = rng.standard_normal((3, 3, 16, 16)).astype(np.float32)
    same = oracle_ops.conv2d(x, w, 2, "SAME")
    expl = oracle_ops.conv2d(x, w, 2, "EXPLICIT")
    assert same.shape == expl.shape == (1, 4, 6, 16)
    np.testing.assert_allclose(same, tconv(x, w, 2, (0, 1, 0, 1)), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(expl, tconv(x, w, 2, (1, 1, 1, 1)), rtol=1e-5, atol=1e-4)
    assert np.abs(same - expl).max() > 0.1
    # window check with a one-hot input: output o reads rows 2o..2o+2 (SAME) vs 2o-1..2o+1
    x1 = np.zeros((1, 8, 8, 1), np.float32)
    x1[0, 0, 0, 0] = 1
    w1 = np.arange(9, dtype=np.float32).reshape(3, 3, 1, 1) + 1
    assert oracle_ops.conv2d(x1, w1, 2, "SAME")[0, 0, 0, 0] == 1      # tap (0,0)
    assert oracle_ops.conv2d(x1, w1, 2, "EXPLICIT")[0, 0, 0, 0] == 5  # tap (1,1)


@pytest.mark.parametrize("stride", [1, 2])
def test_depthwise(oracle_ops, stride):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 8, 10, 12)).astype(np.float32)
    w = rng.standard_normal((3, 3, 12, 1)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    wt = torch.from_numpy(w).permute(2, 3, 0, 1)          # [C,1,3,3]
    pad = (1, 1, 1, 1) if stride == 1 else (0, 1, 0, 1)
    ref = nhwc(F.conv2d(F.pad(xt, pad), wt, stride=stride, groups=12))
    np.testing.assert_allclose(oracle_ops.depthwise3x3(x, w, stride), ref, rtol=1e-5, atol=1e-5)


def test_maxpool(oracle_ops):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 8, 10, 5)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    ref = nhwc(F.max_pool2d(F.pad(xt, (0, 1, 0, 1), value=-np.inf), 3, 2))
    assert np.array_equal(oracle_ops.maxpool3x3s2(x), ref)


def test_batch_norm(oracle_ops):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 4, 4, 6)).astype(np.float32) * 4
    g, b, m = (rng.standard_normal(6).astype(np.float32) for _ in range(3))
    v = rng.uniform(0.5, 1.5, 6).astype(np.float32)
    ref = (x - m) / np.sqrt(v + 1e-3) * g + b
    np.testing.assert_allclose(oracle_ops.bn_act(x, g, b, m, v, None), ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(oracle_ops.bn_act(x, g, b, m, v, "relu"), np.maximum(ref, 0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(oracle_ops.bn_act(x, g, b, m, v, "relu6"), np.clip(ref, 0, 6), rtol=1e-5, atol=1e-5)


# --------------------------------------------------------------------------- whole graph
def test_head_layout(oracle_ops, oracle_graph):
    """channel a*C+c <-> anchor off_l + (y*w+x)*6 + a (box_predictor.py:91-102)."""
    rng = np.random.default_rng(6)
    nc = 3
    ps = [rng.standard_normal((1, h, w, 256)).astype(np.float32) * 0.1 for h, w in [(4, 4), (2, 2), (1, 1), (1, 1), (1, 1)]]
    W = {}
    for net, cout in (("box_net", 24), ("class_net", 6 * nc)):
        for i in range(4):
            W["%s/conv3x3_%d/kernel" % (net, i)] = (rng.standard_normal((3, 3, 256, 256)) * 0.02).astype(np.float32)
            for l in range(3, 8):
                s = "%s/batch_norm_%d_for_level_%d" % (net, i, l)
                W[s + "/gamma"] = np.ones(256, np.float32)
                W[s + "/beta"] = np.zeros(256, np.float32)
                W[s + "/moving_mean"] = np.zeros(256, np.float32)
                W[s + "/moving_variance"] = np.ones(256, np.float32)
        last = "encoded_boxes" if net == "box_net" else "logits"
        W["%s/%s/kernel" % (net, last)] = (rng.standard_normal((3, 3, 256, cout)) * 0.02).astype(np.float32)
        W["%s/%s/bias" % (net, last)] = rng.standard_normal(cout).astype(np.float32)
    tow = {}
    codes, logits = oracle_graph.box_predictor(ps, W, nc, tow)
    assert codes.shape == (1, (16 + 4 + 1 + 1 + 1) * 6, 4) and logits.shape == (1, 138, nc)
    z = oracle_ops.bias_add(oracle_ops.conv2d(tow["class_tower_3"], W["class_net/logits/kernel"], 1, "SAME"),
                            W["class_net/logits/bias"])
    y, x, a, c = 2, 1, 4, 2
    assert logits[0, (y * 4 + x) * 6 + a, c] == z[0, y, x, a * nc + c]
    # level 4 starts at 16*6
    z4 = oracle_ops.bias_add(oracle_ops.conv2d(tow["box_tower_4"], W["box_net/encoded_boxes/kernel"], 1, "SAME"),
                             W["box_net/encoded_boxes/bias"])
    assert codes[0, 96 + (1 * 2 + 0) * 6 + 3, 1] == z4[0, 1, 0, 3 * 4 + 1]


def test_detector_call_filter(oracle_graph):
    out = {"boxes": np.zeros((1, 5, 4), np.float32), "labels": np.arange(5, dtype=np.int32)[None],
           "scores": np.array([[0.9, 0.1, 0.5, 0.99, 0.99]], np.float32), "num_boxes": np.array([3], np.int32)}
    b, l, s = oracle_graph.detector_call(out, 0.1)
    assert list(l) == [0, 2] and b.shape == (2, 4)        # strict > on the first n entries only

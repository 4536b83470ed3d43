"""The C oracle (oracle/) against a SECOND, independent restatement of the reference graph written with torch's own
operators (tests/helpers/torch_graph.py: NCHW, F.conv2d / F.max_pool2d / F.interpolate, sort-based NMS).  The two share
no code: agreement per stage within 1e-5 of the tensor's scale checks the oracle's padding rules (SAME vs explicit),
strides, channel orders, the shuffle, the nearest-neighbour resize / upsample index rules, the head layout, anchors,
decode and the NMS decisions.  It does NOT pin either against TensorFlow (the reference cannot run here): parity stays
"unpinned"; this narrows what an error in the oracle could be to something both restatements got wrong the same way."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import torch_graph as tg  # noqa: E402

STAGES = ["c3", "c4", "c5", "p3", "p4", "p5", "p6", "p7", "encoded_boxes", "class_predictions"]


def params_of(backbone, md):
    return {"backbone": backbone, "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15, "iou_threshold": 0.6,
            "max_boxes_per_class": 25, "min_dimension": md}


@pytest.mark.parametrize("backbone,H,W,md", [("mobilenet", 128, 128, 128), ("shufflenet", 128, 128, 128),
                                              ("mobilenet", 256, 384, 256), ("shufflenet", 256, 384, 256),
                                              ("mobilenet", 100, 151, 128), ("shufflenet", 300, 128, 128)])
def test_whole_graph_stage_by_stage(ssd, oracle_graph, backbone, H, W, md):
    params = params_of(backbone, md)
    Wt = ssd.synthetic_weights(params, seed=31, logits_bias=-4.0)
    img = np.random.default_rng(H + W).integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    keep = {}
    ref = oracle_graph.forward(img, Wt, params, keep)
    st, anc, dets = tg.forward(img, Wt, params)
    for n in STAGES:
        a, b = st[n], keep[n].reshape(st[n].shape)
        scale = max(1.0, float(np.abs(b).max()))
        err = float(np.abs(a - b).max())
        assert err <= 1e-5 * scale * 8, (backbone, n, err, scale)      # fp32 sums in another order: a few ulps of the scale
    # anchors: another formulation of the same float32 formulas
    assert np.abs(anc - oracle_graph.ops.anchors(st["p3"].shape[1] * 8, st["p3"].shape[2] * 8)).max() <= 1e-6
    # detections: the torch graph's own heads through its own decode + sort-based NMS vs the oracle's outputs.  Scores that
    # differ in the last bits can flip a threshold or a tie: compare as sets per class with a tolerance and allow a handful.
    mismatched = 0
    total = 0
    for b, (bx, lb, sc, n) in enumerate(dets):
        rn = int(ref["num_boxes"][b])
        total += rn
        if n != rn:
            mismatched += abs(n - rn)
            continue
        assert np.array_equal(lb, ref["labels"][b][:rn])
        ok = (np.abs(sc - ref["scores"][b][:rn]) <= 1e-5) & (np.abs(bx - ref["boxes"][b][:rn]).max(axis=1) <= 1e-4)
        mismatched += int((~ok).sum())
    assert total > 20 and mismatched <= max(2, total // 200), (mismatched, total)


def test_postprocess_on_identical_heads(oracle_ops):
    """Same logits / codes into both post-processings: the decisions (which anchors, which order, which class slots) must
    be IDENTICAL, scores and boxes equal to the last bit or two (double-precision sigmoid / exp on both sides)."""
    rng = np.random.default_rng(7)
    H, W = 128, 256
    anc = oracle_ops.anchors(H, W)
    assert np.abs(anc - tg.anchors(H, W)).max() <= 1e-6
    N, C = anc.shape[0], 80
    for trial in range(3):
        codes = rng.standard_normal((1, N, 4)).astype(np.float32)
        logits = np.full((1, N, C), -6.0, np.float32)
        hot = rng.integers(0, N * C, N * C // 150)
        logits.reshape(-1)[hot] = rng.uniform(-2.5, 3.0, hot.size).astype(np.float32)
        # clusters of overlapping positives so that suppression actually happens
        for a in rng.integers(0, N - 12, 40):
            logits[0, a:a + 12, 3] = rng.uniform(-1.0, 3.0, 12)
            codes[0, a:a + 12] *= 0.05
        scaler = np.array([0.8, 1.0, 0.8, 1.0], np.float32)
        b, l, s, n = oracle_ops.postprocess(logits, codes, anc, 0.15, 0.6, 25, scaler)
        tb, tl, ts, tn = tg.postprocess(logits[0], codes[0], anc, 0.15, 0.6, 25, scaler)
        assert tn == int(n[0]) and tn > 100
        assert np.array_equal(tl, l[0][:tn])
        assert np.abs(ts - s[0][:tn]).max() <= 1e-7 and np.abs(tb - b[0][:tn]).max() <= 1e-6
        assert np.abs(b[0][tn:]).max() == 0 and np.abs(s[0][tn:]).max() == 0


def test_resize_rule_against_torch_interpolate(oracle_ops):
    """pipeline.py:138-194 sizes and the nearest-neighbour index rule against torch's F.interpolate(mode='nearest')."""
    import torch
    rng = np.random.default_rng(1)
    for (H, W, md) in [(100, 151, 128), (300, 128, 128), (77, 201, 128), (480, 640, 640), (333, 500, 256), (128, 128, 128)]:
        img = rng.integers(0, 256, (1, H, W, 3)).astype(np.float32)
        dims, bs = oracle_ops.resize_dims(H, W, md, 128)
        ref = oracle_ops.resize_pad(img, dims)
        got, scaler = tg.resize_keeping_aspect_ratio(torch.from_numpy(img).permute(0, 3, 1, 2), md)
        assert tuple(got.shape[2:]) == ref.shape[1:3], (H, W)
        assert np.array_equal(got.permute(0, 2, 3, 1).numpy(), ref), (H, W)
        assert np.allclose(scaler, bs, rtol=0, atol=1e-7)

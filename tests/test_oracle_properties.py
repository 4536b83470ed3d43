"""Property tests (CPU): the oracle's C NMS / post-processing against an independent
pure-Python restatement of TF r1.12 NonMaxSuppressionV3 (priority queue by score, strict
thresholds, IoU with normalised corners) on random inputs, incl. ties and degenerate boxes."""
import heapq

import numpy as np
from hypothesis import given, settings, strategies as st


def py_iou_greater(a, b, thr):
    a = a.astype(np.float32); b = b.astype(np.float32)
    ymin_i, xmin_i = min(a[0], a[2]), min(a[1], a[3]); ymax_i, xmax_i = max(a[0], a[2]), max(a[1], a[3])
    ymin_j, xmin_j = min(b[0], b[2]), min(b[1], b[3]); ymax_j, xmax_j = max(b[0], b[2]), max(b[1], b[3])
    area_i = np.float32(ymax_i - ymin_i) * np.float32(xmax_i - xmin_i)
    area_j = np.float32(ymax_j - ymin_j) * np.float32(xmax_j - xmin_j)
    if area_i <= 0 or area_j <= 0:
        return False
    ih = max(np.float32(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), np.float32(0))
    iw = max(np.float32(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), np.float32(0))
    inter = np.float32(ih * iw)
    return np.float32(inter / np.float32(np.float32(area_i + area_j) - inter)) > np.float32(thr)


def py_nms(boxes, scores, max_out, iou_thr, score_thr):
    heap = [(-float(s), i) for i, s in enumerate(scores) if s > np.float32(score_thr)]   # ties: lower index first
    heapq.heapify(heap)
    sel = []
    while heap and len(sel) < max_out:
        _, i = heapq.heappop(heap)
        if all(not py_iou_greater(boxes[i], boxes[j], iou_thr) for j in reversed(sel)):
            sel.append(i)
    return sel


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 60), st.sampled_from([0.3, 0.5, 0.6]), st.sampled_from([0.05, 0.15, 0.5]),
       st.integers(1, 25))
def test_nms_matches_python_restatement(oracle_ops, seed, n, iou_thr, score_thr, max_out):
    rng = np.random.default_rng(seed)
    ctr = rng.uniform(0.2, 0.8, (n, 2)); size = rng.uniform(0.0, 0.5, (n, 2))
    boxes = np.concatenate([ctr - size / 2, ctr + size / 2], 1).astype(np.float32)
    boxes[rng.random(n) < 0.1] = boxes[0]                      # duplicates
    flip = rng.random(n) < 0.1
    boxes[flip] = boxes[flip][:, [2, 3, 0, 1]]                 # flipped corners
    z = rng.random(n) < 0.05
    boxes[z, 2] = boxes[z, 0]                                  # zero-height boxes: IoU 0 with everything
    scores = rng.choice(np.linspace(0, 1, 12), n).astype(np.float32)   # many ties, some exactly at thresholds
    got = list(oracle_ops.nms(boxes, scores, max_out, iou_thr, score_thr))
    assert got == py_nms(boxes, scores, max_out, iou_thr, score_thr)


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 2 ** 31 - 1))
def test_postprocess_structure(oracle_ops, seed):
    rng = np.random.default_rng(seed)
    anc = oracle_ops.anchors(128, 128)
    N, C, m = anc.shape[0], 5, 4
    codes = (rng.standard_normal((2, N, 4)) * 0.5).astype(np.float32)
    logits = (rng.standard_normal((2, N, C)) * 1.5 - 4.0).astype(np.float32)
    boxes, labels, scores, num = oracle_ops.postprocess(logits, codes, anc, 0.15, 0.6, m)
    dec = [oracle_ops.decode_clip(codes[b], anc) for b in range(2)]
    for b in range(2):
        n = num[b]
        assert n <= C * m and (np.diff(labels[b][:n]) >= 0).all()
        assert np.bincount(labels[b][:n], minlength=C).max() <= m
        same = np.diff(labels[b][:n]) == 0
        assert (np.diff(scores[b][:n])[same] <= 0).all() and (scores[b][:n] > np.float32(0.15)).all()
        assert not boxes[b][n:].any() and not scores[b][n:].any() and not labels[b][n:].any()
        # per class the selection equals the python restatement on the full (unfiltered) row set
        prob = (1.0 / (1.0 + np.exp(-logits[b].astype(np.float64)))).astype(np.float32)
        for c in range(C):
            sel = py_nms(dec[b], prob[:, c], m, 0.6, 0.15)
            k = labels[b][:n] == c
            assert k.sum() == len(sel)
            assert np.array_equal(boxes[b][:n][k], dec[b][sel])

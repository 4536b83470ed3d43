"""Parity risk register.  TEST INFRASTRUCTURE: runs the CPU oracle only.

The oracle restates five TF r1.12 op semantics from recall -- nothing in the reference pins them: the equal-score order of
NonMaxSuppressionV3, tf.sigmoid / tf.exp as correctly rounded values (TF's are Eigen approximations), tf.round as half-to-even,
the source-index rule of ResizeNearestNeighbor, and the fused form of inference batch norm.  For each, oracle/ssd_oracle.c carries an
ORACLE-SIDE switch to the plausible alternate reading (the product has none).  This script runs the oracle's whole graph on the
benchmark's own frames (bench.py: MobileNet 640x896, seed-0 weights, logits bias -7.5, torch frames of seed 1234) under the
default and under each alternate and counts what changes in the detections -- so "parity unpinned" becomes "unpinned, bounded:
at most k of n detections move under any alternate reading".  Resize / rounding alternates cannot touch frames that arrive at
the network's size (identity resize): they are measured on odd-sized frames as well.

usage: python tests/parity_risk.py [n_frames (default 32)] [n_odd_frames (default 6)]   -> text report on stdout
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import graph, ops  # noqa: E402
import ssd_amd  # noqa: E402  (synthetic_weights / load_config: host-side helpers, no GPU)

PARAMS = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
          "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
LOGITS_BIAS = -7.5          # bench.py LOGITS_BIAS["mobilenet"]


def detections(out):
    """per image: list of (label, score, box[4]) in output order"""
    res = []
    for b in range(out["num_boxes"].shape[0]):
        n = int(out["num_boxes"][b])
        res.append([(int(out["labels"][b, i]), float(out["scores"][b, i]), out["boxes"][b, i].copy()) for i in range(n)])
    return res


def compare(base, alt, tol=1e-4):
    """base / alt: lists (per image) of detections.  A detection of `base` is MATCHED when `alt` holds one of the same label whose
    score and box agree within tol (the north star's tolerance); returns counts."""
    total = moved = added = 0
    images_changed = 0
    max_ds = 0.0
    for db, da in zip(base, alt):
        used = [False] * len(da)
        lost = 0
        for (l, s, bx) in db:
            hit = -1
            for j, (l2, s2, bx2) in enumerate(da):
                if not used[j] and l2 == l and abs(s2 - s) <= tol and np.abs(bx2 - bx).max() <= tol:
                    hit = j
                    break
            if hit >= 0:
                used[hit] = True
                max_ds = max(max_ds, abs(da[hit][1] - s))
            else:
                lost += 1
        total += len(db)
        moved += lost
        added += used.count(False)
        images_changed += 1 if (lost or used.count(False)) else 0
    return {"detections": total, "not_matched_in_alternate": moved, "new_in_alternate": added, "images_changed": images_changed,
            "max_score_diff_of_matched": max_ds}


def post_only(heads, anchors, box_scaler):
    codes, logits = heads
    b, l, s, n = ops.postprocess(logits, codes, anchors, PARAMS["score_threshold"], PARAMS["iou_threshold"],
                                 PARAMS["max_boxes_per_class"], box_scaler)
    return {"boxes": b, "labels": l, "scores": s, "num_boxes": n}


def main():
    import torch
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n_odd = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ops.build()
    Wt = ssd_amd.synthetic_weights(PARAMS, seed=0, logits_bias=LOGITS_BIAS)
    g = torch.Generator().manual_seed(1234)          # bench.py rank 0
    frames = torch.randint(0, 256, (32, 640, 896, 3), dtype=torch.uint8, generator=g).numpy()[:n_frames]
    report = {"frames": int(frames.shape[0]), "frame_size": [640, 896], "tolerance": 1e-4, "alternates": {}}
    t0 = time.time()
    # default reading, frame by frame (memory), keeping the head outputs for the post-processing-only alternates
    base_out, heads = [], []
    for i in range(frames.shape[0]):
        keep = {}
        o = graph.forward(frames[i:i + 1], Wt, PARAMS, keep)
        base_out.append(o)
        heads.append((keep["encoded_boxes"], keep["class_predictions"]))
    base = [d for o in base_out for d in detections(o)]
    anchors = ops.anchors(640, 896)
    one = np.ones(4, np.float32)
    print("# default reading: %d frames, %d detections (%.0f s)" % (len(base), sum(len(d) for d in base), time.time() - t0), flush=True)
    for name, value, what in (("nms_tie", 1, "equal scores: higher box index first"),
                              ("fast_exp", 1, "polynomial fp32 expf / sigmoid (Cephes / Eigen form) instead of the correctly rounded value")):
        ops.set_alternate(name, value)
        alt = [d for h in heads for d in detections(post_only(h, anchors, one))]
        ops.set_alternate(name, 0)
        report["alternates"]["%s=%d" % (name, value)] = dict(compare(base, alt), what=what, scope="post-processing on the default reading's head outputs")
        print(name, value, report["alternates"]["%s=%d" % (name, value)], flush=True)
    ops.set_alternate("bn_form", 1)
    alt = [d for i in range(frames.shape[0]) for d in detections(graph.forward(frames[i:i + 1], Wt, PARAMS))]
    ops.set_alternate("bn_form", 0)
    report["alternates"]["bn_form=1"] = dict(compare(base, alt), what="x * inv + (beta - mean * inv) instead of (x - mean) * inv + beta",
                                             scope="the whole graph (87 batch norms)")
    print("bn_form 1", report["alternates"]["bn_form=1"], flush=True)
    # resize / rounding: identity on the bench frames by construction; measured on odd-sized frames
    rng = np.random.default_rng(7)
    # (256 x 257: the long side becomes 257 * 2.5 = 642.5 -- the one kind of size where half-even and half-up differ)
    sizes = [(256, 257), (427, 640), (480, 640), (375, 500), (500, 333), (612, 612), (333, 500), (640, 427)][:n_odd]

    def smooth(h, w):
        """a frame with natural-image-like spectrum: a coarse random field, bilinearly enlarged (uniform noise frames turn into a
        DIFFERENT image under any other sampling grid, which says nothing about photographs)"""
        t = torch.from_numpy(rng.random((1, 3, 12, 12)).astype(np.float32))
        up = torch.nn.functional.interpolate(t, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)
        return (up[0].permute(1, 2, 0).numpy() * 255.0).astype(np.uint8)[None]
    odd = [smooth(h, w) for (h, w) in sizes]
    base_odd = [d for im in odd for d in detections(graph.forward(im, Wt, PARAMS))]
    for name, value, what in (("round", 1, "tf.round as half-up"), ("resize", 1, "nearest neighbour with half-pixel centres"),
                              ("resize", 2, "nearest neighbour with align_corners")):
        ops.set_alternate(name, value)
        alt = [d for im in odd for d in detections(graph.forward(im, Wt, PARAMS))]
        ops.set_alternate(name, 0)
        report["alternates"]["%s=%d" % (name, value)] = dict(compare(base_odd, alt), what=what,
                                                             scope="%d odd-sized SMOOTH frames %s (the bench frames arrive at the network's size: no resize, 0 changes by construction)" % (len(odd), sizes))
        print(name, value, report["alternates"]["%s=%d" % (name, value)], flush=True)
    report["seconds"] = time.time() - t0
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()

"""Generates tests/golden/tiny_mobilenet_128.npz with the CPU oracle (oracle/), NOT with
TensorFlow: the reference cannot be imported here (TF 1.12 is not installed) and ships no
golden vectors, so these fixtures pin the HIP path to the oracle, not to TensorFlow.

    python tests/golden/make_golden.py

Inputs are regenerated from seeds at test time (numpy default_rng + synthetic_weights);
the fixture stores expected outputs only.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import ssd_amd  # noqa: E402  (only for the seeded synthetic weights / config helpers)
from oracle import graph  # noqa: E402

TINY = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80,
        "score_threshold": 0.15, "iou_threshold": 0.6, "max_boxes_per_class": 25,
        "min_dimension": 128}
SEED, IMG_SEED, B, H, W = 7, 3, 2, 128, 128
LOGITS_BIAS, HEAD_STD = -4.0, 0.01


def inputs(params=TINY, seed=SEED):
    Wt = ssd_amd.synthetic_weights(params, seed=seed, logits_bias=LOGITS_BIAS, head_std=HEAD_STD)
    img = np.random.default_rng(IMG_SEED).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    return Wt, img


if __name__ == "__main__":
    Wt, img = inputs()
    keep = {}
    out = graph.forward(img, Wt, TINY, keep)
    fix = {"boxes": out["boxes"], "labels": out["labels"], "scores": out["scores"],
           "num_boxes": out["num_boxes"], "encoded_boxes": keep["encoded_boxes"],
           "class_predictions_every8": keep["class_predictions"][:, ::8].copy(),
           "c5": keep["c5"], "p5": keep["p5"], "p6": keep["p6"], "p7": keep["p7"],
           "c3_corner": keep["c3"][:, :4, :4].copy(), "p3_corner": keep["p3"][:, :4, :4].copy()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_mobilenet_128.npz")
    np.savez_compressed(path, **fix)
    print("wrote", path, os.path.getsize(path), "bytes; num_boxes", out["num_boxes"])

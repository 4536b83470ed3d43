"""Upper bound on what fusing a ShuffleNet unit's conv1x1_before into its depthwise + conv1x1_after launch can return
(VERDICT r5 item 6): the diagnostics build drops the conv1x1_before launches of the chosen stages altogether (env
SSD_ABL_SKIP_PWG, bit s = Stage s+2; the results are WRONG) -- a step in which the first 1x1 of every unit costs nothing, not even
the recompute of its halo.  No fused kernel can be faster than that.  Same process, same box, BASELINE config 4 (64 frames).

    python scripts/experiments/sn_unit_fusion_bound.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import ssd_amd
from importlib import import_module

import_module("ssd_amd._lib").use_diag()
P = {"backbone": "shufflenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
W = ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5)
img = torch.randint(0, 256, (64, 640, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(4321)).cuda()


def run(mask, steps=20):
    os.environ["SSD_ABL_SKIP_PWG"] = str(mask)
    e = ssd_amd.Engine(P, W)
    for _ in range(3):
        out = e.forward(img)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        out = e.forward(img)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    det = float(out[3].float().mean())
    e.close()
    return ms, det


base = None
for rnd in range(2):
    for mask, what in ((0, "as shipped"), (1, "Stage2 units 2-4 without conv1x1_before (K = 58: the 80x80 rows)"),
                       (3, "... and Stage3 units 2-8 (K = 116)"), (7, "... and Stage4 units 2-4 (K = 232): every unit")):
        ms, det = run(mask)
        if mask == 0:
            base = ms
        print("round %d  mask %d  %-72s %8.3f ms/step  %7.1f img/s  (%+.3f ms)  detections/img %.1f"
              % (rnd, mask, what, ms, 64e3 / ms, ms - base, det), flush=True)

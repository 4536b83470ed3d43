"""Depthwise layer timings at the MobileNet shapes (B=32) via the stage entry (includes two
channel permutes: subtract by timing them?) -- instead time the full forward classes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ssd_amd
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
B = 32
e = ssd_amd.Engine(P, ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5))
img = torch.randint(0, 256, (B, 640, 896, 3), dtype=torch.uint8).cuda()
for _ in range(2):
    e.forward(img)
e.profile_reset(); e.profile_enable(True)
for _ in range(5):
    e.forward(img)
torch.cuda.synchronize(); e.profile_enable(False)
tot = 0
for k, v in e.profile_read().items():
    print("  %-16s %8.3f ms/step  (%d launches)  %.2f TB/s alg" % (k, v["ms"] / 5, v["launches"] / 5, v["bytes"] / max(v["ms"], 1e-9) / 1e9)); tot += v["ms"] / 5
print("  sum %.3f ms -> %.1f img/s" % (tot, B / tot * 1e3))

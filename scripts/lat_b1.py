"""Batch-1 breakdown: per-class kernel time (HIP events) and wall latency of Engine.forward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for kv in sys.argv[2:]:          # library options, key=value
    k, v = kv.split("=")
    ssd_amd.set_option(k, int(v, 0))
W = ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5)
e = ssd_amd.Engine(P, W)
img = torch.randint(0, 256, (B, 640, 896, 3), dtype=torch.uint8).cuda()
for _ in range(5):
    e.forward(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    out = e.forward(img)
torch.cuda.synchronize()
print("B=%d forward wall (no profiling): %.3f ms" % (B, (time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for _ in range(50):
    out = e.forward(img)
host = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
print("host enqueue time per forward: %.3f ms" % host)
e.profile_reset(); e.profile_enable(True)
for _ in range(20):
    e.forward(img)
torch.cuda.synchronize()
e.profile_enable(False)
pr = e.profile_read()
tot = 0
for k, v in pr.items():
    print("  %-16s %8.3f ms/forward  (%d launches/forward)" % (k, v["ms"] / 20, v["launches"] / 20)); tot += v["ms"] / 20
print("  sum of kernel time %.3f ms" % tot)

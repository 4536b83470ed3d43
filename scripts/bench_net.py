"""Per-class kernel time of one forward: python scripts/bench_net.py backbone B H W"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ssd_amd
bb = sys.argv[1] if len(sys.argv) > 1 else "mobilenet"
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (32, 640, 896)
P = {"backbone": bb, "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": min(H, W)}
e = ssd_amd.Engine(P, ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5 if bb == "mobilenet" else -12.0))
img = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8).cuda()
for _ in range(2):
    out = e.forward(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = e.forward(img)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5 * 1e3
e.profile_reset(); e.profile_enable(True)
for _ in range(5):
    e.forward(img)
torch.cuda.synchronize(); e.profile_enable(False)
tot = 0
for k, v in e.profile_read().items():
    print("  %-16s %8.3f ms/step  (%d launches)  %.2f TB/s alg  %.1f TFLOP/s" % (k, v["ms"] / 5, v["launches"] / 5, v["bytes"] / max(v["ms"], 1e-9) / 1e9, v["flops"] / max(v["ms"], 1e-9) / 1e9)); tot += v["ms"] / 5
print("%s B=%d %dx%d: wall %.3f ms -> %.1f img/s (kernel sum %.3f ms), detections/img %.1f" % (bb, B, H, W, wall, B / wall * 1e3, tot, float(out[3].float().mean())))

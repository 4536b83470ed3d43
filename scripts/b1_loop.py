"""Batch-1 forwards for a kernel trace (rocprofv3 --kernel-trace -- python3 scripts/b1_loop.py [n] [f32|f16x3] [option=value ...]):
n synchronised forwards of one resident 640x896 frame; scripts/b1_timeline.py prints one of them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ssd_amd
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for kv in sys.argv[3:]:          # library options, key=value (include/ssd_hip.h ssd_set_option)
    k, v = kv.split("=")
    ssd_amd.set_option(k, int(v, 0))
e = ssd_amd.Engine(P, ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), precision=sys.argv[2] if len(sys.argv) > 2 else "f32")
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (1, 640, 896, 3), dtype=torch.uint8, generator=g).cuda()
for _ in range(n):
    out = e.forward(img)
    torch.cuda.synchronize()
print("detections", int(out[3][0]))

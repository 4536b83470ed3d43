"""Full-size forwards for a kernel trace (rocprofv3 --kernel-trace -- python3 scripts/net_loop.py <mobilenet|shufflenet> [batch] [n] [option=value ...]):
n synchronised forwards of one resident batch (640x896 / 640x640); scripts/b1_timeline.py prints one of them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ssd_amd, bench
NET = sys.argv[1] if len(sys.argv) > 1 else "shufflenet"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if NET == "mobilenet" else 64)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    ssd_amd.set_option(k, int(v, 0))
P = bench.PARAMS if NET == "mobilenet" else bench.PARAMS_SHUFFLE
e = ssd_amd.Engine(P, ssd_amd.synthetic_weights(P, seed=0, logits_bias=bench.LOGITS_BIAS[NET]))
shape = (B, bench.H, bench.W, 3) if NET == "mobilenet" else (B, 640, 640, 3)
img = torch.randint(0, 256, shape, dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).cuda()
for _ in range(n):
    out = e.forward(img)
    torch.cuda.synchronize()
print("detections", int(out[3].sum()))

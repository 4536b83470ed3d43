"""Every op of one forward run ALONE (option debug_sync): class, algorithmic FLOP and bytes, time, achieved rates.
usage: python scripts/op_table.py [mobilenet|shufflenet] [batch]   (the table goes to stderr of the library: captured here)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ssd_amd, bench
NET = sys.argv[1] if len(sys.argv) > 1 else "mobilenet"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if NET == "mobilenet" else 64)
PARAMS = bench.PARAMS if NET == "mobilenet" else bench.PARAMS_SHUFFLE
eng = ssd_amd.Engine(PARAMS, ssd_amd.synthetic_weights(PARAMS, seed=0, logits_bias=bench.LOGITS_BIAS[NET]))
shape = (B, bench.H, bench.W, 3) if NET == "mobilenet" else (B, 640, 640, 3)
frames = torch.randint(0, 256, shape, dtype=torch.uint8).cuda()
for _ in range(3):
    eng.forward(frames)
torch.cuda.synchronize()
eng.set_option("debug_sync", 1)
print("# %s, batch %d: classes 0 = 3x3 igemm, 1 = 1x1 igemm, 2 = depthwise, 3 = first conv, 4 = post, 5 = other, 6 = fused dw+pw" % (NET, B), file=sys.stderr, flush=True)
eng.forward(frames)
torch.cuda.synchronize()

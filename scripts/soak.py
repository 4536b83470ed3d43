"""Race screen at full size: N forwards of the bench workload (32 frames of 640x896, precision f16x3, two
streams, 256x256-tile kernel with LDS-DMA staging and the fused candidate bitmap) on the same input must all
give the same bits as the first.  usage: python scripts/soak.py [N] [f16x3|f32] [mobilenet|shufflenet] [batch]
(batch 1 / 2 screen the small-batch plan of round 3: grouped FPN launch, igemm_lat kernel, the post-processing's LDS-aggregated list appends and its top-score trial)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ssd_amd
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
MODE = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
NET = sys.argv[3] if len(sys.argv) > 3 else "mobilenet"
PARAMS = bench.PARAMS if NET == "mobilenet" else bench.PARAMS_SHUFFLE
Wt = ssd_amd.synthetic_weights(PARAMS, seed=0, logits_bias=bench.LOGITS_BIAS[NET])
eng = ssd_amd.Engine(PARAMS, Wt, precision=MODE)
g = torch.Generator().manual_seed(1234)
shape = (32, bench.H, bench.W, 3) if NET == "mobilenet" else (64, 640, 640, 3)
if len(sys.argv) > 4:
    shape = (int(sys.argv[4]),) + shape[1:]
frames = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g).cuda()
first = [t.clone() for t in eng.forward(frames)]
bad = 0
for i in range(N):
    out = eng.forward(frames)
    if not all(torch.equal(a, b) for a, b in zip(first, out)):
        bad += 1
        print("forward %d differs from the first" % i, flush=True)
torch.cuda.synchronize()
print("batch %d:" % shape[0], end=" ")
print("%d forwards, %d different from the first; status %d; detections per image %.1f"
      % (N, bad, eng.status(), float(first[3].float().mean())))
sys.exit(1 if bad else 0)

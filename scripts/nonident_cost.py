"""What the frames' size costs at 32 frames per step: 640x896 frames (identity resize: the fused front launch) against 480x640 and
375x500 frames (both resize to the same 640x896 network input: general first convolution + Conv2d_1 as its own launches) and a
mixed batch of the three.  usage: python scripts/nonident_cost.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd, bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
P = bench.PARAMS
eng = ssd_amd.Engine(P, ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), device=0)
g = torch.Generator().manual_seed(1)


def run(tag, fn, n=steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    eng.profile_reset(); eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    eng.profile_enable(False)
    prof = eng.profile_read()
    print("%-34s %.3f ms per step = %.1f img/s   " % (tag, dt, 32e3 / dt) + "  ".join("%s %.3f" % (k, v["ms"] / n) for k, v in prof.items() if v["launches"]), flush=True)


for h, w in ((640, 896), (480, 640), (375, 500), (640, 896)):
    fr = torch.randint(0, 256, (32, h, w, 3), dtype=torch.uint8, generator=g).cuda()
    run("32 frames of %dx%d" % (h, w), lambda: eng.forward(fr))
for fuse in (0, 1):
    eng.set_option("front_fuse", fuse)
    fr = torch.randint(0, 256, (32, 640, 896, 3), dtype=torch.uint8, generator=g).cuda()
    run("640x896, front_fuse=%d" % fuse, lambda: eng.forward(fr))

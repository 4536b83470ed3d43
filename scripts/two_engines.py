"""Upper bound of cross-step overlap: N engines on N streams, steps issued round-robin (the backbone of one step may run
beside the heads of another).  usage: python scripts/two_engines.py [precision] [backbone]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ssd_amd
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
net = sys.argv[2] if len(sys.argv) > 2 else "mobilenet"
PARAMS = bench.PARAMS if net == "mobilenet" else bench.PARAMS_SHUFFLE
Wt = ssd_amd.synthetic_weights(PARAMS, seed=0, logits_bias=bench.LOGITS_BIAS[net])
B, H, W = (32, 640, 896) if net == "mobilenet" else (64, 640, 640)
frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8).cuda()


def run(n_eng, steps=12):
    engs = [ssd_amd.Engine(PARAMS, Wt, precision=prec) for _ in range(n_eng)]
    streams = [torch.cuda.Stream() for _ in range(n_eng)]
    for w in range(3):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.forward(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        e, s = engs[k % n_eng], streams[k % n_eng]
        with torch.cuda.stream(s):
            out = e.forward(frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e in engs:
        e.close()
    return B * steps / dt, dt / steps * 1e3


for n in (1, 2, 3, 1, 2):
    v, ms = run(n)
    print("%s %s engines/streams %d: %.0f img/s, %.2f ms per step" % (net, prec, n, v, ms), flush=True)

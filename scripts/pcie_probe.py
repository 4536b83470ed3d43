"""Is the host-fed rate (Detector.detect_stream) really above the resident rate?  Resident steps before / after a precision round
trip (which drops and rebuilds the plan: a new arena), then detect_stream timed the way bench.py times it and with an explicit
device synchronisation around N batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd, bench
W = ssd_amd.synthetic_weights(bench.PARAMS, seed=0, logits_bias=-7.5)
det = ssd_amd.Detector(W, config=bench.PARAMS)
eng = det.engine
g = torch.Generator().manual_seed(1234)
frames = torch.randint(0, 256, (32, 640, 896, 3), dtype=torch.uint8, generator=g).cuda()
host = frames.cpu().numpy()


def resident(n=20):
    for _ in range(3):
        eng.forward(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward(frames)
    torch.cuda.synchronize()
    return 32 * n / (time.perf_counter() - t0)


def stream_bench_way(n=10):
    it = det.detect_stream(host for _ in range(n + 1))
    next(it)
    t1 = time.perf_counter()
    for o in it:
        pass
    return 32 * n / (time.perf_counter() - t1)


def stream_synced(n=10):
    list(det.detect_stream(host for _ in range(3)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    for o in det.detect_stream(host for _ in range(n)):
        k += 1
    torch.cuda.synchronize()
    return 32 * k / (time.perf_counter() - t0)


for rnd in range(3):
    print("round %d: resident %.1f img/s   detect_stream (bench.py's window) %.1f   detect_stream (N batches between device syncs) %.1f"
          % (rnd, resident(), stream_bench_way(), stream_synced()), flush=True)
    eng.set_precision("f16x3"); eng.forward(frames); torch.cuda.synchronize(); eng.set_precision("f32")
    print("   after a precision round trip: resident %.1f   detect_stream (bench) %.1f   (synced) %.1f" % (resident(), stream_bench_way(), stream_synced()), flush=True)

# ... bench.py's own sequence: its Timed.run (profiling on) in f16x3 through detect_sharded, back to f32, then the stream leg
import torch.distributed as dist
timed = bench.Timed(1, dist, torch.cuda.synchronize, torch.device("cuda", 0), False)
step = lambda: ssd_amd.detect_sharded(eng, frames, total=32, force=False)
eng.set_precision("f16x3")
dt, out, prof = timed.run(eng, step, 5, 2)
print("f16x3 leg: %.1f img/s" % (32 * 5 / dt))
eng.set_precision("f32")
for k in range(4):
    print("   after bench's f16x3 leg, try %d: detect_stream (bench) %.1f   resident %.1f" % (k, stream_bench_way(), resident()), flush=True)

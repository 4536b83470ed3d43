"""Does the batch-1 Detector latency carry a transient after bench.py's sustained leg?  One Detector: `seconds` of back-to-back
32-frame steps (the sustained leg), then batch-1 calls for a few seconds, p50 per block of 100 calls with the time since the
load ended and the shader clock (sysfs).  usage: python scripts/lat_transient.py [seconds] [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd
import bench
P = bench.PARAMS
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
numa = ssd_amd.bind_to_gpu_numa_node(0)
det = ssd_amd.Detector(ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), config=P)
e = det.engine
img = np.random.default_rng(0).integers(0, 256, (640, 896, 3), dtype=np.uint8)
img32 = torch.randint(0, 256, (32, 640, 896, 3), dtype=torch.uint8).cuda()
read_mhz, read_w = bench._sysfs_probe(0)


def blocks(tag, n):
    t_start = time.perf_counter()
    ts, at = [], []
    for _ in range(n):
        t0 = time.perf_counter(); det(img, score_threshold=0.5); t1 = time.perf_counter()
        ts.append((t1 - t0) * 1e3); at.append(t1 - t_start)
    for k in range(0, n, 100):
        b = ts[k:k + 100]
        print("%-28s calls %4d..%4d  t=%6.2f s  p50 %.4f  p95 %.4f ms" % (tag, k, k + len(b) - 1, at[k], np.percentile(b, 50), np.percentile(b, 95)), flush=True)


print("numa node", numa, "sclk", read_mhz() if read_mhz else None)
blocks("fresh process", 300)
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < seconds:
    e.forward(img32); n += 1
    if n % 8 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
print("load: %d steps in %.1f s; sclk %s, power %s" % (n, time.perf_counter() - t0, read_mhz() if read_mhz else None, read_w() if read_w else None), flush=True)
blocks("after %.0f s of 32-frame steps" % seconds, calls)
print("sclk", read_mhz() if read_mhz else None)
time.sleep(5.0)
blocks("after 5 s of idle", 300)

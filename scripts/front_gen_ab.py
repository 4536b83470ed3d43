"""The fused first-layers launch on RESIZED frames, on (default) and off (option front_fuse = 0): Detector.__call__ p50 on 480 x 640
and 375 x 500 frames and a 32-frame step of 480 x 640 frames, both backbones.  usage: python scripts/front_gen_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd, bench
ssd_amd.bind_to_gpu_numa_node(0)
for P in (bench.PARAMS, bench.PARAMS_SHUFFLE):
    W = ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5)
    dets = {}
    for fuse in (1, 0):
        d = ssd_amd.Detector(W, config=P)
        d.engine.set_option("front_fuse", fuse)
        dets[fuse] = d
    rng = np.random.default_rng(0)
    for h, w in ((480, 640), (375, 500), (640, 896) if P["backbone"] == "mobilenet" else (640, 640)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        res = {}
        for rnd in range(2):
            for fuse in (1, 0):
                d = dets[fuse]
                for _ in range(10):
                    d(img, score_threshold=0.5)
                t = []
                for _ in range(100):
                    t0 = time.perf_counter(); out = d(img, score_threshold=0.5); t.append((time.perf_counter() - t0) * 1e3)
                res.setdefault(fuse, []).append(float(np.percentile(t, 50)))
        same = all(np.array_equal(a, b) for a, b in zip(dets[1](img, 0.5), dets[0](img, 0.5)))
        print("%-10s %dx%d batch 1: fused %s ms, two launches %s ms; identical %s" % (P["backbone"], h, w, ["%.4f" % v for v in res[1]], ["%.4f" % v for v in res[0]], same), flush=True)
    fr = torch.randint(0, 256, (32, 480, 640, 3), dtype=torch.uint8).cuda()
    for rnd in range(2):
        for fuse in (1, 0):
            e = dets[fuse].engine
            for _ in range(3):
                e.forward(fr)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                e.forward(fr)
            torch.cuda.synchronize()
            print("%-10s 32 frames of 480x640, front_fuse=%d: %.3f ms per step" % (P["backbone"], fuse, (time.perf_counter() - t0) * 100), flush=True)
    for d in dets.values():
        d.close()

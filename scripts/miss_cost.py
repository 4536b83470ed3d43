"""What a plan-cache MISS costs: the first forward of a new (network shape, batch) on a warm engine, against the same call again.
usage: python scripts/miss_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd, bench
for P in (bench.PARAMS, bench.PARAMS_SHUFFLE):
    eng = ssd_amd.Engine(P, ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), device=0)
    eng.forward(torch.zeros((1, 640, 640, 3), dtype=torch.uint8, device="cuda")); torch.cuda.synchronize()     # streams, kernels loaded
    for B, h, w in ((1, 640, 896), (1, 480, 640), (1, 427, 640), (2, 640, 896), (4, 640, 896), (8, 640, 896), (16, 640, 896), (32, 640, 896), (32, 427, 640), (64, 640, 640)):
        fr = torch.zeros((B, h, w, 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.forward(fr); t_enq = time.perf_counter() - t0; torch.cuda.synchronize(); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); eng.forward(fr); torch.cuda.synchronize(); t2 = time.perf_counter() - t0
        st = eng.plan_cache_stats()
        print("%-10s B=%2d %dx%d: first call %7.2f ms (host returns after %7.2f), again %6.2f ms -> the miss costs %6.2f ms; cache %d plans, %d misses"
              % (P["backbone"], B, h, w, t1 * 1e3, t_enq * 1e3, t2 * 1e3, (t1 - t2) * 1e3, st["plans"], st["misses"]), flush=True)
    eng.close()

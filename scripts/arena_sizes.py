"""Arena bytes of the layer plans a handle keeps, per (network shape, batch): ssd_plan_cache_stats after one forward each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ssd_amd, bench
for P in (bench.PARAMS, bench.PARAMS_SHUFFLE):
    eng = ssd_amd.Engine(P, ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), device=0)
    prev = 0
    for B, h, w in ((1, 640, 896), (1, 480, 640), (8, 640, 896), (32, 640, 896), (32, 640, 1024), (64, 640, 640)):
        eng.forward(torch.zeros((B, h, w, 3), dtype=torch.uint8, device="cuda"))
        torch.cuda.synchronize()
        st = eng.plan_cache_stats()
        print("%-10s B=%2d %4dx%-4d -> plan arena %8.1f MB (%.1f MB per frame)" % (P["backbone"], B, h, w, (st["arena_bytes"] - prev) / 2 ** 20, (st["arena_bytes"] - prev) / 2 ** 20 / B), flush=True)
        prev = st["arena_bytes"]
    eng.close()

"""Workload for `rocprofv3 --hip-trace --stats`: one Detector, one warm-up pass over 13 COCO-typical image sizes, then
`cycles` passes over the same mix.  Two traced runs with different `cycles` differ ONLY in steady-state calls: an API whose
call count is the same in both is not called on a plan-cache hit (scripts/mixed_trace_summary.py makes that table).

    rocprofv3 --hip-trace --stats -d OUT -o mix5 -- python3 scripts/mixed_trace.py 5
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ssd_amd
from lat_mixed import SIZES

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 5
P = ssd_amd.load_config(os.path.join(ROOT, "tests", "golden", "config_mobilenet.json"))
det = ssd_amd.Detector(ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), config=P)
rng = np.random.default_rng(0)
frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in SIZES]
for f in frames:
    det(f, 0.15)
n = 0
for _ in range(cycles):
    for f in frames:
        n += len(det(f, 0.15)[2])
print("cycles", cycles, "calls", cycles * len(frames), "detections", n, det.engine.plan_cache_stats())

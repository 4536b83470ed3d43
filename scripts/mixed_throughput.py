"""bench.py's throughput_mixed_sizes leg alone, at several batch caps: a shuffled stream of host frames of 13 COCO-typical sizes, one
Detector call per image against Detector.detect_many.    python scripts/mixed_throughput.py [n_images]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import ssd_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 520
W = ssd_amd.synthetic_weights(bench.PARAMS, seed=0, logits_bias=-7.5)
det = ssd_amd.Detector(W, config=bench.PARAMS)
for mb in (4, 8, 16, 32, 64):
    r = bench.throughput_mixed_sizes(det, n_images=n, max_batch=mb)
    print(json.dumps({k: r[k] for k in ("max_batch", "one_call_per_image_img_s", "detect_many_img_s", "speedup", "results_identical")}), flush=True)
print(json.dumps(r["frames_per_network_shape"]), det.engine.plan_cache_stats())

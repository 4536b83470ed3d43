"""Thread-count sweep of the CPU oracle on this host (run before importing numpy-heavy libs)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = '''
import sys, time, os
sys.path.insert(0, %r)
import numpy as np, ssd_amd
from oracle import graph
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
W = ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5)
img = np.random.default_rng(0).integers(0, 256, (1, 640, 896, 3), dtype=np.uint8)
graph.forward(img, W, P)
ts = []
for _ in range(3):
    t = time.perf_counter(); graph.forward(img, W, P); ts.append(time.perf_counter() - t)
print("threads", os.environ.get("OMP_NUM_THREADS"), "proc_bind", os.environ.get("OMP_PROC_BIND"), "best %%.3f s" %% min(ts))
''' % ROOT
for th, bind in [(8, "close"), (16, "close"), (32, "spread"), (64, "spread"), (128, "spread"), (256, "close"), (64, "false")]:
    env = dict(os.environ, OMP_NUM_THREADS=str(th), OMP_PROC_BIND=bind, OMP_PLACES="cores")
    subprocess.run([sys.executable, "-c", CODE], env=env)

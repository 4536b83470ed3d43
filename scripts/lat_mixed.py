"""Detector.__call__ latency over a cyclic mix of COCO-typical source sizes through ONE Detector
(inference/evaluate_on_COCO.ipynb:125-150 feeds val2017 -- dozens of sizes -- through one session), next to
the same sizes each timed alone.  Prints one JSON object.

    python scripts/lat_mixed.py [config.json] [iterations per size, default 40] [key=value library options ...]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ssd_amd

# (height, width) of frequent COCO val2017 frames, portrait and landscape, plus one at the network's own size
SIZES = [(480, 640), (640, 480), (427, 640), (640, 427), (375, 500), (500, 375), (360, 640), (333, 500),
         (640, 428), (612, 612), (426, 640), (500, 333), (640, 896)]


def pct(v, q):
    return float(np.percentile(np.asarray(v), q))


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".json") else os.path.join(ROOT, "tests", "golden", "config_mobilenet.json")
    rest = [a for a in sys.argv[1:] if not a.endswith(".json")]
    iters = int(rest[0]) if rest and rest[0].isdigit() else 40
    for kv in rest:
        if "=" in kv:
            k, v = kv.split("=")
            ssd_amd.set_option(k, int(v, 0))
    P = ssd_amd.load_config(cfg)
    W = ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5)
    det = ssd_amd.Detector(W, config=P)
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in SIZES]
    out = {"config": os.path.basename(cfg), "sizes": SIZES, "iterations_per_size": iters}

    # each size alone: warm-up 5, then `iters` calls
    alone = {}
    for (h, w), f in zip(SIZES, frames):
        for _ in range(5):
            det(f, 0.15)
        t = []
        for _ in range(iters):
            t0 = time.perf_counter()
            det(f, 0.15)
            t.append((time.perf_counter() - t0) * 1e3)
        nh, nw, _ = ssd_amd.network_input_size(h, w, P["min_dimension"])
        alone["%dx%d" % (h, w)] = {"net": [nh, nw], "p50_ms": round(pct(t, 50), 4), "p95_ms": round(pct(t, 95), 4)}
    out["alone"] = alone
    out["alone_mean_p50_ms"] = round(float(np.mean([v["p50_ms"] for v in alone.values()])), 4)

    # the mix: one warm-up cycle, then `iters` cycles over all sizes (every call a different size than the one before)
    for f in frames:
        det(f, 0.15)
    t = []
    t_all0 = time.perf_counter()
    for _ in range(iters):
        for f in frames:
            t0 = time.perf_counter()
            det(f, 0.15)
            t.append((time.perf_counter() - t0) * 1e3)
    wall = time.perf_counter() - t_all0
    out["mixed"] = {"calls": len(t), "p50_ms": round(pct(t, 50), 4), "p95_ms": round(pct(t, 95), 4), "mean_ms": round(float(np.mean(t)), 4),
                    "img_per_s": round(len(t) / wall, 2)}
    out["mixed_over_alone"] = round(out["mixed"]["mean_ms"] / float(np.mean([v["p50_ms"] for v in alone.values()])), 4)
    stats = getattr(det.engine, "plan_cache_stats", None)
    if stats is not None:
        out["plan_cache"] = stats()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""Interleaved A/B of library options on the batch-1 latency through Detector.__call__ (host frame in, filtered detections out:
what bench.py's latency_batch1 times).  usage: python scripts/ab_det.py name[:key=value[,key=value...]] ...
Every variant is its own Detector; rounds alternate between them so that clock and box drift hit all alike."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ssd_amd
ssd_amd.bind_to_gpu_numa_node(0)
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
W = ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5)
img = np.random.default_rng(0).integers(0, 256, (640, 896, 3), dtype=np.uint8)
dets, ref = [], None
for spec in sys.argv[1:]:
    name, _, opts = spec.partition(":")
    d = ssd_amd.Detector(W, config=P)
    for kv in opts.split(","):
        if kv:
            k, v = kv.split("=")
            if k == "one_call":                   # Detector.__call__ through ssd_detect_host (1) or detect_host + numpy filter (0)
                d.engine.one_call_detect = bool(int(v))
            else:
                d.engine.set_option(k, int(v, 0))
    for _ in range(10):
        out = d(img, score_threshold=0.5)
    got = [np.asarray(t) for t in out]
    if ref is None:
        ref = got
    dets.append((name, d, [], all(np.array_equal(a, b) for a, b in zip(ref, got))))
for r in range(12):
    for name, d, ts, _ in dets:
        for _ in range(3):
            d(img, score_threshold=0.5)
        for _ in range(40):
            t0 = time.perf_counter(); d(img, score_threshold=0.5); ts.append((time.perf_counter() - t0) * 1e3)
for name, d, ts, same in dets:
    print("%-24s p50 %.4f  p10 %.4f  mean %.4f ms   %s" % (name, np.percentile(ts, 50), np.percentile(ts, 10), np.mean(ts),
                                                          "bits = first variant" if same else "BITS DIFFER"))

"""A val2017-like stream through ONE Detector: `n` host frames whose sizes follow COCO's habits (most of them 640 x 480 / 480 x 640 /
640 x 427 / 500 x 375 and their neighbours, a tail of odd sizes and panoramas: ~45 source sizes, ~20 network shapes) in shuffled
order, in chunks of 256 through Detector.detect_many (what coco_eval.evaluate does) -- img/s, and what the plan cache held at the end
(plans, GB, evictions).  usage: python scripts/val_like_stream.py [n_images] [max_batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ssd_amd, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
max_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
P = bench.PARAMS
ssd_amd.bind_to_gpu_numa_node(0)
det = ssd_amd.Detector(ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), config=P)
common = [(480, 640)] * 24 + [(640, 480)] * 7 + [(427, 640)] * 12 + [(640, 427)] * 4 + [(375, 500)] * 6 + [(500, 375)] * 2 + [(426, 640)] * 5 + [(428, 640)] * 3 + \
         [(425, 640)] * 2 + [(424, 640)] * 2 + [(640, 426)] * 2 + [(333, 500)] * 3 + [(500, 333)] * 1 + [(360, 640)] * 2 + [(480, 480)] + [(612, 612)] * 2 + [(640, 640)] * 2
tail = [(400, 600), (512, 640), (640, 512), (478, 640), (359, 640), (500, 400), (640, 359), (281, 500), (500, 281), (213, 640), (640, 213), (300, 400), (240, 320),
        (320, 240), (453, 640), (640, 453), (383, 640), (536, 640), (640, 536), (429, 640), (361, 640), (500, 334), (332, 500), (464, 640), (595, 640), (375, 640)]
rng = np.random.default_rng(0)
sizes = [common[i] for i in rng.integers(0, len(common), n - n // 8)] + [tail[i] for i in rng.integers(0, len(tail), n // 8)]
rng.shuffle(sizes)
frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
shapes = {}
for h, w in sizes:
    k = ssd_amd.network_input_size(h, w, P["min_dimension"])[:2]
    shapes[k] = shapes.get(k, 0) + 1
print("%d frames, %d source sizes, %d network shapes: %s" % (n, len(set(sizes)), len(shapes), dict(sorted(shapes.items(), key=lambda kv: -kv[1]))), flush=True)
for rnd in range(2):
    t0 = time.perf_counter()
    got = 0
    for k in range(0, n, 256):
        got += len(det.detect_many(frames[k:k + 256], score_threshold=0.15, max_batch=max_batch))
    dt = time.perf_counter() - t0
    st = det.engine.plan_cache_stats()
    print("pass %d (%s): %.2f s = %.0f img/s; plan cache: %d plans, %.1f GB of %.1f, %d misses, %d evictions" %
          (rnd, "plans built on the way" if rnd == 0 else "plans cached", dt, got / dt, st["plans"], st["arena_bytes"] / 2 ** 30, st["budget_bytes"] / 2 ** 30, st["misses"], st["evictions"]), flush=True)

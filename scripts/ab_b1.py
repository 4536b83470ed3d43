"""Interleaved A/B of library options on the batch-B forward latency (one process, one box, engines side by side).
usage: python scripts/ab_b1.py B name[:key=value[,key=value...]] ...
Every variant is its own Engine (options set on its handle); rounds alternate between the engines so that clock and box
drift hit all of them alike.  Prints p50 / p10 / mean of forward + synchronize per variant, and checks that every
variant returns the bits of the first one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
B = int(sys.argv[1])
variants = []
for spec in sys.argv[2:]:
    name, _, opts = spec.partition(":")
    variants.append((name, [kv.split("=") for kv in opts.split(",") if kv]))
W = ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5)
img = torch.randint(0, 256, (B, 640, 896, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
engines, ref = [], None
for name, opts in variants:
    e = ssd_amd.Engine(P, W)
    for k, v in opts:
        e.set_option(k, int(v, 0))
    for _ in range(5):
        out = e.forward(img)
    torch.cuda.synchronize()
    got = [t.cpu().numpy() for t in out]
    if ref is None:
        ref = got
    same = all(np.array_equal(a, b) for a, b in zip(ref, got))
    engines.append((name, e, [], same))
ROUNDS, PER = 10, 30
for r in range(ROUNDS):
    for name, e, ts, _ in engines:
        for _ in range(3):
            e.forward(img)
        torch.cuda.synchronize()
        for _ in range(PER):
            t0 = time.perf_counter()
            e.forward(img)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
for name, e, ts, same in engines:
    a = np.array(ts)
    print("%-28s p50 %.4f  p10 %.4f  mean %.4f ms   %s" % (name, np.percentile(a, 50), np.percentile(a, 10), a.mean(),
                                                          "bits = first variant" if same else "BITS DIFFER"), flush=True)

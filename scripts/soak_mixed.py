"""Plan-cache soak: ONE engine serving a shuffled sequence of batch sizes, image sizes and precision modes (a new network shape
builds a plan beside the cached ones; a precision change drops them all), then two host threads on two streams, then the same
two threads under a budget of about two plans (option plan_cache_mb: every other call evicts, which drains the device).
Every result must equal the first result for the same (shape, mode).  usage: python scripts/soak_mixed.py [rounds]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
eng = ssd_amd.Engine(P, ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5))
g = torch.Generator().manual_seed(7)
# (1, 480, 640) and (1, 375, 500) resize to 640 x 896 like the identity frames (one plan for both, another for the identity form);
# (1, 427, 640) and (1, 426, 640) share 640 x 1024 with different resize targets
shapes = [(1, 640, 896), (2, 640, 896), (3, 640, 896), (5, 640, 896), (32, 640, 896), (1, 512, 640), (4, 384, 640), (1, 700, 500),
          (1, 480, 640), (1, 375, 500), (1, 427, 640), (1, 426, 640), (2, 480, 640)]
frames = {s: torch.randint(0, 256, s + (3,), dtype=torch.uint8, generator=g).cuda() for s in shapes}
ref, bad, n = {}, 0, 0
lock = threading.Lock()
rng = np.random.default_rng(3)


def check(key, out):
    global bad, n
    got = [t.cpu().numpy() for t in out]
    with lock:
        n += 1
        if key not in ref:
            ref[key] = got
        elif not all(np.array_equal(a, b) for a, b in zip(ref[key], got)):
            bad += 1
            print("MISMATCH", key, flush=True)


for r in range(ROUNDS):
    for mode in ("f32", "f16x3") if r % 4 == 3 else ("f32",):
        eng.set_precision(mode)
        for i in rng.permutation(len(shapes)):
            s = shapes[i]
            if s[0] == 32 and r % 5:
                continue
            check((s, mode), eng.forward(frames[s]))
            if r % 7 == 0:
                check((s, mode), eng.forward(frames[s]))        # same plan again
eng.set_precision("f32")
print("single thread: %d forwards, %d mismatches, %d distinct (shape, mode) keys; plan cache %s" % (n, bad, len(ref), eng.plan_cache_stats()), flush=True)


SMALL = [s for s in shapes if s[0] <= 5]


def worker(tid, iters, sync_each):
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        pend = []
        for it in range(iters):
            s = SMALL[(tid * 5 + it * (3 + tid)) % len(SMALL)]
            out = eng.forward(frames[s])
            if sync_each:
                torch.cuda.current_stream().synchronize()
                check((s, "f32"), out)
            else:                                   # results read only at the end: forwards of many shapes in flight behind each other
                pend.append((s, out))
        torch.cuda.current_stream().synchronize()
        for s, out in pend:
            check((s, "f32"), out)


for phase, (iters, sync_each, budget) in enumerate([(60, True, 0), (40, False, 0), (60, True, -1)]):
    if budget < 0:                                  # about two batch-1 plans: every other miss evicts (and drains the device)
        st = eng.plan_cache_stats()
        eng.set_option("plan_cache_mb", max(64, int(2.2 * st["arena_bytes"] / max(st["plans"], 1)) >> 20))
    ts = [threading.Thread(target=worker, args=(t, iters, sync_each)) for t in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    print("two threads on two streams, phase %d (%s, budget %s): %d forwards in all, %d mismatches; status %d; plan cache %s"
          % (phase, "sync per call" if sync_each else "no host wait until the end", "default" if budget == 0 else "~2 plans", n, bad,
             eng.status(), eng.plan_cache_stats()), flush=True)
sys.exit(1 if bad else 0)

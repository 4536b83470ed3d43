"""Re-planning soak: ONE engine serving a shuffled sequence of batch sizes, image sizes and precision modes (every
change rebuilds the layer plan on the handle's persistent streams), from two host threads on two streams at the end.
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
shapes = [(1, 640, 896), (2, 640, 896), (3, 640, 896), (5, 640, 896), (32, 640, 896), (1, 512, 640), (4, 384, 640), (1, 700, 500)]
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
print("single thread: %d forwards, %d mismatches, %d distinct (shape, mode) keys" % (n, bad, len(ref)), flush=True)


def worker(tid):
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        for it in range(60):
            s = shapes[(tid * 3 + it) % 4]
            out = eng.forward(frames[s])
            torch.cuda.current_stream().synchronize()
            check((s, "f32"), out)


ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
[t.start() for t in ts]
[t.join() for t in ts]
print("two threads on two streams: %d forwards in all, %d mismatches; status %d" % (n, bad, eng.status()))
sys.exit(1 if bad else 0)

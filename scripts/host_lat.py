"""Where the host side of a batch-1 Detector call goes: each segment of Engine.detect_host timed with a device
synchronisation behind it (attribution, not the pipelined cost), then the whole call as the benchmark times it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
det = ssd_amd.Detector(ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), config=P)
e = det.engine
img = np.random.default_rng(0).integers(0, 256, (640, 896, 3), dtype=np.uint8)
for _ in range(10):
    det(img, score_threshold=0.5)
slot = dict(e._out_slot(1))
_, slot["dev_in"], slot["pin_in"] = e._in_slot((1, 640, 896, 3), index=3, pinned=True)
slot["pin_in_np"] = slot["pin_in"].numpy()
sync = torch.cuda.synchronize
seg = {k: [] for k in ("copyto_pinned", "h2d", "forward", "d2h", "filter")}
for _ in range(50):
    t0 = time.perf_counter(); np.copyto(slot["pin_in_np"], img[None]); t1 = time.perf_counter()
    slot["dev_in"].copy_(slot["pin_in"], non_blocking=True); sync(); t2 = time.perf_counter()
    e.forward(slot["dev_in"], records=slot["block"]); sync(); t3 = time.perf_counter()
    slot["pin_out"].copy_(slot["block"], non_blocking=True); torch.cuda.current_stream().synchronize(); t4 = time.perf_counter()
    b, l, s, n = slot["host"]; k = s[0][:n[0]] > 0.5; _ = b[0][:n[0]][k], l[0][:n[0]][k], s[0][:n[0]][k]; t5 = time.perf_counter()
    for name, a, c in zip(seg, (t0, t1, t2, t3, t4), (t1, t2, t3, t4, t5)):
        seg[name].append((c - a) * 1e6)
for k, v in seg.items():
    print("%-14s p50 %7.1f us" % (k, np.percentile(v, 50)))
ts = []
for _ in range(110):
    t0 = time.perf_counter(); det(img, score_threshold=0.5); ts.append((time.perf_counter() - t0) * 1e3)
print("Detector.__call__ p50 %.3f ms (mean %.3f)" % (np.percentile(ts[10:], 50), np.mean(ts[10:])))
for nc in (1, 2, 3, 4, 6, 8):
    e.set_option("h2d_chunks", nc)           # ssd_forward_host: pieces of the staging copy + upload
    for _ in range(5):
        det(img, score_threshold=0.5)
    ts = []
    for _ in range(110):
        t0 = time.perf_counter(); det(img, score_threshold=0.5); ts.append((time.perf_counter() - t0) * 1e3)
    print("h2d_chunks %d: Detector.__call__ p50 %.3f ms" % (nc, np.percentile(ts[10:], 50)))

"""The COCO harness end to end from files on disk (coco_eval.evaluate: JPEGs decoded by a thread pool one chunk ahead, frames
batched by network shape, records, COCO statistics) on a synthetic val2017-like folder: `n` JPEGs of the 13 COCO-typical sizes.
usage: python scripts/eval_from_disk.py [n_images]"""
import os, sys, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd, bench
from PIL import Image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1040
P = bench.PARAMS
ssd_amd.bind_to_gpu_numa_node(0)
det = ssd_amd.Detector(ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), config=P)
d = tempfile.mkdtemp(prefix="val_like_")
rng = np.random.default_rng(0)
order = rng.permutation(np.repeat(np.arange(len(bench.MIXED_SIZES)), -(-n // len(bench.MIXED_SIZES))))[:n]
images, anns = [], []
t0 = time.perf_counter()
for k, i in enumerate(order):
    h, w = bench.MIXED_SIZES[i]
    # a smooth scene + a little texture: compresses like a photograph (~100 KB), unlike uniform noise
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([127 + 100 * np.sin(xx / (20 + 3 * c) + k) * np.cos(yy / (25 + 2 * c)) for c in range(3)], -1)
    im = np.clip(base + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
    name = "%012d.jpg" % (k + 1)
    Image.fromarray(im).save(os.path.join(d, name), quality=90)
    images.append({"id": k + 1, "file_name": name, "height": h, "width": w})
    anns.append({"id": k + 1, "image_id": k + 1, "category_id": 1, "bbox": [10, 10, w // 3, h // 3], "area": float((w // 3) * (h // 3)), "iscrowd": 0})
cats = [{"id": i + 1 + (i > 10), "name": nm} for i, nm in enumerate(ssd_amd.coco_eval.COCO_NAMES)]
gt = {"images": images, "annotations": anns, "categories": cats}
size_mb = sum(os.path.getsize(os.path.join(d, m["file_name"])) for m in images) / 2 ** 20
print("%d JPEGs, %.1f MB (%.0f KB each) written in %.1f s" % (n, size_mb, size_mb * 1024 / n, time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); [np.asarray(Image.open(os.path.join(d, m["file_name"])).convert("RGB")) for m in images[:100]]; dec = (time.perf_counter() - t0) / 100
print("decode on one thread: %.2f ms per image" % (dec * 1e3), flush=True)
ssd_amd.coco_eval.evaluate(det, gt, d, predictions_json=None)            # builds the plans
for workers in (1, 4, None):
    t0 = time.perf_counter()
    st = ssd_amd.coco_eval.evaluate(det, gt, d, predictions_json=os.path.join(d, "pred.json"), read_workers=workers)
    dt = time.perf_counter() - t0
    print("evaluate, read_workers=%s: %.2f s = %.0f img/s end to end (files -> COCO statistics); AP %.3f" % (workers, dt, n / dt, st[0]), flush=True)
print(det.engine.plan_cache_stats())

"""What in bench.py's flow changes the batch-1 Detector latency measured afterwards?  One Detector; the latency leg
(110 calls, first 10 dropped) after each stage of the benchmark's own sequence.
usage: python scripts/heat_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
det = ssd_amd.Detector(ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), config=P)
e = det.engine
img = np.random.default_rng(0).integers(0, 256, (640, 896, 3), dtype=np.uint8)
img32 = torch.randint(0, 256, (32, 640, 896, 3), dtype=torch.uint8).cuda()


def lat(tag):
    ts = []
    for _ in range(110):
        t0 = time.perf_counter(); det(img, score_threshold=0.5); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-52s p50 %.4f ms" % (tag, np.percentile(ts[10:], 50)), flush=True)


lat("fresh process")
for _ in range(13):
    e.forward(img32)
torch.cuda.synchronize()
lat("after 13 batch-32 forwards")
e.profile_reset(); e.profile_enable(True)
for _ in range(10):
    e.forward(img32)
torch.cuda.synchronize()
e.profile_enable(False); e.profile_read()
lat("after 10 profiled batch-32 forwards")
host = img32.cpu().numpy()
for o in det.detect_stream(host for _ in range(6)):
    pass
lat("after detect_stream of 6 host batches")
e.set_precision("f16x3")
for _ in range(5):
    e.forward(img32)
torch.cuda.synchronize()
e.set_precision("f32")
lat("after a batch-32 leg in mode f16x3 and back")
e.set_precision("f32")
lat("after set_precision('f32') again")

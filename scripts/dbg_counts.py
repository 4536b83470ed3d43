import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}
for bias in (-6.5, -7.0, -7.5, -8.0):
    W = ssd_amd.synthetic_weights(P, seed=0, logits_bias=bias)
    e = ssd_amd.Engine(P, W)
    g = torch.Generator().manual_seed(1234)
    img = torch.randint(0, 256, (2, 640, 896, 3), dtype=torch.uint8, generator=g).cuda()
    out = e.forward(img)
    lg = e.get_tensor("class_predictions").reshape(2, -1, 80)
    thr = np.log(0.15 / 0.85)
    cnt = (lg > thr).sum(axis=1)
    print("bias", bias, "logit mean %.2f std %.2f" % (lg.mean(), lg.std()), "cands/img", cnt.sum(axis=1), "per-class max", cnt.max(axis=1),
          "classes>512:", (cnt > 512).sum(axis=1), ">8192:", (cnt > 8192).sum(axis=1), "num_boxes", out[3].cpu().numpy())
    print("   sorted top counts img0:", np.sort(cnt[0])[::-1][:12])
    e.close()

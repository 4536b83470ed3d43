"""Timing of the post-processing on adversarial candidate lists: one (image, class) list of ~8 000 candidates made of
K clusters of mutually overlapping boxes.  K < max_boxes_per_class: both top-score trials fail (they keep K < 25 boxes) and the
block runs the exact global-memory rounds (`nms_global`, postprocess.hip) -- K rounds of one pass over the list each; K >= 25: the
first trial settles it.  Against the benchmark's typical frame.
usage: python scripts/nms_worst_case.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ssd_amd

anc = np.asarray(ssd_amd.AnchorGenerator()(640, 896), np.float32)
N, C = anc.shape[0], 80
rng = np.random.default_rng(0)


def case(n_clusters, per_cluster, spread=0.0):
    logits = np.full((1, N, C), -9.0, np.float32)
    codes = np.zeros((1, N, 4), np.float32)
    idx = rng.permutation(53760)[:n_clusters * per_cluster]              # level-3 anchors
    ha, wa = anc[idx, 2] - anc[idx, 0], anc[idx, 3] - anc[idx, 1]
    cya, cxa = anc[idx, 0] + 0.5 * ha, anc[idx, 1] + 0.5 * wa
    cl = np.arange(idx.size) // per_cluster
    cy = 0.1 + 0.8 * (cl // 6) / 5.0 + spread * rng.standard_normal(idx.size) * 0.002      # clusters on a 6 x 5 grid, far apart
    cx = 0.1 + 0.8 * (cl % 6) / 6.0 + spread * rng.standard_normal(idx.size) * 0.002
    codes[0, idx, 0] = 10.0 * (cy - cya) / ha
    codes[0, idx, 1] = 10.0 * (cx - cxa) / wa
    codes[0, idx, 2] = 5.0 * np.log(0.08 / ha)
    codes[0, idx, 3] = 5.0 * np.log(0.08 / wa)
    logits[0, idx, 7] = rng.uniform(0.0, 4.0, idx.size).astype(np.float32)
    return codes, logits


def run(name, codes, logits):
    c, l, a = (torch.from_numpy(x).cuda() for x in (codes, logits, anc))
    for _ in range(3):
        out = ssd_amd.batch_multiclass_non_max_suppression(c, a, l, 0.15, 0.6, 25)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ssd_amd.batch_multiclass_non_max_suppression(c, a, l, 0.15, 0.6, 25)
    e1.record()
    torch.cuda.synchronize()
    print("%-64s %8.1f us per call   kept %d" % (name, e0.elapsed_time(e1) * 1000 / 20, int(out[3][0])), flush=True)


run("no candidates", *case(0, 1))
run("24 clusters x 340 = 8 160 candidates, 24 kept (trials fail)", *case(24, 340))
run("12 clusters x 680 = 8 160 candidates, 12 kept (trials fail)", *case(12, 680))
run("1 cluster x 8 160 candidates, 1 kept (trials fail, one round)", *case(1, 8160))
run("30 clusters x 272 = 8 160 candidates, 25 kept (first trial settles)", *case(30, 272))
run("24 clusters x 1 500 = 36 000 candidates, 24 kept", *case(24, 1500))
run("24 clusters x 80 = 1 920 candidates, 24 kept (four waves, registers)", *case(24, 80))

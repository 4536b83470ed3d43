"""COCO bounding-box AP / AR without pycocotools -- the metric of the reference's README.md:3-10 (AP 31.7, AP@0.5 49.8 on val2017).

The reference computes it in inference/evaluate_on_COCO.ipynb cell 17 with `COCOeval(cocoGt, cocoDt, iouType='bbox')` from
cocoapi's PythonAPI (pycocotools: third party, appended to sys.path by cell 0, not vendored by the reference, not installed
here).  This module restates that published algorithm (cocoeval.py `evaluate` / `computeIoU` / `evaluateImg` / `accumulate` /
`summarize`, maskApi.c `bbIou`, coco.py `loadRes` for bbox results) so that `coco_eval.evaluate` ends in the same twelve numbers:

  * per (image, category): detections by descending score (stable), at most 100; IoU of xywh boxes, with a CROWD groundtruth
    box the union is the detection's own area;
  * per area range (all / small < 32^2 / medium / large > 96^2): groundtruth outside the range or crowd is "ignore", sorted behind
    the others; for every IoU threshold 0.50:0.05:0.95 each detection, best score first, takes the still-free groundtruth box of
    the highest IoU >= threshold (a crowd box stays free; once a regular match is held an ignore box cannot replace it); a
    detection matched to an ignore box, or unmatched with its own area outside the range, is ignored;
  * per (category, area range, maxDets 1 / 10 / 100): detections of all images merged by descending score (stable), cumulative
    TP / FP over the non-ignored ones, recall = TP / #non-ignored groundtruth, precision made non-increasing from the right and
    sampled at the 101 recall points 0:0.01:1 (0 beyond the reached recall);
  * AP = mean over thresholds, recall points and categories with groundtruth; AP50 / AP75; APs / APm / APl; AR@1 / @10 / @100;
    ARs / ARm / ARl.
Groundtruth `ignore` = `iscrowd` (bbox evaluation).  Known answers and an independently written brute-force AP are in
tests/test_host.py; no pycocotools run is reachable offline to compare with."""
import json

import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
REC_THRS = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
MAX_DETS = (1, 10, 100)
AREA_RNG = ((0 ** 2, 1e5 ** 2), (0 ** 2, 32 ** 2), (32 ** 2, 96 ** 2), (96 ** 2, 1e5 ** 2))
AREA_LBL = ("all", "small", "medium", "large")
STAT_NAMES = ("AP", "AP50", "AP75", "APs", "APm", "APl", "AR1", "AR10", "AR100", "ARs", "ARm", "ARl")
THR_L = [min(float(t), 1 - 1e-10) for t in IOU_THRS]        # cocoeval.py evaluateImg: iou = min([t, 1 - 1e-10])


def bbox_iou(dt, gt, iscrowd):
    """IoU matrix [D, G] of xywh boxes (maskApi.c bbIou): 0 unless the overlap has positive width and height; for a crowd
    groundtruth box the denominator is the detection's area."""
    dt = np.asarray(dt, np.float64).reshape(-1, 4)
    gt = np.asarray(gt, np.float64).reshape(-1, 4)
    crowd = np.asarray(iscrowd, bool).reshape(-1)
    w = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    h = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    ok = (w > 0) & (h > 0)
    inter = np.where(ok, w * h, 0.0)
    da = (dt[:, 2] * dt[:, 3])[:, None]
    ga = (gt[:, 2] * gt[:, 3])[None, :]
    union = np.where(crowd[None, :], da, da + ga - inter)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(ok, inter / union, 0.0)


class CocoBoxEval:
    """groundtruth: the annotation file's dict (or its path): 'images', 'annotations' (image_id, category_id, bbox xywh, area,
    iscrowd), 'categories'.  results: the list of result dicts of evaluate_on_COCO.ipynb cell 10 (image_id, category_id, bbox,
    score), or the path of its JSON.  img_ids / cat_ids default to all of the annotation file (cell 17 sets exactly those)."""

    def __init__(self, groundtruth, results, img_ids=None, cat_ids=None):
        gt = json.load(open(groundtruth)) if isinstance(groundtruth, str) else groundtruth
        dt = json.load(open(results)) if isinstance(results, str) else results
        self.img_ids = sorted(set(img_ids if img_ids is not None else [im["id"] for im in gt["images"]]))
        self.cat_ids = sorted(set(cat_ids if cat_ids is not None else [c["id"] for c in gt["categories"]]))
        known = set(im["id"] for im in gt["images"])
        imgs, cats = set(self.img_ids), set(self.cat_ids)
        self._gts, self._dts = {}, {}
        for a in gt["annotations"]:
            if a["image_id"] in imgs and a["category_id"] in cats:
                crowd = int(a.get("iscrowd", 0))
                area = float(a["area"]) if "area" in a else float(a["bbox"][2] * a["bbox"][3])
                self._gts.setdefault((a["image_id"], a["category_id"]), []).append((list(a["bbox"]), area, crowd))
        for k, d in enumerate(dt):
            if d["image_id"] not in known:          # coco.loadRes asserts this
                raise ValueError("results[%d] names image_id %r, which the annotation file does not have" % (k, d["image_id"]))
            if d["image_id"] in imgs and d["category_id"] in cats:
                bb = list(d["bbox"])
                self._dts.setdefault((d["image_id"], d["category_id"]), []).append((bb, float(bb[2] * bb[3]), float(d["score"])))
        self.eval_imgs = None
        self.precision = self.recall = self.stats = None

    # -- cocoeval.py evaluate(): computeIoU + evaluateImg for every (category, area range, image)
    def evaluate(self):
        T = len(IOU_THRS)
        maxdet = MAX_DETS[-1]
        self.eval_imgs = {}
        for cat in self.cat_ids:
            for img in self.img_ids:
                gts = self._gts.get((img, cat), [])
                dts = self._dts.get((img, cat), [])
                if not gts and not dts:
                    continue
                scores = np.array([d[2] for d in dts], np.float64)
                order = np.argsort(-scores, kind="mergesort")[:maxdet]
                dbox = [dts[i][0] for i in order]
                darea = np.array([dts[i][1] for i in order], np.float64)
                dscore = scores[order]
                gcrowd = np.array([g[2] for g in gts], np.int64)
                garea = np.array([g[1] for g in gts], np.float64)
                ious_all = bbox_iou(dbox, [g[0] for g in gts], gcrowd) if gts and dts else np.zeros((len(dbox), len(gts)))
                for ai, (lo, hi) in enumerate(AREA_RNG):
                    if not gts:            # (most cells of a real run: detections of a category the image does not have)
                        oor = (darea < lo) | (darea > hi)
                        self.eval_imgs[(cat, ai, img)] = (dscore, np.zeros((T, len(dbox)), bool), np.broadcast_to(oor, (T, len(dbox))), np.zeros(0, bool))
                        continue
                    gig = (gcrowd != 0) | (garea < lo) | (garea > hi)
                    gind = np.argsort(gig.astype(np.int64), kind="mergesort")
                    gig_s, crowd_s = gig[gind], gcrowd[gind]
                    ious = ious_all[:, gind]
                    D, G = len(dbox), len(gts)
                    dtm = np.zeros((T, D), bool)
                    dtig = np.zeros((T, D), bool)
                    if G and D:
                        # (plain Python lists inside the loops: indexing numpy scalars costs more than the comparisons)
                        iou_l, ig_l, cr_l = ious.tolist(), gig_s.tolist(), [bool(c) for c in crowd_s.tolist()]
                        for ti, t in enumerate(THR_L):
                            gtm = [False] * G
                            row_m, row_ig = dtm[ti], dtig[ti]
                            for di in range(D):
                                best = t
                                m = -1
                                row = iou_l[di]
                                for gi in range(G):
                                    if gtm[gi] and not cr_l[gi]:
                                        continue                       # already matched, and not a crowd
                                    if m > -1 and not ig_l[m] and ig_l[gi]:
                                        break                          # a regular match is held: ignore boxes (sorted last) cannot take it
                                    if row[gi] < best:
                                        continue
                                    best = row[gi]
                                    m = gi
                                if m == -1:
                                    continue
                                row_ig[di] = ig_l[m]
                                row_m[di] = True
                                gtm[m] = True
                    out_of_range = (darea < lo) | (darea > hi)
                    dtig = dtig | (~dtm & out_of_range[None, :])
                    self.eval_imgs[(cat, ai, img)] = (dscore, dtm, dtig, gig_s)
        return self

    # -- cocoeval.py accumulate()
    def accumulate(self):
        if self.eval_imgs is None:
            self.evaluate()
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(self.cat_ids), len(AREA_RNG), len(MAX_DETS)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        eps = np.spacing(1)
        for k, cat in enumerate(self.cat_ids):
            for a in range(A):
                E = [self.eval_imgs[(cat, a, img)] for img in self.img_ids if (cat, a, img) in self.eval_imgs]
                if not E:
                    continue
                gig = np.concatenate([e[3] for e in E])
                npig = int(np.count_nonzero(~gig))
                if npig == 0:
                    continue
                for m, maxdet in enumerate(MAX_DETS):
                    sc = np.concatenate([e[0][:maxdet] for e in E])
                    inds = np.argsort(-sc, kind="mergesort")
                    dtm = np.concatenate([e[1][:, :maxdet] for e in E], axis=1)[:, inds]
                    dig = np.concatenate([e[2][:, :maxdet] for e in E], axis=1)[:, inds]
                    tp_sum = np.cumsum(dtm & ~dig, axis=1).astype(np.float64)
                    fp_sum = np.cumsum(~dtm & ~dig, axis=1).astype(np.float64)
                    for t in range(T):
                        tp, fp = tp_sum[t], fp_sum[t]
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + eps)
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = np.maximum.accumulate(pr[::-1])[::-1] if nd else pr      # non-increasing from the right
                        q = np.zeros(R)
                        pos = np.searchsorted(rc, REC_THRS, side="left")
                        ok = pos < nd
                        q[ok] = pr[pos[ok]]
                        precision[t, :, k, a, m] = q
        self.precision, self.recall = precision, recall
        return self

    # -- cocoeval.py summarize()
    def summarize(self, out=None):
        if self.precision is None:
            self.accumulate()

        def stat(ap, iou=None, area="all", maxdet=100):
            a, m = AREA_LBL.index(area), MAX_DETS.index(maxdet)
            s = self.precision[:, :, :, a, m] if ap else self.recall[:, :, a, m]
            if iou is not None:
                s = s[np.where(np.isclose(IOU_THRS, iou))[0]]
            s = s[s > -1]
            return float(np.mean(s)) if s.size else -1.0
        self.stats = np.array([stat(1), stat(1, 0.5), stat(1, 0.75), stat(1, area="small"), stat(1, area="medium"), stat(1, area="large"),
                               stat(0, maxdet=1), stat(0, maxdet=10), stat(0), stat(0, area="small"), stat(0, area="medium"),
                               stat(0, area="large")])
        if out is not None:
            rows = [("Average Precision", "0.50:0.95", "all", 100), ("Average Precision", "0.50", "all", 100), ("Average Precision", "0.75", "all", 100),
                    ("Average Precision", "0.50:0.95", "small", 100), ("Average Precision", "0.50:0.95", "medium", 100),
                    ("Average Precision", "0.50:0.95", "large", 100), ("Average Recall", "0.50:0.95", "all", 1), ("Average Recall", "0.50:0.95", "all", 10),
                    ("Average Recall", "0.50:0.95", "all", 100), ("Average Recall", "0.50:0.95", "small", 100),
                    ("Average Recall", "0.50:0.95", "medium", 100), ("Average Recall", "0.50:0.95", "large", 100)]
            for (title, iou, area, md), v in zip(rows, self.stats):
                out.write(" %-18s (%s) @[ IoU=%-9s | area=%6s | maxDets=%3d ] = %0.3f\n" % (title, "AP" if "Precision" in title else "AR", iou, area, md, v))
        return self.stats


def evaluate_boxes(groundtruth, results, img_ids=None, cat_ids=None, out=None):
    """The twelve COCO statistics (STAT_NAMES order) of `results` against `groundtruth`."""
    return CocoBoxEval(groundtruth, results, img_ids, cat_ids).evaluate().accumulate().summarize(out)

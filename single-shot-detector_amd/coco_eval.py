"""COCO val2017 evaluation harness around `Detector` -- the product-side mirror of the
reference's inference/evaluate_on_COCO.ipynb (cells 6-17) -- and the VOC-style AP@IoU
self-check of the reference's metrics.py:156-282 (`Evaluator`, `evaluate_detector`).
No COCO images or annotations exist offline, so the tests run the harness on synthetic scenes;
`evaluate` needs the dataset on disk and nothing else: the COCO statistics of cell 17 come from
coco_metric.py, this build's restatement of pycocotools' COCOeval for boxes.

Label ids: the detector's integer label i is line i of the reference's data/coco_labels.txt,
which is the standard 80-name COCO order (COCO_NAMES below); the official category ids come
from the annotation file (name -> id), exactly as cell 7 builds `integer_to_coco_id`.
"""
import json

import numpy as np

COCO_NAMES = [
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
    "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant",
    "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard",
    "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard", "tennis racket", "bottle",
    "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple", "sandwich", "orange", "broccoli",
    "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch", "potted plant", "bed", "dining table", "toilet",
    "tv", "laptop", "mouse", "remote", "keyboard", "cell phone", "microwave", "oven", "toaster", "sink",
    "refrigerator", "book", "clock", "vase", "scissors", "teddy bear", "hair drier", "toothbrush"]


def integer_to_coco_id(categories):
    """categories: coco.loadCats(coco.getCatIds()) -> {detector label: official category id}
    (evaluate_on_COCO.ipynb cell 7)."""
    labels = {n: i for i, n in enumerate(COCO_NAMES)}
    return {labels[c["name"]]: c["id"] for c in categories}


def detection_records(detector, image, image_id, label_to_coco_id, score_threshold=0.15, detections=None):
    """One image -> COCO result dicts, exactly as cell 10: boxes scaled to pixels of the
    ORIGINAL image, x, y = int(xmin), int(ymin); w, h = int(xmax - xmin), int(ymax - ymin).
    `detections` = (boxes, labels, scores) already computed for this image (a batched run), else the detector is called."""
    height, width, _ = image.shape
    boxes, labels, scores = detections if detections is not None else detector(image, score_threshold=score_threshold)
    scaler = np.array([height, width, height, width], dtype="float32")
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4) * scaler
    # cell 10 per detection: x, y = int(xmin), int(ymin); w, h = int(xmax - xmin), int(ymax - ymin) on float32 values -- the same
    # float32 differences and truncations toward zero, for all rows at once
    x, y = boxes[:, 1].astype(np.int64).tolist(), boxes[:, 0].astype(np.int64).tolist()
    w, h = (boxes[:, 3] - boxes[:, 1]).astype(np.int64).tolist(), (boxes[:, 2] - boxes[:, 0]).astype(np.int64).tolist()
    cats = [int(label_to_coco_id[int(l)]) for l in np.asarray(labels).reshape(-1).tolist()]
    sc = np.asarray(scores, np.float32).reshape(-1).astype(np.float64).tolist()         # float(np.float32) per element
    image_id = int(image_id)
    return [{"image_id": image_id, "category_id": c, "bbox": [xi, yi, wi, hi], "score": s} for c, xi, yi, wi, hi, s in zip(cats, x, y, w, h, sc)]


def detection_records_many(detector, images, image_ids, label_to_coco_id, score_threshold=0.15, max_batch=32):
    """The same for a list of images of any sizes, with the detector's batched form when it has one (Detector.detect_many: images
    grouped by the size the network sees, frames of different source sizes in one batch; each result bit for bit the single call's)."""
    many = getattr(detector, "detect_many", None)
    dets = many(images, score_threshold=score_threshold, max_batch=max_batch) if many else [None] * len(images)
    out = []
    for image, image_id, d in zip(images, image_ids, dets):
        out += detection_records(detector, image, image_id, label_to_coco_id, score_threshold, detections=d)
    return out


def evaluate(detector, annotations_json, images_dir, read_image=None, predictions_json="coco_predictions.json", out=None,
             score_threshold=0.15, max_batch=32, read_workers=None):
    """Cells 4-17 end to end: every image of the annotation file through the detector (score_threshold 0.15, cell 10), the
    results written as `predictions_json` (cell 11), then the twelve COCO box statistics (cell 17: COCOeval over all image and
    category ids) -- computed by coco_metric.py, this build's restatement of pycocotools' COCOeval (not installed here, not
    vendored by the reference).  `annotations_json`: path of instances_val2017.json or its dict; `read_image(path) -> uint8 RGB
    ndarray`, default PIL (the notebook reads with cv2 and converts BGR -> RGB: the same array when both decoders are built on the
    same libjpeg, which is the usual case but nothing here can check; pass the notebook's reader to be sure); `out`: a stream for
    the summary table; `read_workers`: threads that read and decode images (default min(16, CPUs): a JPEG decodes in ~5 ms on
    one core, the detector takes ~1.4 ms per image in batches -- the next chunk of files is decoded while this one is detected).
    Returns the statistics in coco_metric.STAT_NAMES order (AP, AP50, AP75, APs, APm, APl, AR1, ...)."""
    import os
    from . import coco_metric
    if read_image is None:
        from PIL import Image

        def read_image(path):
            return np.asarray(Image.open(path).convert("RGB"))
    gt = json.load(open(annotations_json)) if isinstance(annotations_json, str) else annotations_json
    mapping = integer_to_coco_id(gt["categories"])
    metas = sorted(gt["images"], key=lambda m: m["id"])
    results = []
    chunk = 256           # images read, then detected as batches grouped by network shape (the notebook's loop, one image per sess.run, at batch throughput)
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, int(read_workers if read_workers is not None else min(16, os.cpu_count() or 1)))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        def start(k):      # (PIL and cv2 release the interpreter lock while they decode)
            return [pool.submit(read_image, os.path.join(images_dir, m["file_name"])) for m in metas[k:k + chunk]]
        ahead = start(0)
        for k in range(0, len(metas), chunk):
            part = metas[k:k + chunk]
            images = [f.result() for f in ahead]
            ahead = start(k + chunk)
            results += detection_records_many(detector, images, [m["id"] for m in part], mapping, score_threshold, max_batch)
    if predictions_json:
        with open(predictions_json, "w") as f:
            json.dump(results, f)
    return coco_metric.evaluate_boxes(gt, results, out=out)


# ----------------------------------------------------------------------------- VOC-style AP self-check
def _iou_matrix(det, gt):
    """IoU of every detection [n,4] with every groundtruth box [m,4] (ymin, xmin, ymax, xmax) as metrics.py:234-246
    computes it: 0 unless both the overlap's width and height are positive."""
    w = np.minimum(det[:, None, 3], gt[None, :, 3]) - np.maximum(det[:, None, 1], gt[None, :, 1])
    h = np.minimum(det[:, None, 2], gt[None, :, 2]) - np.maximum(det[:, None, 0], gt[None, :, 0])
    inter = np.where((w > 0) & (h > 0), w * h, 0.0)
    a_d = (det[:, 3] - det[:, 1]) * (det[:, 2] - det[:, 0])
    a_g = (gt[:, 3] - gt[:, 1]) * (gt[:, 2] - gt[:, 0])
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / (a_d[:, None] + a_g[None, :] - inter)
    return np.where(inter > 0, iou, 0.0)


def average_precision(groundtruth, detections, iou_threshold=0.5):
    """One class.  Semantics of the reference's `evaluate_detector` (metrics.py:156-213):
      groundtruth: {image: float array [m,4]};  detections: list of (image, box[4], confidence).
      * detections are visited in decreasing confidence (stable for ties, like list.sort);
      * a detection's partner is the groundtruth box of its image with the largest IoU (the FIRST such box when IoUs
        tie, `iou > max_iou` in `match`, metrics.py:249-264); it is a true positive iff that IoU >= iou_threshold and the
        box has no earlier partner (a second detection of a matched box is a false positive);
      * precision[k] = TP / (k+1), recall[k] = TP / max(#groundtruth, 1); AP = sum precision[k] * (recall[k] - recall[k-1])
        (metrics.py:274-282); best threshold = the confidence maximising P*R*(1 - |P - R|) (:216-231).
    Returns the reference's seven values."""
    num_gt = max(sum(len(b) for b in groundtruth.values()), 1)
    conf = np.array([d[2] for d in detections], dtype=np.float64)
    order = np.argsort(-conf, kind="stable")
    matched = {img: np.zeros(len(b), bool) for img, b in groundtruth.items()}
    tp = np.zeros(len(detections), bool)
    iou_sum = 0.0
    for k, di in enumerate(order):
        img, box, _c = detections[di]
        gt = groundtruth.get(img)
        if gt is None or len(gt) == 0:
            continue
        iou = _iou_matrix(np.asarray(box, np.float64)[None], np.asarray(gt, np.float64))[0]
        best = int(np.argmax(iou))                      # first maximum, like the strict `>` scan
        if iou[best] > 0 and iou[best] >= iou_threshold and not matched[img][best]:
            matched[img][best] = True
            tp[k] = True
            iou_sum += float(iou[best])
    ctp = np.cumsum(tp)
    n = np.arange(1, len(detections) + 1)
    precision = ctp / np.maximum(n, 1)
    recall = ctp / num_gt
    ap = float(np.sum(precision * np.diff(np.concatenate([[0.0], recall])))) if len(detections) else 0.0
    if len(detections):
        best_i = int(np.argmax(precision * recall * (1.0 - np.abs(precision - recall))))
        best = (float(conf[order][best_i]), float(precision[best_i]), float(recall[best_i]))
    else:
        best = (0.0, 0.0, 0.0)
    ntp = int(ctp[-1]) if len(detections) else 0
    return {"AP": ap, "precision": best[1], "recall": best[2], "best_threshold": best[0],
            "mean_iou_for_TP": iou_sum / max(ntp, 1), "total_FP": len(detections) - ntp, "total_FN": num_gt - ntp}


class Evaluator:
    """Mirror of metrics.py:15-131 without the TF plumbing: detections and groundtruth are kept per label, `evaluate`
    returns the per-label metrics and, for more than one class, their mean AP under the key 'mAP'."""

    def __init__(self, num_classes):
        assert num_classes > 0
        self.num_classes = num_classes
        self.initialize()

    def initialize(self):
        self.detections = {label: [] for label in range(self.num_classes)}
        self.groundtruth = {label: {} for label in range(self.num_classes)}
        self.unique_image_id = 0

    def add_groundtruth(self, image_name, boxes, labels):
        for box, label in zip(np.asarray(boxes, np.float64).reshape(-1, 4), np.asarray(labels).reshape(-1)):
            self.groundtruth[int(label)].setdefault(image_name, []).append(box)

    def add_detections(self, image_name, boxes, labels, scores):
        for box, label, score in zip(np.asarray(boxes, np.float64).reshape(-1, 4), np.asarray(labels).reshape(-1),
                                     np.asarray(scores).reshape(-1)):
            self.detections[int(label)].append((image_name, box, float(score)))

    def add_image(self, gt_boxes, gt_labels, boxes, labels, scores):
        """One image's groundtruth and the detector's (already num_boxes-trimmed) outputs: `update_op_func`, metrics.py:55-59."""
        name = str(self.unique_image_id)
        self.unique_image_id += 1
        self.add_groundtruth(name, gt_boxes, gt_labels)
        self.add_detections(name, boxes, labels, scores)

    def evaluate(self, iou_threshold=0.5):
        self.metrics = {}
        for label in range(self.num_classes):
            gt = {img: np.array(b, np.float64).reshape(-1, 4) for img, b in self.groundtruth[label].items()}
            self.metrics[label] = average_precision(gt, self.detections[label], iou_threshold)
        if self.num_classes > 1:
            self.metrics["mAP"] = float(np.mean([self.metrics[label]["AP"] for label in range(self.num_classes)]))
        return self.metrics

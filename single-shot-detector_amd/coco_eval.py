"""COCO val2017 evaluation harness around `Detector` -- the product-side mirror of the
reference's inference/evaluate_on_COCO.ipynb (cells 6-17).  No COCO images, annotations or
pycocotools exist offline, so only the record construction is exercised by the tests;
`evaluate` needs pycocotools and the dataset on disk.

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


def detection_records(detector, image, image_id, label_to_coco_id, score_threshold=0.15):
    """One image -> COCO result dicts, exactly as cell 10: boxes scaled to pixels of the
    ORIGINAL image, x, y = int(xmin), int(ymin); w, h = int(xmax - xmin), int(ymax - ymin)."""
    height, width, _ = image.shape
    boxes, labels, scores = detector(image, score_threshold=score_threshold)
    scaler = np.array([height, width, height, width], dtype="float32")
    boxes = boxes * scaler
    out = []
    for i in range(len(boxes)):
        ymin, xmin, ymax, xmax = boxes[i]
        x, y = int(xmin), int(ymin)
        w, h = int(xmax - xmin), int(ymax - ymin)
        out.append({"image_id": int(image_id), "category_id": int(label_to_coco_id[int(labels[i])]),
                    "bbox": [x, y, w, h], "score": float(scores[i])})
    return out


def evaluate(detector, annotations_json, images_dir, read_image, predictions_json="coco_predictions.json"):
    """Cells 4-17 end to end; `read_image(path) -> uint8 RGB ndarray`.  Needs pycocotools."""
    import os
    from pycocotools.coco import COCO
    from pycocotools.cocoeval import COCOeval
    coco = COCO(annotations_json)
    mapping = integer_to_coco_id(coco.loadCats(coco.getCatIds()))
    img_ids = coco.getImgIds()
    results = []
    for image_id in img_ids:
        meta = coco.loadImgs(image_id)[0]
        image = read_image(os.path.join(images_dir, meta["file_name"]))
        results += detection_records(detector, image, meta["id"], mapping)
    with open(predictions_json, "w") as f:
        json.dump(results, f)
    ev = COCOeval(cocoGt=coco, cocoDt=coco.loadRes(predictions_json), iouType="bbox")
    ev.params.imgIds = img_ids
    ev.params.catIds = coco.getCatIds()
    ev.evaluate()
    ev.accumulate()
    ev.summarize()
    return ev.stats

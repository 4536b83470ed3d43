"""-m gpu: SURVEY 8(f) row 4 -- the COCO record construction (inference/evaluate_on_COCO.ipynb cells 10/17) and the VOC-style
AP self-check (metrics.py:156-282) driven by the REAL Detector on synthetic images, against the same harness driven by the CPU
oracle; and the SSD mirror with the reference's constructor signature on images whose size is not a multiple of 128."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PARAMS = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
          "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 128}


class OracleDetector:
    """inference/detector.py's __call__ on top of the CPU oracle (test side only)."""

    def __init__(self, graph, Wt, params):
        self.graph, self.Wt, self.params = graph, Wt, params

    def __call__(self, image, score_threshold=0.1):
        return self.graph.detector_call(self.graph.forward(image[None], self.Wt, self.params), score_threshold)


def test_coco_records_and_voc_ap_through_the_real_detector(cuda, ssd, oracle_graph):
    Wt = ssd.synthetic_weights(PARAMS, seed=3, logits_bias=-4.0)
    det = ssd.Detector(Wt, config=PARAMS)
    ora = OracleDetector(oracle_graph, Wt, ssd.load_config(PARAMS))
    cats = [{"id": i + 1 + (i > 10), "name": n} for i, n in enumerate(ssd.coco_eval.COCO_NAMES)]     # ids with a gap, like COCO's
    mapping = ssd.coco_eval.integer_to_coco_id(cats)
    rng = np.random.default_rng(17)
    ev_gpu, ev_self = ssd.coco_eval.Evaluator(80), ssd.coco_eval.Evaluator(80)
    total = 0
    for image_id, shape in enumerate([(128, 128, 3), (100, 151, 3), (300, 128, 3), (97, 203, 3), (480, 640, 3)]):
        image = rng.integers(0, 256, shape, dtype=np.uint8)
        recs = ssd.coco_eval.detection_records(det, image, image_id, mapping)
        want = ssd.coco_eval.detection_records(ora, image, image_id, mapping)
        assert recs == want and len(recs) > 0, shape         # the integers AND the float scores of every record identical
        for r in recs:
            x, y, w, h = r["bbox"]
            # (x, y may exceed the image: the graph clips to the PADDED frame, then divides by box_scaler, model.py:67-68;
            #  an untrained net detects things in the pad band)
            assert all(isinstance(v, int) for v in r["bbox"]) and x >= 0 and y >= 0 and w >= 0 and h >= 0
        total += len(recs)
        # VOC-style self-check: the oracle's detections above 0.3 as "groundtruth", the GPU detector's as detections
        gb, gl, gs = ora(image, 0.3)
        b, l, s = det(image, 0.15)
        ev_gpu.add_image(gb, gl, b, l, s)
        ev_self.add_image(gb, gl, gb, gl, gs)
    # the harness's batched form (Detector.detect_many: images grouped by network shape, frames of different sizes in one batch):
    # the same records, in the same order, as one call per image -- for the GPU detector and for a detector without detect_many
    shapes = [(128, 128, 3), (100, 151, 3), (300, 128, 3), (97, 203, 3), (128, 200, 3), (64, 100, 3), (100, 151, 3)]
    images = [rng.integers(0, 256, sh, dtype=np.uint8) for sh in shapes]
    ids = list(range(100, 100 + len(images)))
    single = []
    for im, i in zip(images, ids):
        single += ssd.coco_eval.detection_records(det, im, i, mapping)
    assert ssd.coco_eval.detection_records_many(det, images, ids, mapping, max_batch=4) == single and len(single) > 50
    assert ssd.coco_eval.detection_records_many(ora, images[:2], ids[:2], mapping) == [r for r in single if r["image_id"] in ids[:2]]
    m_gpu, m_self = ev_gpu.evaluate(), ev_self.evaluate()
    assert total > 100
    assert m_self["mAP"] == np.mean([1.0 if len(ev_self.detections[c]) else 0.0 for c in range(80)])
    # every groundtruth box is found (recall 1: the detector's output contains it bit for bit), at full precision down to it
    for c in range(80):
        n_gt = sum(len(v) for v in ev_gpu.groundtruth[c].values())
        if n_gt:
            assert m_gpu[c]["total_FN"] == 0, c
            assert m_gpu[c]["AP"] > 0.0


def test_coco_evaluation_end_to_end_without_pycocotools(cuda, ssd, oracle_graph, tmp_path):
    """evaluate_on_COCO.ipynb cells 4-17 on a synthetic "dataset" on disk (images of five sizes as PNG, an annotation file whose
    objects are the oracle's detections above 0.3): the real Detector through coco_eval.evaluate -- files read, batched by network
    shape, records written, the twelve COCO statistics from coco_metric.py -- equals the same harness driven by the CPU oracle,
    number for number; every annotated object is found ahead of every unannotated detection, so AP = AR@100 = 1."""
    import json
    from PIL import Image
    Wt = ssd.synthetic_weights(PARAMS, seed=3, logits_bias=-4.0)
    det = ssd.Detector(Wt, config=PARAMS)
    ora = OracleDetector(oracle_graph, Wt, ssd.load_config(PARAMS))
    cats = [{"id": i + 1 + (i > 10), "name": n} for i, n in enumerate(ssd.coco_eval.COCO_NAMES)]
    mapping = ssd.coco_eval.integer_to_coco_id(cats)
    rng = np.random.default_rng(23)
    images, anns = [], []
    for k, shape in enumerate([(128, 128, 3), (100, 151, 3), (300, 128, 3), (97, 203, 3), (128, 200, 3), (100, 151, 3)]):
        im = rng.integers(0, 256, shape, dtype=np.uint8)
        name = "%012d.png" % (k + 1)
        Image.fromarray(im).save(str(tmp_path / name))
        images.append({"id": 1000 + k, "file_name": name, "height": shape[0], "width": shape[1]})
        for r in ssd.coco_eval.detection_records(ora, im, 1000 + k, mapping, score_threshold=0.3):
            x, y, w, h = r["bbox"]
            if w > 0 and h > 0:
                anns.append({"id": len(anns) + 1, "image_id": r["image_id"], "category_id": r["category_id"], "bbox": r["bbox"],
                             "area": float(w * h), "iscrowd": 0})
    gt = {"images": images, "annotations": anns, "categories": cats}
    with open(tmp_path / "instances.json", "w") as f:
        json.dump(gt, f)
    assert len(anns) > 30
    st_gpu = ssd.coco_eval.evaluate(det, str(tmp_path / "instances.json"), str(tmp_path), predictions_json=str(tmp_path / "pred_gpu.json"), max_batch=4)
    st_ora = ssd.coco_eval.evaluate(ora, gt, str(tmp_path), predictions_json=str(tmp_path / "pred_ora.json"))
    assert np.array_equal(st_gpu, st_ora)
    assert json.load(open(tmp_path / "pred_gpu.json")) == json.load(open(tmp_path / "pred_ora.json"))
    # identical boxes of one category in one image (integer truncation) could only swap partners, never lose one
    assert st_gpu[0] == 1.0 and st_gpu[1] == 1.0 and st_gpu[8] == 1.0, st_gpu


def test_ssd_mirror_reference_signature_any_image_size(cuda, ssd, oracle_graph):
    """SSD(images, feature_extractor, anchor_generator, box_predictor, num_classes) (detector/ssd.py:10) on frames that the
    serving graph resizes and pads (100x151 -> 128x256): anchors for the NETWORK's size, predictions == the oracle's."""
    Wt = ssd.synthetic_weights(PARAMS, seed=3, logits_bias=-4.0)
    eng = ssd.Engine(PARAMS, Wt)
    for shape in [(2, 100, 151, 3), (1, 300, 128, 3), (1, 128, 128, 3)]:
        img = np.random.default_rng(shape[1]).integers(0, 256, shape, dtype=np.uint8)
        s = ssd.SSD(cuda.from_numpy(img).cuda(), ssd.RetinaNetFeatureExtractor(eng), ssd.AnchorGenerator(),
                    ssd.RetinaNetBoxPredictor(eng), 80)
        nh, nw, _bs = ssd.network_input_size(shape[1], shape[2], 128)
        assert s.anchors.shape[0] == sum(-(-nh // st) * -(-nw // st) * 6 for st in (8, 16, 32, 64, 128))
        assert np.array_equal(s.anchors.cpu().numpy(), oracle_graph.ops.anchors(nh, nw))
        pred = s.get_predictions(score_threshold=0.3, iou_threshold=0.5, max_boxes_per_class=10)
        ref = oracle_graph.forward(img, Wt, ssd.load_config(dict(PARAMS, score_threshold=0.3, iou_threshold=0.5, max_boxes_per_class=10)))
        assert np.array_equal(pred["num_boxes"].cpu().numpy(), ref["num_boxes"]) and ref["num_boxes"].min() > 0
        assert np.array_equal(pred["labels"].cpu().numpy(), ref["labels"])
        assert np.array_equal(pred["boxes"].cpu().numpy(), ref["boxes"]) and np.array_equal(pred["scores"].cpu().numpy(), ref["scores"])
    with pytest.raises(TypeError):
        ssd.SSD(cuda.from_numpy(img).cuda(), ssd.RetinaNetFeatureExtractor(eng))
    eng.close()

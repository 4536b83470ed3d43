"""CPU-side host logic: config surface, weight container, C ABI symbols, anchors."""
import ctypes
import os
import re

import numpy as np
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers.pb_writer import write_frozen_graph  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_config_surface(ssd):
    p = ssd.load_config(os.path.join(HERE, "golden", "config_mobilenet.json"))
    assert p == {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80,
                 "score_threshold": 0.15, "iou_threshold": 0.6, "max_boxes_per_class": 25,
                 "min_dimension": 640}
    assert ssd.load_config(os.path.join(HERE, "golden", "config_shufflenet.json"))["backbone"] == "shufflenet"
    with pytest.raises(KeyError):
        ssd.load_config({"backbone": "mobilenet"})
    with pytest.raises(ValueError):
        ssd.load_config(dict(p, backbone="resnet"))


def test_variable_catalogue(ssd):
    p = ssd.load_config(os.path.join(HERE, "golden", "config_mobilenet.json"))
    shapes = ssd.variable_shapes(p)
    assert shapes["MobilenetV1/Conv2d_0/weights"] == (3, 3, 3, 32)
    assert shapes["MobilenetV1/Conv2d_13_pointwise/weights"] == (1, 1, 1024, 1024)
    assert shapes["fpn/p6/kernel"] == (3, 3, 1024, 256)
    assert shapes["class_net/logits/kernel"] == (3, 3, 256, 480)
    assert shapes["box_net/batch_norm_3_for_level_7/moving_variance"] == (256,)
    conv = sum(int(np.prod(s)) for n, s in shapes.items() if n.endswith(("weights", "kernel")))
    assert abs(conv / 1e6 - 14.24) < 0.01                       # 14.24 M conv params
    ps = ssd.load_config(os.path.join(HERE, "golden", "config_shufflenet.json"))
    ss = ssd.variable_shapes(ps)
    assert ss["ShuffleNetV2/Stage2/unit_1/conv1x1_after/weights"] == (1, 1, 24, 58)
    assert ss["ShuffleNetV2/Stage3/unit_8/depthwise/depthwise_weights"] == (3, 3, 116, 1)
    assert ss["ShuffleNetV2/Conv5/weights"] == (1, 1, 464, 1024)
    assert ss["fpn/lateral3/kernel"] == (1, 1, 116, 256)
    conv = sum(int(np.prod(s)) for n, s in ss.items() if n.endswith(("weights", "kernel")))
    assert abs(conv / 1e6 - 12.18) < 0.01                       # 12.18 M


def test_weights_roundtrip(ssd, tmp_path):
    p = {"backbone": "mobilenet", "depth_multiplier": 0.25, "num_classes": 3, "score_threshold": 0.1,
         "iou_threshold": 0.5, "max_boxes_per_class": 5, "min_dimension": 128}
    W = ssd.synthetic_weights(p, seed=1)
    W2 = ssd.synthetic_weights(p, seed=1)
    assert all(np.array_equal(W[k], W2[k]) for k in W)
    ssd.save_weights(str(tmp_path / "w.npz"), W)
    L = ssd.load_weights(str(tmp_path / "w.npz"))
    assert set(L) == set(W) and all(np.array_equal(W[k], L[k]) for k in W)
    assert abs(float(W["class_net/logits/bias"][0]) + np.log(99.0)) < 1e-6   # box_predictor.py:121-127


def test_frozen_graph_reader(ssd, tmp_path):
    """`.pb` weights without TensorFlow: GraphDef wire format round trip (Const nodes with
    tensor_content or packed float_val, non-Const nodes skipped, 'import/' prefix stripped)."""
    p = {"backbone": "shufflenet", "depth_multiplier": 0.5, "num_classes": 2, "score_threshold": 0.1,
         "iou_threshold": 0.5, "max_boxes_per_class": 5, "min_dimension": 128}
    W = ssd.synthetic_weights(p, seed=4)
    names = list(W)
    data = write_frozen_graph(W, str(tmp_path / "model.pb"), use_float_val=set(names[::7]))
    consts = ssd.read_frozen_graph(str(tmp_path / "model.pb"))
    assert set(W) <= set(consts) and "images" not in consts
    L = ssd.load_pb_weights(data, p)
    assert set(L) == set(W) and all(np.array_equal(W[k], L[k]) and L[k].dtype == np.float32 for k in W)
    pref = write_frozen_graph({"import/" + k: v for k, v in list(W.items())[:3]}, extra_nodes=False)
    assert set(ssd.read_frozen_graph(pref)) == set(names[:3])
    broken = dict(W)
    del broken[names[5]]
    with pytest.raises(KeyError):
        ssd.load_pb_weights(write_frozen_graph(broken), p)
    with pytest.raises(ValueError):
        ssd.read_frozen_graph(data[:len(data) // 2 + 3])


@pytest.mark.parametrize("unpacked", [False, True])
@pytest.mark.parametrize("prefix", ["", "import/"])
def test_frozen_graph_reader_against_the_official_protobuf_encoder(ssd, unpacked, prefix):
    """pb_import.py (a hand-written wire-format reader) against google.protobuf's own encoder over TF's GraphDef message
    family (tests/helpers/tf_protos.py), in the shape create_pb.py:57-85 leaves a frozen graph: Const + `/read` Identity
    pairs (optionally under `import/`, inference/detector.py:13-19), a uint8 Placeholder whose shape has unknown (-1)
    dimensions, Conv2D / FusedBatchNorm consumers carrying list / string / bool / float attributes, int32 / int64 / string
    Consts that are not weights, constant-filled batch-norm vectors in TF's one-value form, tensors as float_val lists
    (packed, and unpacked with `unpacked`), everything else as tensor_content."""
    from helpers.tf_protos import frozen_graph
    p = {"backbone": "shufflenet", "depth_multiplier": 0.5, "num_classes": 2, "score_threshold": 0.1,
         "iou_threshold": 0.5, "max_boxes_per_class": 5, "min_dimension": 128}
    W = ssd.synthetic_weights(p, seed=4)
    names = list(W)
    splat = {n for n in names if n.endswith("moving_variance")}
    for n in splat:
        W[n] = np.full_like(W[n], 1.0)                    # e.g. an untrained moving variance: TF stores ONE float_val
    as_vals = set(names[::9]) - splat
    data = frozen_graph(W, prefix=prefix, unpacked=unpacked, splat=splat, as_vals=as_vals)
    consts = ssd.read_frozen_graph(data)
    assert set(W) <= set(consts) and "images" not in consts and "Assert/data_0" not in consts
    assert np.array_equal(consts["resize/size"], np.array([640, 896], np.int32))
    L = ssd.load_pb_weights(data, p)
    assert set(L) == set(W) and all(np.array_equal(W[k], L[k]) and L[k].dtype == np.float32 and L[k].shape == W[k].shape for k in W)
    # the official encoder and this repository's test writer agree on what a plain graph is
    from helpers.pb_writer import write_frozen_graph as own
    small = {k: W[k] for k in names[:4]}
    assert ssd.read_frozen_graph(own(small)).keys() >= small.keys()
    with pytest.raises(ValueError):
        ssd.read_frozen_graph(data[:len(data) // 3 + 1])


def test_coco_records(ssd):
    """evaluate_on_COCO.ipynb cell 10 record construction (no GPU: a stub detector)."""
    def det(image, score_threshold=0.15):
        return (np.array([[0.1, 0.2, 0.5, 0.9], [0.0, 0.0, 1.0, 1.0]], np.float32), np.array([0, 79], np.int32),
                np.array([0.9, 0.2], np.float32))
    cats = [{"name": n, "id": i + 1 if i < 11 else i + 2} for i, n in enumerate(ssd.coco_eval.COCO_NAMES)]
    m = ssd.coco_eval.integer_to_coco_id(cats)
    assert m[0] == 1 and m[79] == 81 and len(m) == 80
    recs = ssd.coco_eval.detection_records(det, np.zeros((480, 640, 3), np.uint8), 42, m)
    assert recs[0] == {"image_id": 42, "category_id": 1, "bbox": [128, 48, 448, 192], "score": float(np.float32(0.9))}
    assert recs[1]["bbox"] == [0, 0, 640, 480] and recs[1]["category_id"] == 81


def test_abi_exports_every_declared_symbol(ssd):
    """The C-ABI library loads and exports every function include/ssd_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "ssd_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ssd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 19
    lib = ctypes.CDLL(ssd.build())
    for name in declared:
        assert hasattr(lib, name), name
    from importlib import import_module
    sigs = import_module("ssd_amd._lib").SIGNATURES
    assert declared == set(sigs)
    # diagnostics (ablation kernels, tile overrides, timing entry points) live in libssd_hip_diag.so only
    dh = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ssd_hip_diag.h")).read(), flags=re.S)
    diag = set(re.findall(r"\b(ssd_[a-z0-9_]+)\s*\(", dh))
    assert diag == {"ssd_bench_conv", "ssd_bench_dwpw"} == set(import_module("ssd_amd._lib").DIAG_SIGNATURES)
    for name in diag:
        assert not hasattr(lib, name), name + " exported by the shipped library"
    # ... and NOTHING else: the header is the boundary (csrc/exports.map); C++ internals stay local to the library
    import subprocess
    for path in (ssd.lib_path(),):
        names = [l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", path], text=True).splitlines() if l.strip()]
        assert names and all(n.startswith("ssd_") for n in names), [n for n in names if not n.startswith("ssd_")][:5]
        assert set(names) == declared
    blob = open(ssd.lib_path(), "rb").read()
    for switch in (b"SSD_IGEMM16_DBG", b"SSD_BENCH_PRECISION", b"SSD_TS_DUMP"):
        assert switch not in blob, switch


def test_library_reads_one_environment_variable_and_options_go_through_the_abi(ssd):
    """The shipped library's configuration surface: SSD_PRECISION is the only environment variable it reads; every kernel /
    schedule selector is an ssd_set_option key (process-wide with a NULL handle, no GPU needed for that)."""
    csrc = os.path.join(ROOT, "single-shot-detector_amd", "csrc")
    n = 0
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            src = open(os.path.join(csrc, f)).read()
            src = re.sub(r"#ifdef SSD_DIAG.*?#e(?:lse|ndif)", "", src, flags=re.S)     # diagnostics build only
            n += len(re.findall(r"\bgetenv\s*\(", src))
    assert n == 1, n
    blob = open(ssd.lib_path(), "rb").read()
    for gone in (b"SSD_IGEMM_TILE", b"SSD_IGEMM16", b"SSD_NSUB", b"SSD_FUSE_DW", b"SSD_GRAPH", b"SSD_NMS_FAST_MAX", b"SSD_DEBUG_SYNC",
                 b"SSD_LEVEL_SPLIT", b"SSD_BACKBONE_SPLIT", b"SSD_LATERAL_SPLIT", b"SSD_IGEMM_96"):
        assert gone not in blob, gone
    unset = -2 ** 31
    for key in ("streams", "h2d_chunks", "front_fuse", "fuse_dw", "backbone_split", "event_fence", "plan_cache_mb", "igemm_tile", "igemm16", "igemm_96",
                "igemm_lat", "igemm_deep64", "lateral_split", "fpn_group", "fpn_p7_group", "fpn_early_lat", "nsub", "nms_fast_max", "debug_sync"):
        v = 64 if key == "igemm_tile" else 3
        assert ssd.get_option(key) == unset
        ssd.set_option(key, v)
        assert ssd.get_option(key) == v
        ssd.set_option(key, unset)
        assert ssd.get_option(key) == unset
    # values the build does not implement are refused, not ignored (igemm_tile 28, 29, 31, 32: diagnostics build only)
    for bad in (3, 28, 29, 31, 32, 96, 256):
        with pytest.raises(ssd.SsdError, match="igemm_tile"):
            ssd.set_option("igemm_tile", bad)
    for good in (128, 64, 20, 27, 30, 0, unset):
        ssd.set_option("igemm_tile", good)
    with pytest.raises(ssd.SsdError, match="plan_cache_mb"):
        ssd.set_option("plan_cache_mb", -5)
    with pytest.raises(ssd.SsdError, match="unknown option"):
        ssd.set_option("no_such_switch", 1)
    # the schedule experiments of rounds 1-4 are out of the shipped library (scripts/experiments/README.md)
    for gone in ("tower_group", "head_serial", "side_priority", "level_split", "fpn_p6_first", "lat_one", "dwpw_lat", "graph"):
        with pytest.raises(ssd.SsdError, match="unknown option"):
            ssd.set_option(gone, 1)


def test_anchors_host_side(ssd, oracle_ops):
    """ssd_anchors is host arithmetic (no GPU needed): identical to the oracle's table."""
    for H, W in [(640, 896), (640, 640), (128, 128), (256, 384)]:
        a = ssd.AnchorGenerator()(H, W)
        assert np.array_equal(a, oracle_ops.anchors(H, W))
    # any hyper-parameters (anchor_generator.py:13-38): ssd_anchors_ex against the oracle's twin
    other = dict(strides=[16, 32, 64], scales=[40, 96.5, 200], scale_multipliers=[1.0, 1.26, 1.5874],
                 aspect_ratios=[1.0, 3.0, 1.0 / 3.0, 0.5])
    for H, W in [(256, 384), (250, 330), (640, 896)]:
        g = ssd.AnchorGenerator(**other)
        a = g(H, W)
        assert g.num_anchors_per_location == 12 and sum(g.num_anchors_per_feature_map) == len(a)
        assert np.array_equal(a, oracle_ops.anchors_ex(H, W, other["strides"], other["scales"], other["scale_multipliers"],
                                                       other["aspect_ratios"]))
    with pytest.raises(AssertionError):
        ssd.AnchorGenerator(strides=[8, 16], scales=[32])
    with pytest.raises(ssd.SsdError):
        ssd.AnchorGenerator(strides=[0], scales=[32])(128, 128)


def test_no_cpu_fallback(ssd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = ssd.load_config(os.path.join(HERE, "golden", "config_mobilenet.json"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ssd.Engine(p, {})
    with pytest.raises(RuntimeError):
        ssd.ssd.conv2d(torch.zeros(1, 4, 4, 8), np.zeros((1, 1, 8, 8), np.float32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "single-shot-detector_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libssd_oracle" not in src, f


def test_voc_ap_self_check_known_answers(ssd):
    """coco_eval.average_precision / Evaluator restate metrics.py:156-282; answers worked out by hand from those rules."""
    ce = ssd.coco_eval
    gt = {"a": np.array([[0.0, 0.0, 1.0, 1.0], [2.0, 2.0, 3.0, 3.0]]), "b": np.array([[0.0, 0.0, 2.0, 2.0]])}
    dets = [("a", [0.0, 0.0, 1.0, 1.0], 0.9),        # TP (IoU 1)
            ("a", [0.0, 0.0, 1.0, 0.9], 0.8),        # same box again: already matched -> FP
            ("b", [0.0, 0.0, 2.0, 1.0], 0.7),        # IoU exactly 0.5 -> TP (>=)
            ("a", [2.0, 2.0, 3.0, 2.4], 0.6),        # IoU 0.4 -> FP
            ("c", [0.0, 0.0, 1.0, 1.0], 0.5)]        # image without groundtruth -> FP
    m = ce.average_precision(gt, dets, 0.5)
    # precision by rank: 1, 1/2, 2/3, 2/4, 2/5; recall: 1/3, 1/3, 2/3, 2/3, 2/3; AP = 1 * 1/3 + 2/3 * 1/3
    assert abs(m["AP"] - (1 / 3 + 2 / 9)) < 1e-12
    assert m["total_FP"] == 3 and m["total_FN"] == 1 and abs(m["mean_iou_for_TP"] - 0.75) < 1e-12
    # P*R*(1-|P-R|): rank 0: 1/3 * (1 - 2/3) = 1/9; rank 2: 4/9 * 1 = 4/9 -> best threshold 0.7
    assert m["best_threshold"] == 0.7 and abs(m["precision"] - 2 / 3) < 1e-12 and abs(m["recall"] - 2 / 3) < 1e-12
    # ties in confidence keep insertion order (list.sort is stable); ties in IoU pick the first groundtruth box
    gt2 = {"a": np.array([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 1.0]])}
    m2 = ce.average_precision(gt2, [("a", [0, 0, 1, 1], 0.5), ("a", [0, 0, 1, 1], 0.5)], 0.5)
    assert m2["AP"] == 0.5 and m2["total_FP"] == 1 and m2["total_FN"] == 1      # both want box 0; box 1 is never matched
    # no detections, no groundtruth
    assert ce.average_precision({}, [], 0.5)["AP"] == 0.0
    ev = ce.Evaluator(3)
    ev.add_image([[0, 0, 1, 1], [0, 0, 2, 2]], [0, 2], [[0, 0, 1, 1], [0, 0, 2, 2], [5, 5, 6, 6]], [0, 2, 1], [0.9, 0.8, 0.3])
    ev.add_image([[1, 1, 2, 2]], [0], [[1, 1, 2, 2]], [0], [0.7])
    out = ev.evaluate()
    assert out[0]["AP"] == 1.0 and out[2]["AP"] == 1.0 and out[1]["AP"] == 0.0 and abs(out["mAP"] - 2 / 3) < 1e-12


def test_numa_binding_helper_is_harmless_without_a_gpu():
    """bind_to_gpu_numa_node: one process per GPU on the GPU's NUMA node (bench.py calls it).  Where the device or the
    topology cannot be read it must change nothing and say so."""
    import os
    import ssd_amd
    before = os.sched_getaffinity(0)
    node = ssd_amd.bind_to_gpu_numa_node(0)
    after = os.sched_getaffinity(0)
    if node is None:
        assert after == before
    else:
        assert isinstance(node, int) and node >= 0 and after and after <= before
        os.sched_setaffinity(0, before)

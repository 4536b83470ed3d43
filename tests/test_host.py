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


def _train_like_checkpoint(ssd, W, rng):
    """What a train.py run's Saver holds next to the model's variables: global_step (int64), Adam's slots and power
    accumulators (model.py:115-118), the moving averages of the trainable variables (model.py:124-127)."""
    extra = {"global_step": np.array(150000, np.int64), "optimizer/beta1_power": np.array(0.0, np.float32),
             "optimizer/beta2_power": np.array(0.5, np.float32)}
    ema = {}
    for i, (k, v) in enumerate(W.items()):
        leaf = k.rsplit("/", 1)[1]
        if leaf not in ("moving_mean", "moving_variance"):
            if v.size <= 4096 or i % 16 == 0:                # (slots of every small variable and of a sample of the large ones)
                extra["optimizer/" + k + "/Adam"] = np.zeros_like(v)
                extra["optimizer/" + k + "/Adam_1"] = np.zeros_like(v)
            ema[k] = (v + np.float32(1e-3)).astype(np.float32)
            extra[k + "/ExponentialMovingAverage"] = ema[k]
    return extra, ema


def _hybrid_crc(ssd):
    """The writer's own byte loop for tensors below 32 KB (hundreds of batch-norm vectors and small kernels), the product's
    many-lane CRC -- pinned against that byte loop in test_crc32c_known_answers -- for the large ones (5 MB/s would cost minutes)."""
    from helpers.tf_bundle_writer import crc32c_bytewise
    return lambda raw: crc32c_bytewise(raw) if len(raw) < 32768 else ssd.crc32c(raw)


def test_crc32c_known_answers(ssd):
    """CRC-32C (Castagnoli) check values: "123456789" and the iSCSI vectors of RFC 3720 B.4; the many-lane numpy path
    against the byte loop on ragged lengths; continuation from a previous value."""
    assert ssd.crc32c(b"123456789") == 0xE3069283 and ssd.crc32c(b"") == 0
    assert ssd.crc32c(bytes(32)) == 0x8A9136AA and ssd.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert ssd.crc32c(bytes(range(32))) == 0x46DD794E and ssd.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    from helpers.tf_bundle_writer import crc32c_bytewise
    rng = np.random.default_rng(0)
    for n in (1, 16383, 16384, 16385, 70001, 262144 + 5):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert ssd.crc32c(d) == crc32c_bytewise(d), n
        assert ssd.crc32c(d[n // 3:], ssd.crc32c(d[:n // 3])) == crc32c_bytewise(d)
    a = rng.normal(size=(3, 3, 64, 96)).astype(np.float32)
    assert ssd.crc32c(a) == crc32c_bytewise(a.tobytes())


@pytest.mark.parametrize("num_shards,block_size,restart_interval", [(1, 4096, 16), (3, 300, 2), (2, 1 << 20, 16)])
def test_checkpoint_reader(ssd, tmp_path, num_shards, block_size, restart_interval):
    """A V2 checkpoint (tensor bundle) read without TensorFlow: the .index table in one block and in many (prefix-compressed
    keys across restart points), one and several data shards, the Saver's other tensors stepped over, raw variables
    (create_pb.py's choice) and moving averages (train.py's evaluation, model.py:148-161), the three places a user has
    one: a prefix, the model_dir's `checkpoint` state file, a SavedModel directory."""
    from helpers.tf_bundle_writer import write_bundle
    p = {"backbone": "shufflenet", "depth_multiplier": 0.5, "num_classes": 2, "score_threshold": 0.1,
         "iou_threshold": 0.5, "max_boxes_per_class": 5, "min_dimension": 128}
    rng = np.random.default_rng(7)
    W = ssd.synthetic_weights(p, seed=4)
    extra, ema = _train_like_checkpoint(ssd, W, rng)
    mdir = tmp_path / "run00"
    mdir.mkdir()
    prefix = str(mdir / "model.ckpt-150000")
    nblocks = write_bundle(prefix, {**W, **extra}, num_shards=num_shards, block_size=block_size, restart_interval=restart_interval,
                           strings=("_CHECKPOINTABLE_OBJECT_GRAPH",), sliced=("optimizer/beta2_power",), fast_crc=_hybrid_crc(ssd))
    assert (nblocks > 20) == (block_size == 300)
    header, entries = ssd.read_checkpoint_index(prefix)
    assert header["num_shards"] == num_shards and set(entries) == set(W) | set(extra) | {"_CHECKPOINTABLE_OBJECT_GRAPH"}
    e = entries["fpn/p6/kernel"]
    assert e["dtype"] == 1 and e["shape"] == W["fpn/p6/kernel"].shape and e["size"] == W["fpn/p6/kernel"].nbytes
    allv = ssd.read_checkpoint(prefix)
    assert "_CHECKPOINTABLE_OBJECT_GRAPH" not in allv and "optimizer/beta2_power" not in allv
    assert allv["global_step"].dtype == np.int64 and allv["global_step"].shape == () and int(allv["global_step"]) == 150000
    L = ssd.load_ckpt_weights(prefix, p)
    assert set(L) == set(W) and all(np.array_equal(W[k], L[k]) and L[k].dtype == np.float32 and L[k].flags.c_contiguous for k in W)
    E = ssd.load_ckpt_weights(prefix, p, use_ema=True)
    assert all(np.array_equal(E[k], ema.get(k, W[k])) for k in W) and any(not np.array_equal(E[k], W[k]) for k in ema)
    # the places a checkpoint is found
    (mdir / "checkpoint").write_text('model_checkpoint_path: "model.ckpt-150000"\nall_model_checkpoint_paths: "model.ckpt-100\n')
    for where in (str(mdir), prefix + ".index", "%s.data-00000-of-%05d" % (prefix, num_shards)):
        assert ssd.resolve_checkpoint(where) == prefix
    assert ssd.resolve_checkpoint(str(tmp_path)) is None and ssd.resolve_checkpoint(str(tmp_path / "nothing")) is None
    sm = tmp_path / "export" / "1546300800" / "variables"
    sm.mkdir(parents=True)
    write_bundle(str(sm / "variables"), W, block_size=block_size, fast_crc=_hybrid_crc(ssd))
    for where in (str(tmp_path / "export"), str(tmp_path / "export" / "1546300800")):
        assert ssd.resolve_checkpoint(where) == str(sm / "variables")
        S = ssd.load_ckpt_weights(where, p)
        assert all(np.array_equal(W[k], S[k]) for k in W)
    with pytest.raises(KeyError):
        ssd.read_checkpoint(prefix, ["no/such/variable"])
    with pytest.raises(ValueError):
        ssd.read_checkpoint(prefix, ["optimizer/beta2_power"])          # a partitioned variable's entry
    missing = dict(W)
    del missing["fpn/lateral4/kernel"]
    write_bundle(str(tmp_path / "partial"), missing, fast_crc=_hybrid_crc(ssd))
    with pytest.raises(KeyError):
        ssd.load_ckpt_weights(str(tmp_path / "partial"), p)
    with pytest.raises(FileNotFoundError):
        ssd.load_ckpt_weights(str(tmp_path / "nothing"), p)


def test_convert_any_container_to_npz(ssd, tmp_path):
    """`python -m ssd_amd.convert` -- the place of create_pb.py: .pb / checkpoint / model_dir / .npz -> the .npz the Detector
    loads, checked against the architecture the config names."""
    import importlib
    import json
    import subprocess
    from helpers.tf_bundle_writer import write_bundle
    convert = importlib.import_module("ssd_amd.convert")
    p = {"backbone": "mobilenet", "depth_multiplier": 0.25, "num_classes": 3, "score_threshold": 0.1,
         "iou_threshold": 0.5, "max_boxes_per_class": 5, "min_dimension": 128}
    W = ssd.synthetic_weights(p, seed=2)
    json.dump(p, open(tmp_path / "config.json", "w"))
    ema = {k + "/ExponentialMovingAverage": v + np.float32(0.5) for k, v in W.items() if k.endswith("gamma")}
    write_bundle(str(tmp_path / "model.ckpt-9"), {**W, **ema}, fast_crc=_hybrid_crc(ssd))
    write_frozen_graph(W, str(tmp_path / "model.pb"))
    for src in ("model.ckpt-9", "model.pb"):
        assert convert.main([str(tmp_path / src), str(tmp_path / "config.json"), str(tmp_path / "out.npz")]) == 0
        L = ssd.load_weights(str(tmp_path / "out.npz"))
        assert set(L) == set(W) and all(np.array_equal(L[k], W[k]) for k in W)
    r = subprocess.run([sys.executable, "-m", "ssd_amd.convert", str(tmp_path / "model.ckpt-9"), str(tmp_path / "config.json"),
                        str(tmp_path / "ema.npz"), "--ema"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "variables" in r.stdout, r.stderr
    E = ssd.load_weights(str(tmp_path / "ema.npz"))
    assert all(np.array_equal(E[k], ema.get(k + "/ExponentialMovingAverage", W[k])) for k in W)
    assert convert.main([str(tmp_path / "ema.npz"), str(tmp_path / "config.json"), str(tmp_path / "again.npz")]) == 0
    assert convert.main([]) == 2
    with pytest.raises(ValueError):
        convert.load_any(str(tmp_path / "out.npz"), dict(p, num_classes=4))
    with pytest.raises(FileNotFoundError):
        convert.load_any(str(tmp_path / "nothing"), p)


def test_checkpoint_index_table_round_trips_any_keys(ssd, tmp_path):
    """The .index file's table format under random contents (hypothesis): keys with long shared prefixes, empty values, values
    larger than a block, any block size and restart interval -- what the independent writer wrote is what the reader returns, in
    key order."""
    from hypothesis import given, settings, strategies as st
    from helpers.tf_bundle_writer import write_table
    ck = __import__("importlib").import_module("ssd_amd.ckpt_import")
    path = str(tmp_path / "t.index")
    stem = st.sampled_from([b"", b"MobilenetV1/Conv2d_", b"fpn/p", b"class_net/batch_norm_", b"\xff\x00"])
    key = st.builds(lambda a, b: a + b, stem, st.binary(min_size=0, max_size=12))

    @settings(max_examples=60, deadline=None)
    @given(st.dictionaries(key, st.binary(min_size=0, max_size=300), min_size=0, max_size=60),
           st.integers(min_value=16, max_value=5000), st.integers(min_value=1, max_value=20))
    def run(items, block_size, restart_interval):
        ordered = sorted(items.items())
        write_table(path, ordered, block_size=block_size, restart_interval=restart_interval)
        got = ck.read_table(path)
        assert list(got.items()) == ordered
    run()


def test_checkpoint_reader_detects_damage(ssd, tmp_path):
    """What TensorFlow's BundleReader reports as DataLoss: a flipped bit in an index block, in a tensor's bytes, a truncated
    index, a missing shard, a file that is no table at all."""
    from helpers.tf_bundle_writer import write_bundle, write_table
    rng = np.random.default_rng(3)
    T = {"a/weights": rng.normal(size=(3, 3, 8, 16)).astype(np.float32), "a/BatchNorm/beta": rng.normal(size=(16,)).astype(np.float32),
         "b/kernel": rng.normal(size=(70000,)).astype(np.float32)}
    prefix = str(tmp_path / "m")
    write_bundle(prefix, T, block_size=64)
    good = ssd.read_checkpoint(prefix)
    assert all(np.array_equal(good[k], T[k]) for k in T)
    idx = open(prefix + ".index", "rb").read()
    bad = bytearray(idx)
    bad[20] ^= 0x04
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="checksum"):
        ssd.read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(idx[:len(idx) - 9])
    with pytest.raises(ValueError):
        ssd.read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(idx)
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[len(data) // 2] ^= 0x10                                     # inside b/kernel (the lanes path of the CRC)
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="b/kernel"):
        ssd.read_checkpoint(prefix)
    assert np.array_equal(ssd.read_checkpoint(prefix, ["a/weights"])["a/weights"], T["a/weights"])
    assert not np.array_equal(ssd.read_checkpoint(prefix, verify=False)["b/kernel"], T["b/kernel"])
    os.remove(prefix + ".data-00000-of-00001")
    with pytest.raises(FileNotFoundError):
        ssd.read_checkpoint(prefix)
    open(str(tmp_path / "v1.index"), "wb").write(b"\x00" * 100)
    with pytest.raises(ValueError, match="not a TensorFlow checkpoint index"):
        ssd.read_checkpoint(str(tmp_path / "v1"))


def test_abi_refuses_null_arguments_without_crashing(ssd):
    """Nothing throws or faults across the C ABI (include/ssd_hip.h: return codes + ssd_last_error): every entry point that takes
    a handle, called with a null handle and null buffers -- no GPU needed, the checks come before any HIP call."""
    L = ssd.lib()
    f32 = ctypes.c_float
    calls = {
        "ssd_create": (None, None), "ssd_set_precision": (None, 0), "ssd_get_precision": (None,), "ssd_get_option": (None, b"streams", None),
        "ssd_status": (None, None), "ssd_load_weight": (None, b"x", None, None, 0), "ssd_finalize": (None,),
        "ssd_forward": (None, None, 1, 128, 128, None, None, None, None, None), "ssd_forward_records": (None, None, 1, 128, 128, None, None),
        "ssd_forward_host": (None, None, 1, 128, 128, None, None), "ssd_network_shape": (None, 100, 100, None),
        "ssd_forward_mixed": (None, None, 1, None, None, None, None), "ssd_forward_mixed_host": (None, None, 1, None, None, None),
        "ssd_detect_host": (None, None, 128, 128, f32(0.1), None, None, None, None, 0, None, None),
        "ssd_get_tensor": (None, b"p3", None, 0, None), "ssd_get_tensor_dev": (None, b"p3", None, 0, None, None),
        "ssd_plan_cache_stats": (None, None), "ssd_plan_cache_clear": (None,), "ssd_profile_enable": (None, 1),
        "ssd_profile_read": (None, 0, None, None, None, None), "ssd_profile_reset": (None,), "ssd_anchors": (128, 128, None)}
    for name, args in calls.items():
        rc = getattr(L, name)(*args)
        assert rc < 0, (name, rc)
        assert name == "ssd_get_precision" or len(L.ssd_last_error()) > 0, name
    L.ssd_destroy(None)                                   # a no-op, like free(NULL)
    assert L.ssd_record_words(None) == 0 and L.ssd_num_anchors(-5, 128) == 0


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


def _coco_gt(boxes_by_image_cat, crowd=(), areas=None):
    """{(image, category): [xywh, ...]} -> an annotation-file dict (ids from 1, area = w * h unless given)."""
    imgs = sorted({k[0] for k in boxes_by_image_cat})
    cats = sorted({k[1] for k in boxes_by_image_cat})
    anns = []
    for (img, cat), boxes in sorted(boxes_by_image_cat.items()):
        for j, b in enumerate(boxes):
            aid = len(anns) + 1
            anns.append({"id": aid, "image_id": img, "category_id": cat, "bbox": list(b), "iscrowd": int((img, cat, j) in crowd),
                         "area": float(b[2] * b[3]) if areas is None else float(areas[(img, cat, j)])})
    return {"images": [{"id": i} for i in imgs], "categories": [{"id": c, "name": str(c)} for c in cats], "annotations": anns}


def test_coco_box_metric_known_answers(ssd):
    """coco_metric.py (cocoapi's COCOeval for boxes, restated: evaluate_on_COCO.ipynb cell 17) on cases worked by hand."""
    cm = ssd.coco_metric
    assert np.allclose(cm.IOU_THRS, [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]) and len(cm.REC_THRS) == 101
    # IoU: xywh, no overlap -> 0, touching edges -> 0, crowd -> intersection / detection area
    iou = cm.bbox_iou([[0, 0, 10, 10], [20, 20, 5, 5], [10, 0, 10, 10]], [[0, 0, 10, 5], [0, 0, 100, 100]], [0, 1])
    assert np.allclose(iou, [[0.5, 1.0], [0.0, 1.0], [0.0, 1.0]])
    # (1) two objects; detections: IoU 0.81 with the first (score .9), a false positive (.8), IoU 0.62 with the second (.7).
    #     t <= 0.60: TP FP TP -> precision envelope 1, 2/3, 2/3 at recall .5, .5, 1 -> (51 + 50 * 2/3) / 101;
    #     0.65 .. 0.80: TP FP FP -> 51 / 101;  0.85 ..: nothing matches -> 0
    gt = _coco_gt({(1, 7): [[0, 0, 100, 100], [200, 0, 100, 100]]})
    dt = [{"image_id": 1, "category_id": 7, "bbox": [0, 0, 100, 81], "score": 0.9},
          {"image_id": 1, "category_id": 7, "bbox": [500, 500, 50, 50], "score": 0.8},
          {"image_id": 1, "category_id": 7, "bbox": [200, 0, 100, 62], "score": 0.7}]
    st = cm.evaluate_boxes(gt, dt)
    lo, mid = (51 + 50 * 2 / 3) / 101, 51 / 101
    assert np.isclose(st[0], (3 * lo + 4 * mid) / 10) and np.isclose(st[1], lo) and np.isclose(st[2], mid)
    # both objects are "large" (area 10 000 > 96^2); in that range the stray 50 x 50 detection is out of range and unmatched, i.e.
    # ignored: t <= 0.60 -> TP TP -> 1;  0.65 .. 0.80 -> TP FP -> 51 / 101
    assert st[3] == -1 and st[4] == -1 and np.isclose(st[5], (3 * 1.0 + 4 * mid) / 10)
    assert np.isclose(st[6], (7 * 0.5) / 10) and np.isclose(st[8], (3 * 1.0 + 4 * 0.5) / 10)     # AR@1: only the best detection counts
    # the order of the result list does not matter, the scores do: with the false positive ranked first the precisions are
    # 0, 1/2, 2/3 -- the envelope from the right lifts them all to 2/3, at every recall point
    dt2 = [dict(dt[1], score=0.95), dt[2], dt[0]]
    st2 = cm.evaluate_boxes(gt, dt2)
    assert np.isclose(st2[1], 2 / 3) and st2[0] < st[0]
    # (2) perfect detections over images / categories: every defined statistic is 1, the empty area ranges are -1
    boxes = {(i, c): [[10 * c, 5 * i, 40, 50]] for i in (1, 2, 3) for c in (1, 2)}
    gt = _coco_gt(boxes)
    dt = [{"image_id": i, "category_id": c, "bbox": b[0], "score": 0.5 + 0.01 * i} for (i, c), b in boxes.items()]
    st = cm.evaluate_boxes(gt, dt)
    assert np.allclose(st[[0, 1, 2, 4, 6, 7, 8, 10]], 1.0) and np.all(st[[3, 5, 9, 11]] == -1)      # 40 x 50 = medium
    # a category without groundtruth does not enter the mean; detections of it on its own would only be false positives
    st3 = cm.evaluate_boxes(dict(gt, categories=gt["categories"] + [{"id": 9, "name": "9"}]), dt + [{"image_id": 1, "category_id": 9, "bbox": [0, 0, 5, 5], "score": 0.9}])
    assert np.allclose(st3[[0, 1, 2]], 1.0)
    # (3) a crowd region: detections inside it are neither true nor false positives, and it is not an object to find
    gt = _coco_gt({(1, 1): [[0, 0, 50, 50], [100, 100, 200, 200]]}, crowd={(1, 1, 1)})
    dt = [{"image_id": 1, "category_id": 1, "bbox": [120, 120, 30, 30], "score": 0.99},
          {"image_id": 1, "category_id": 1, "bbox": [150, 150, 30, 30], "score": 0.98},
          {"image_id": 1, "category_id": 1, "bbox": [0, 0, 50, 50], "score": 0.5},
          {"image_id": 1, "category_id": 1, "bbox": [400, 400, 30, 30], "score": 0.4}]
    st = cm.evaluate_boxes(gt, dt)
    assert np.isclose(st[0], 1.0) and np.isclose(st[8], 1.0)
    # (4) maxDets: three objects of one category in one image, three perfect detections: AR@1 = 1/3
    gt = _coco_gt({(1, 1): [[0, 0, 40, 40], [100, 0, 40, 40], [200, 0, 40, 40]]})
    dt = [{"image_id": 1, "category_id": 1, "bbox": a["bbox"], "score": sc} for a, sc in zip(gt["annotations"], (0.9, 0.8, 0.7))]
    st = cm.evaluate_boxes(gt, dt)
    assert np.isclose(st[6], 1 / 3) and np.isclose(st[7], 1.0) and np.isclose(st[0], 1.0)
    # (5) area ranges: a small object (annotation area 900) and a large one; the detection of the large one is ignored in the
    #     "small" range (unmatched there and itself out of range), the small object's detection is ignored in "large"
    gt = _coco_gt({(1, 1): [[0, 0, 30, 30], [100, 100, 120, 120]]})
    dt = [{"image_id": 1, "category_id": 1, "bbox": [0, 0, 30, 30], "score": 0.6}, {"image_id": 1, "category_id": 1, "bbox": [100, 100, 120, 120], "score": 0.9}]
    st = cm.evaluate_boxes(gt, dt)
    assert np.isclose(st[3], 1.0) and st[4] == -1 and np.isclose(st[5], 1.0) and np.isclose(st[0], 1.0)
    # ... and a MISSED small object shows in APs / ARs only
    st = cm.evaluate_boxes(gt, dt[1:])
    assert np.isclose(st[3], 0.0) and np.isclose(st[9], 0.0) and np.isclose(st[5], 1.0) and np.isclose(st[8], 0.5)
    # (6) the ignore rules between area ranges.  Two objects whose boxes overlap: A (annotation area 40 000: large) and B (its box is
    #     large too, its segmentation area 5 000: medium); one detection with IoU 0.92 to A and 0.674 to B, itself large.
    #     all:    it takes A (the higher IoU) up to t = 0.90, B is never found: 9 x (51 / 101) / 10
    #     medium: A is out of range = ignore, sorted behind B.  t <= 0.65: B matches and holds against the ignore box -> a true
    #             positive although the detection's own area is out of range; above: only A matches -> the detection is ignored
    #             (no false positive), B stays unfound: 4 / 10
    #     large:  B is ignore; the detection takes A up to 0.90 -> 1; at 0.95 it matches nothing and counts as a false positive: 0.9
    gt = _coco_gt({(1, 1): [[0, 0, 200, 200], [0, 0, 200, 124]]}, areas={(1, 1, 0): 40000.0, (1, 1, 1): 5000.0})
    dt = [{"image_id": 1, "category_id": 1, "bbox": [0, 0, 200, 184], "score": 0.8}]
    assert np.allclose(cm.bbox_iou([dt[0]["bbox"]], [a["bbox"] for a in gt["annotations"]], [0, 0]), [[0.92, 124 / 184]])
    st = cm.evaluate_boxes(gt, dt)
    assert np.isclose(st[0], 9 * (51 / 101) / 10) and np.isclose(st[4], 0.4) and np.isclose(st[5], 0.9) and st[3] == -1
    assert np.isclose(st[10], 0.4) and np.isclose(st[11], 0.9) and np.isclose(st[8], 0.45)
    with pytest.raises(ValueError):
        cm.evaluate_boxes(gt, [{"image_id": 99, "category_id": 1, "bbox": [0, 0, 1, 1], "score": 1.0}])
    import io
    buf = io.StringIO()
    cm.evaluate_boxes(gt, dt, out=buf)
    assert buf.getvalue().count("\n") == 12 and "Average Precision  (AP) @[ IoU=0.50:0.95 | area=   all | maxDets=100 ] = 0.454" in buf.getvalue()


def test_coco_evaluate_harness_reads_files_and_scores(ssd, tmp_path):
    """coco_eval.evaluate (evaluate_on_COCO.ipynb cells 4-17) with a stand-in detector, no GPU: annotation file and images from
    disk, label -> category id through the names, pixel xywh records, predictions file, the COCO statistics."""
    import json
    from PIL import Image
    cats = [{"id": i + 1 + (i > 10), "name": n} for i, n in enumerate(ssd.coco_eval.COCO_NAMES)]
    images = []
    for k, (h, w) in enumerate([(100, 200), (80, 80)]):
        Image.fromarray(np.full((h, w, 3), 10 * k, np.uint8)).save(str(tmp_path / ("%d.png" % k)))
        images.append({"id": 5 + k, "file_name": "%d.png" % k, "height": h, "width": w})
    anns = [{"id": 1, "image_id": 5, "category_id": 1, "bbox": [20, 10, 100, 50], "area": 5000.0, "iscrowd": 0},      # person
            {"id": 2, "image_id": 6, "category_id": 18, "bbox": [8, 8, 40, 40], "area": 1600.0, "iscrowd": 0}]        # dog (label 16)
    gt = {"images": images, "annotations": anns, "categories": cats}
    seen = []

    def det(image, score_threshold=0.15):
        seen.append((image.shape, score_threshold, int(image[0, 0, 0])))
        if image.shape[0] == 100:        # normalised ymin, xmin, ymax, xmax: the person exactly, plus a stray dog
            return (np.array([[0.1, 0.1, 0.6, 0.6], [0.5, 0.5, 0.9, 0.9]], np.float32), np.array([0, 16], np.int32), np.array([0.9, 0.4], np.float32))
        return (np.array([[0.1, 0.1, 0.6, 0.6]], np.float32), np.array([16], np.int32), np.array([0.3], np.float32))
    st = ssd.coco_eval.evaluate(det, gt, str(tmp_path), predictions_json=str(tmp_path / "pred.json"))
    assert seen == [((100, 200, 3), 0.15, 0), ((80, 80, 3), 0.15, 10)]
    pred = json.load(open(tmp_path / "pred.json"))
    assert pred[0] == {"image_id": 5, "category_id": 1, "bbox": [20, 10, 100, 50], "score": float(np.float32(0.9))}
    assert pred[2]["category_id"] == 18 and pred[2]["bbox"] == [8, 8, 40, 40]
    # person: AP 1.  dog: the stray (0.4) outranks the hit (0.3): precision 1/2 at every recall.  Mean over the two categories.
    assert np.isclose(st[0], 0.75) and np.isclose(st[1], 0.75) and np.isclose(st[8], 1.0)


def _brute_force_ap(gt, dt, thr):
    """AP at one IoU threshold straight from the definition, written independently of coco_metric.py (no crowd, all areas,
    fewer than 100 detections per image and category): detections in descending score take the free groundtruth box of highest
    IoU >= thr in their image; AP = mean over the 101 recall points r of the best precision at any rank whose recall >= r;
    mean over the categories that have groundtruth."""
    def iou(a, b):
        w = min(a[0] + a[2], b[0] + b[2]) - max(a[0], b[0])
        h = min(a[1] + a[3], b[1] + b[3]) - max(a[1], b[1])
        if w <= 0 or h <= 0:
            return 0.0
        return w * h / (a[2] * a[3] + b[2] * b[3] - w * h)
    aps, recalls = [], []
    for cat in sorted({c["id"] for c in gt["categories"]}):
        g = [a for a in gt["annotations"] if a["category_id"] == cat]
        if not g:
            continue
        taken = set()
        hits = []
        for d in sorted((d for d in dt if d["category_id"] == cat), key=lambda d: -d["score"]):
            cands = [(iou(d["bbox"], a["bbox"]), a["id"]) for a in g if a["image_id"] == d["image_id"] and a["id"] not in taken]
            cands = [c for c in cands if c[0] >= thr]
            if cands:
                taken.add(max(cands)[1])
            hits.append(bool(cands))
        tp = np.cumsum(hits)
        prec = tp / (np.arange(len(hits)) + 1.0) if hits else np.zeros(0)
        rec = tp / len(g) if hits else np.zeros(0)
        aps.append(np.mean([max([p for p, r in zip(prec, rec) if r >= x], default=0.0) for x in np.linspace(0, 1, 101)]))
        recalls.append(rec[-1] if hits else 0.0)
    return float(np.mean(aps)), float(np.mean(recalls))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_coco_box_metric_against_a_brute_force_ap(ssd, seed):
    """Random scenes (jittered copies of the objects as detections, duplicates, strays, missed objects, several categories and
    images): AP50, AP75, AP and AR@100 of coco_metric.py equal the definition evaluated by brute force."""
    cm = ssd.coco_metric
    rng = np.random.default_rng(seed)
    boxes, dt = {}, []
    for img in range(1, 7):
        for cat in (1, 2, 3):
            n = int(rng.integers(0, 6))
            bs = [[float(rng.uniform(0, 400)), float(rng.uniform(0, 400)), float(rng.uniform(20, 120)), float(rng.uniform(20, 120))] for _ in range(n)]
            if bs:
                boxes[(img, cat)] = bs
            for b in bs:
                for _ in range(int(rng.integers(0, 3))):             # 0 (missed), 1 or 2 (a duplicate) jittered detections
                    j = rng.normal(0, 6, 4)
                    dt.append({"image_id": img, "category_id": cat, "bbox": [b[0] + j[0], b[1] + j[1], max(b[2] + j[2], 4.0), max(b[3] + j[3], 4.0)],
                               "score": float(rng.uniform(0.2, 1.0))})
            for _ in range(int(rng.integers(0, 3))):                 # strays
                dt.append({"image_id": img, "category_id": cat, "bbox": [float(rng.uniform(0, 450)), float(rng.uniform(0, 450)), 40.0, 40.0],
                           "score": float(rng.uniform(0.1, 0.6))})
    boxes.setdefault((7, 1), [[5.0, 5.0, 50.0, 50.0]])               # an image nobody detected anything in
    gt = _coco_gt(boxes)
    ev = cm.CocoBoxEval(gt, dt).evaluate().accumulate()
    st = ev.summarize()
    per_t = [_brute_force_ap(gt, dt, t) for t in cm.IOU_THRS]
    assert np.isclose(st[1], per_t[0][0], atol=1e-12) and np.isclose(st[2], per_t[5][0], atol=1e-12)
    assert np.isclose(st[0], np.mean([p[0] for p in per_t]), atol=1e-12)
    assert np.isclose(st[8], np.mean([p[1] for p in per_t]), atol=1e-12)
    assert 0.05 < st[0] < 0.95


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
                "igemm_lat", "igemm_deep64", "lateral_split", "fpn_group", "fpn_p7_group", "fpn_early_lat", "nsub", "nms_fast_max", "first_conv_px",
                "debug_sync"):
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

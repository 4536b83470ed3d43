"""-m gpu: the whole drop-in path (ssd_forward behind Detector / Engine / SSD) against the
CPU oracle and against the committed golden fixture."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers.pb_writer import write_frozen_graph  # noqa: E402

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
TOL = 1e-4


def compare_outputs(got, ref, what):
    gb, gl, gs, gn = got
    assert np.array_equal(gn, ref["num_boxes"]), (what, gn, ref["num_boxes"])
    assert np.array_equal(gl, ref["labels"]), what + ": labels"
    assert np.abs(gs - ref["scores"]).max() <= TOL and np.abs(gb - ref["boxes"]).max() <= TOL, what
    print(what, "num", gn.tolist(), "scores bit-equal", float((gs == ref["scores"]).mean()),
          "boxes bit-equal", float((gb == ref["boxes"]).mean()))


def stage_check(engine, keep, names, what):
    worst = 1.0
    for n in names:
        got = engine.get_tensor(n)
        ref = keep[n]
        if n in ("encoded_boxes", "class_predictions"):
            ref = ref.reshape(got.shape)
        err = np.abs(got - ref).max()
        scale = max(1.0, float(np.abs(ref).max()))
        eq = float((got == ref).mean())
        worst = min(worst, eq)
        print("%s %s: max err %.3g scale %.3g bit-equal %.6f" % (what, n, err, scale, eq))
        assert err <= TOL * scale, (what, n)
    return worst


STAGES = ["c3", "c4", "c5", "p3", "p4", "p5", "p6", "p7", "encoded_boxes", "class_predictions"]


def test_golden_tiny(cuda, ssd):
    import make_golden as mg
    Wt, img = mg.inputs()
    z = np.load(os.path.join(HERE, "golden", "tiny_mobilenet_128.npz"))
    eng = ssd.Engine(mg.TINY, Wt)
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    compare_outputs(out, z, "golden tiny")
    assert np.abs(eng.get_tensor("encoded_boxes").reshape(2, -1, 4) - z["encoded_boxes"]).max() <= TOL
    cp = eng.get_tensor("class_predictions").reshape(2, -1, 80)[:, ::8]
    assert np.abs(cp - z["class_predictions_every8"]).max() <= TOL * 10
    for n in ("c5", "p5", "p6", "p7"):
        assert np.abs(eng.get_tensor(n) - z[n]).max() <= TOL * max(1.0, np.abs(z[n]).max())
    assert np.abs(eng.get_tensor("c3")[:, :4, :4] - z["c3_corner"]).max() <= TOL * 10
    assert np.abs(eng.get_tensor("p3")[:, :4, :4] - z["p3_corner"]).max() <= TOL * 10
    eng.close()


@pytest.mark.parametrize("backbone,H,W,B", [("mobilenet", 128, 256, 9), ("shufflenet", 128, 128, 2), ("shufflenet", 128, 128, 5)])
def test_forward_vs_oracle_small(cuda, ssd, oracle_graph, backbone, H, W, B):
    # (ShuffleNet at 5 images: its backbone as two UNEVEN half-batch chains, 2 + 3 images, round 4)
    params = {"backbone": backbone, "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 128}
    Wt = ssd.synthetic_weights(params, seed=11, logits_bias=-4.0)
    img = np.random.default_rng(5).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    keep = {}
    ref = oracle_graph.forward(img, Wt, params, keep)
    eng = ssd.Engine(params, Wt)
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    worst = stage_check(eng, keep, STAGES, backbone)
    compare_outputs(out, ref, backbone + " small")
    assert ref["num_boxes"].min() > 0
    assert worst == 1.0, "stages within tolerance but not bit-identical to the oracle"
    # the device-side fetch (ssd_get_tensor_dev: one permute launch, no host copy) returns the same logical tensors -- for ShuffleNet
    # c3 / c4 out of the stage outputs' two-part rows
    for n in ("c3", "c4", "c5", "p3", "encoded_boxes"):
        host = eng.get_tensor(n)
        assert np.array_equal(eng.get_tensor_dev(n, host.shape).cpu().numpy(), host), n
    # a second call with another batch size re-plans the arena and stays correct
    out1 = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img[:1].copy()).cuda())]
    for a, b in zip(out, out1):
        assert np.array_equal(a[:1], b)
    if B >= 4:      # one backbone chain instead of two: the same bits
        eng.set_option("backbone_split", 1)
        out2 = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
        for a, b in zip(out, out2):
            assert np.array_equal(a, b)
    # the backbone's first layers as separate launches again (front.hip off): the same bits
    eng.set_option("front_fuse", 0)
    out3 = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    for a, b in zip(out, out3):
        assert np.array_equal(a, b)
    eng.close()


@pytest.mark.parametrize("backbone,dm,classes,H,W", [("mobilenet", 0.5, 20, 128, 256), ("mobilenet", 0.75, 3, 256, 128),
                                                     ("shufflenet", 0.5, 20, 128, 128), ("shufflenet", 1.5, 80, 128, 256),
                                                     ("shufflenet", 2.0, 20, 128, 128)])
def test_forward_other_widths_and_class_counts(cuda, ssd, oracle_graph, backbone, dm, classes, H, W):
    """depth_multiplier and num_classes are part of the config surface (model.py:22-30): narrow
    backbones exercise the channel padding (16 -> 32, 24 -> 32, 88 -> 96 ...), other class counts
    the head widths (6*C = 18 / 120 / 480 columns) and every tile variant behind them."""
    params = {"backbone": backbone, "depth_multiplier": dm, "num_classes": classes, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 128}
    Wt = ssd.synthetic_weights(params, seed=21, logits_bias=-4.0)
    img = np.random.default_rng(7).integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    keep = {}
    ref = oracle_graph.forward(img, Wt, params, keep)
    eng = ssd.Engine(params, Wt)
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    assert stage_check(eng, keep, STAGES, "%s x%.2f" % (backbone, dm)) == 1.0
    compare_outputs(out, ref, "%s x%.2f C=%d" % (backbone, dm, classes))
    assert out[0].shape == (2, classes * 25, 4)
    eng.close()


@pytest.mark.parametrize("backbone", ["mobilenet", "shufflenet"])
def test_sub_batch_plans(cuda, ssd, oracle_graph, libopt, backbone):
    """option nsub forces the consecutive sub-batch plans that a batch past 2 GiB of activations takes (uneven split
    5 = 2+2+1): same results, same retained tensors (ShuffleNet: every plan has its own stage allocations and two-part rows,
    and ssd_get_tensor assembles a stage from all of them)."""
    libopt(nsub=3)
    params = {"backbone": backbone, "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 128}
    Wt = ssd.synthetic_weights(params, seed=12, logits_bias=-4.0)
    img = np.random.default_rng(6).integers(0, 256, (5, 128, 128, 3), dtype=np.uint8)
    keep = {}
    ref = oracle_graph.forward(img, Wt, params, keep)
    eng = ssd.Engine(params, Wt)
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    assert stage_check(eng, keep, STAGES, "nsub3 " + backbone) == 1.0
    compare_outputs(out, ref, "sub-batch plans " + backbone)
    eng.close()


@pytest.mark.parametrize("H,W,B", [(256, 384, 1), (640, 896, 1), (256, 256, 2)])
def test_small_batch_plan_variants_are_bit_identical(cuda, ssd, oracle_graph, H, W, B):
    """The batch-1 / batch-2 plan (round 3) picks other kernels and another launch structure than the serving plan: the
    latency form of the implicit GEMM on v_mfma_f32_16x16x4_f32 for tiny launches (fpn p6 / p7, lateral5: igemm_lat.hip), fpn
    p3 + p4 + p5 as one grouped launch, 64x64 tiles with deep prefetch.  Every one of them is
    bit-identical to what it replaces by construction (the same k-ordered fmaf chain per output): switching each off, all of
    them off, or everything onto one stream changes no bit of any retained tensor or output -- and the default equals the
    oracle."""
    params = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": min(H, W)}
    Wt = ssd.synthetic_weights(params, seed=11, logits_bias=-6.0)
    img = np.random.default_rng(H + W + B).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    dimg = cuda.from_numpy(img).cuda()
    names = ["c3", "c4", "c5", "p3", "p4", "p5", "p6", "p7", "encoded_boxes", "class_predictions"]
    eng = ssd.Engine(params, Wt, precision="f32")

    def run():
        out = [t.cpu().numpy() for t in eng.forward(dimg)]
        return out, {n: eng.get_tensor(n) for n in names}
    base_out, base_t = run()
    assert int(base_out[3].sum()) > 10
    if H * W <= 256 * 384:            # (the oracle at full size runs in test_forward_vs_oracle_full_size)
        keep = {}
        ref = oracle_graph.forward(img, Wt, params, keep)
        compare_outputs(base_out, ref, "small-batch plan %dx%d B=%d" % (H, W, B))
        assert all(np.array_equal(a, ref[k]) for a, k in zip(base_out, ("boxes", "labels", "scores", "num_boxes")))
        for n in names:
            assert np.array_equal(base_t[n], keep[n].reshape(base_t[n].shape)), n
    variants = [{"igemm_lat": 0}, {"fpn_group": 0}, {"igemm_deep64": 0}, {"streams": 1}, {"igemm_tile": 20}, {"igemm_tile": 25},
                {"igemm_lat": 0, "fpn_group": 0, "igemm_deep64": 0},
                # round 4: p7 out of the grouped launch, the laterals in their chain, fenced events
                {"fpn_p7_group": 0}, {"fpn_early_lat": 0}, {"fpn_early_lat": 0, "fpn_p7_group": 0}, {"event_fence": 1},
                # the first convolution and Conv2d_1 as two launches again (front.hip off)
                {"front_fuse": 0}, {"front_fuse": 0, "fuse_dw": 0}]
    for v in variants:
        for k, val in v.items():
            eng.set_option(k, val)
        out, t = run()
        for a, b in zip(out, base_out):
            assert np.array_equal(a, b), v
        for n in names:
            assert np.array_equal(t[n], base_t[n]), (v, n)
        for k in v:
            eng.set_option(k, -2 ** 31)           # back to "unset"
    eng.close()


def test_fpn_stride2_convs_on_the_block_form_are_bit_identical(cuda, ssd):
    """fpn p6 / p7 (3x3 stride 2) and, at small serving batches, p4 / p5 run on the four-wave block form of igemm_lat.hip (plan.hip;
    8 frames of 640x896: 1 120, 280, 72 and 20 tiles of 64x64); option igemm_lat = 3 keeps them on the 64x64 tiles: the same bits
    in p4 ... p7 and every output."""
    params = ssd.load_config(os.path.join(HERE, "golden", "config_mobilenet.json"))
    Wt = ssd.synthetic_weights(params, seed=3, logits_bias=-5.0)
    img = cuda.from_numpy(np.random.default_rng(8).integers(0, 256, (8, 640, 896, 3), dtype=np.uint8)).cuda()
    eng = ssd.Engine(params, Wt)
    a = [t.cpu().numpy() for t in eng.forward(img)]
    ta = {n: eng.get_tensor(n) for n in ("p4", "p5", "p6", "p7", "class_predictions", "encoded_boxes")}
    eng.set_option("igemm_lat", 3)
    b = [t.cpu().numpy() for t in eng.forward(img)]
    tb = {n: eng.get_tensor(n) for n in ta}
    assert int(a[3].sum()) > 50
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for n in ta:
        assert np.array_equal(ta[n], tb[n]), n
    eng.close()


@pytest.mark.parametrize("cfg,H,W", [("config_mobilenet.json", 640, 896), ("config_shufflenet.json", 640, 640)])
def test_forward_vs_oracle_full_size(cuda, ssd, oracle_graph, cfg, H, W):
    """BASELINE config 2 (MobileNet-v1 + FPN + heads at 640x896, batch 1) and config 4's
    network (ShuffleNet-v2 + FPN at 640x640) at full size."""
    params = ssd.load_config(os.path.join(HERE, "golden", cfg))
    Wt = ssd.synthetic_weights(params, seed=0, logits_bias=-4.0)
    img = np.random.default_rng(0).integers(0, 256, (1, H, W, 3), dtype=np.uint8)
    keep = {}
    ref = oracle_graph.forward(img, Wt, params, keep)
    eng = ssd.Engine(params, Wt)
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    assert out[0].shape == (1, 2000, 4) and out[1].dtype == np.int32 and out[3].dtype == np.int32
    assert stage_check(eng, keep, STAGES, "full") == 1.0, "full-size stages within tolerance but not bit-identical to the oracle"
    compare_outputs(out, ref, "full size")
    assert ref["num_boxes"][0] > 50, ref["num_boxes"]
    eng.close()


def test_detector_drop_in(cuda, ssd, oracle_graph, tmp_path):
    """inference/detector.py API: Detector(model_path)(image, score_threshold) ->
    (boxes, labels, scores)."""
    import json
    params = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 128,
              "weight_decay": 5e-5, "batch_size": 8}          # training keys are ignored
    Wt = ssd.synthetic_weights(params, seed=3, logits_bias=-4.0)
    ssd.save_weights(str(tmp_path / "model.npz"), Wt)
    with open(tmp_path / "config.json", "w") as f:
        json.dump(params, f)
    det = ssd.Detector(str(tmp_path / "model.npz"), gpu_memory_fraction=0.25, visible_device_list="0")
    img = np.random.default_rng(9).integers(0, 256, (128, 128, 3), dtype=np.uint8)
    boxes, labels, scores = det(img, score_threshold=0.2)
    ref = oracle_graph.forward(img[None], Wt, ssd.load_config(params))
    rb, rl, rs = oracle_graph.detector_call(ref, 0.2)
    assert boxes.shape == rb.shape and boxes.ndim == 2 and boxes.shape[1] == 4
    assert np.array_equal(labels, rl) and np.abs(scores - rs).max() <= TOL and np.abs(boxes - rb).max() <= TOL
    assert (scores > 0.2).all() and len(scores) > 0
    with pytest.raises(ValueError):
        det(img.astype(np.float32))
    # any image size: resize_keeping_aspect_ratio (pipeline.py:138-194) is fused into the first kernel
    for shape in [(100, 151, 3), (300, 128, 3), (128, 128, 3), (77, 201, 3)]:
        im = np.random.default_rng(shape[0]).integers(0, 256, shape, dtype=np.uint8)
        b2, l2, s2 = det(im, score_threshold=0.2)
        r2 = oracle_graph.detector_call(oracle_graph.forward(im[None], Wt, ssd.load_config(params)), 0.2)
        assert np.array_equal(l2, r2[1]) and len(l2) > 0, shape
        assert np.abs(s2 - r2[2]).max() <= TOL and np.abs(b2 - r2[0]).max() <= TOL, shape
        assert np.array_equal(b2, r2[0]) and np.array_equal(s2, r2[2]), shape
    # an image path instead of an array (north star: Detector(image_path)): read with PIL as the reference's notebooks do
    from PIL import Image
    Image.fromarray(img).save(str(tmp_path / "frame.png"))
    bp, lp, sp = det(str(tmp_path / "frame.png"), score_threshold=0.2)
    assert np.array_equal(bp, boxes) and np.array_equal(lp, labels) and np.array_equal(sp, scores)
    with pytest.raises(FileNotFoundError):
        ssd.Detector(str(tmp_path / "nope.npz"))
    # the reference's own container: a frozen GraphDef (.pb), read without TensorFlow
    write_frozen_graph(Wt, str(tmp_path / "model.pb"))
    det_pb = ssd.Detector(str(tmp_path / "model.pb"))
    b3, l3, s3 = det_pb(img, score_threshold=0.2)
    assert np.array_equal(b3, boxes) and np.array_equal(l3, labels) and np.array_equal(s3, scores)
    # ... and the same graph serialised by an encoder this repository did not write: google.protobuf's own, over TF's
    # GraphDef message family declared at test time (tests/helpers/tf_protos.py) -- `import/` prefix, /read Identity nodes,
    # consumers with list / string / bool attributes, non-float Consts, unpacked repeated fields
    from helpers.tf_protos import frozen_graph
    with open(tmp_path / "official.pb", "wb") as f:
        f.write(frozen_graph(Wt, prefix="import/", unpacked=True))
    det_pb2 = ssd.Detector(str(tmp_path / "official.pb"), config=str(tmp_path / "config.json"))
    b4, l4, s4 = det_pb2(img, score_threshold=0.2)
    assert np.array_equal(b4, boxes) and np.array_equal(l4, labels) and np.array_equal(s4, scores)
    # what the reference's TRAINING leaves behind, for a user without TensorFlow to run create_pb.py with: the model_dir
    # (checkpoint state file -> model.ckpt-N.index / .data shards, the Saver's optimizer slots and moving averages beside
    # the variables) and the SavedModel folder of create_pb.py:27-52 (ckpt_import.py)
    from helpers.tf_bundle_writer import write_bundle, crc32c_bytewise
    crc = (lambda raw: crc32c_bytewise(raw) if len(raw) < 32768 else ssd.crc32c(raw))
    mdir = tmp_path / "run00"
    mdir.mkdir()
    ema = {k + "/ExponentialMovingAverage": (v * np.float32(1.25)) for k, v in Wt.items() if not k.endswith(("moving_mean", "moving_variance"))}
    slots = {"optimizer/" + k + "/Adam": np.zeros_like(v) for k, v in list(Wt.items())[::25]}
    write_bundle(str(mdir / "model.ckpt-1200"), {**Wt, **ema, **slots, "global_step": np.array(1200, np.int64)}, num_shards=2,
                 block_size=2048, fast_crc=crc)
    (mdir / "checkpoint").write_text('model_checkpoint_path: "model.ckpt-1200"\nall_model_checkpoint_paths: "model.ckpt-1200"\n')
    with open(mdir / "config.json", "w") as f:
        json.dump(params, f)
    for where in (str(mdir), str(mdir / "model.ckpt-1200")):
        det_ck = ssd.Detector(where)
        b5, l5, s5 = det_ck(img, score_threshold=0.2)
        assert np.array_equal(b5, boxes) and np.array_equal(l5, labels) and np.array_equal(s5, scores)
    det_ema = ssd.Detector(str(mdir), use_ema=True)          # the averages are other weights: other detections
    b6, l6, s6 = det_ema(img, score_threshold=0.2)
    We = {k: ema.get(k + "/ExponentialMovingAverage", v) for k, v in Wt.items()}
    rbe, rle, rse = oracle_graph.detector_call(oracle_graph.forward(img[None], We, ssd.load_config(params)), 0.2)
    assert np.array_equal(l6, rle) and np.array_equal(s6, rse) and np.array_equal(b6, rbe) and not np.array_equal(s6, scores)
    # SSD mirror (ssd.py:10-69): raw predictions + get_predictions with other thresholds
    s = ssd.SSD(cuda.from_numpy(img[None].copy()).cuda(), det.engine)
    pred = s.get_predictions(score_threshold=0.3, iou_threshold=0.5, max_boxes_per_class=10)
    p2 = dict(params, score_threshold=0.3, iou_threshold=0.5, max_boxes_per_class=10)
    ref2 = oracle_graph.forward(img[None], Wt, ssd.load_config(p2))
    assert np.array_equal(pred["num_boxes"].cpu().numpy(), ref2["num_boxes"])
    assert np.array_equal(pred["labels"].cpu().numpy(), ref2["labels"])
    assert np.abs(pred["boxes"].cpu().numpy() - ref2["boxes"]).max() <= TOL


@pytest.mark.parametrize("cfg", ["config_mobilenet.json", "config_shufflenet.json"])
def test_fused_depthwise_pointwise_plan(cuda, ssd, libopt, cfg):
    # option fuse_dw picks the blocks that run as one dw+pw launch (MobileNet: bit i = Conv2d_{i+1};
    # ShuffleNet: non-zero = every unit): any choice gives the same bits
    params = ssd.load_config(os.path.join(HERE, "golden", cfg))
    Wt = ssd.synthetic_weights(params, seed=11, logits_bias=-4.0)
    rng = np.random.default_rng(11)
    img = cuda.from_numpy(rng.integers(0, 256, (2, 256, 384, 3), dtype=np.uint8)).cuda()
    outs = []
    for mask in ("0", "0x1fff", "0x15"):
        libopt(fuse_dw=int(mask, 0))
        eng = ssd.Engine(params, Wt)
        o = [t.cpu().numpy() for t in eng.forward(img)]
        outs.append(o + [eng.get_tensor("c3"), eng.get_tensor("c4"), eng.get_tensor("c5")])
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)


def test_forward_cached_equals_forward(cuda, ssd):
    """The serving path with persistent buffers (Engine.forward_cached: host batch -> cached device image -> records): identical
    to the plain forward, call after call."""
    params = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 128}
    Wt = ssd.synthetic_weights(params, seed=21, logits_bias=-4.0)
    eng = ssd.Engine(params, Wt)
    rng = np.random.default_rng(8)
    imgs = [rng.integers(0, 256, (2, 128, 128, 3), dtype=np.uint8) for _ in range(4)]
    eager = [[t.cpu().numpy() for t in eng.forward(cuda.from_numpy(im).cuda())] for im in imgs]
    for rep in range(2):
        for im, ref in zip(imgs, eager):
            got = [t.cpu().numpy() for t in eng.forward_cached(im)]
            for a, b in zip(got, ref):
                assert np.array_equal(a, b)
    eng.close()


def test_missing_weight_fails_loudly(cuda, ssd):
    params = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
              "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 128}
    Wt = ssd.synthetic_weights(params, seed=3)
    del Wt["fpn/p6/kernel"]
    with pytest.raises(ssd.SsdError, match="fpn/p6/kernel"):
        ssd.Engine(params, Wt)
    Wt = ssd.synthetic_weights(params, seed=3)
    Wt["fpn/p6/kernel"] = Wt["fpn/p6/kernel"][:, :, :8]
    with pytest.raises(ssd.SsdError, match="shape"):
        ssd.Engine(params, Wt)


def test_batch_independence_full_batch(cuda, ssd, oracle_graph):
    """BASELINE config 5 shard size (32 images / GPU) at 640x896: every image of a batch
    gives exactly the result of its own batch-1 run, and permuting the batch permutes the
    outputs (the path has no cross-image term).  Eight images of the batch -- four per backbone
    chain of the serving plan (128x128 tiles, two half-batch chains), first and last of each --
    are also compared with the oracle directly: outputs and every retained stage, bit for bit."""
    params = ssd.load_config(os.path.join(HERE, "golden", "config_mobilenet.json"))
    Wt = ssd.synthetic_weights(params, seed=0, logits_bias=-4.0)
    eng = ssd.Engine(params, Wt)
    rng = np.random.default_rng(1)
    imgs = rng.integers(0, 256, (32, 640, 896, 3), dtype=np.uint8)
    d = cuda.from_numpy(imgs).cuda()
    full = [t.cpu().numpy() for t in eng.forward(d)]
    perm = rng.permutation(32)
    permd = [t.cpu().numpy() for t in eng.forward(d[cuda.from_numpy(perm).cuda()].contiguous())]
    for a, b in zip(full, permd):
        assert np.array_equal(a[perm], b)
    one = [t.cpu().numpy() for t in eng.forward(d[17:18].contiguous())]
    for a, b in zip(full, one):
        assert np.array_equal(a[17:18], b)
    assert full[3].min() > 0
    # the serving plan against the oracle itself (not only against its own batch-1 run): the batch is forwarded again (the
    # batch-1 run above ran another plan; its arena is its own) and images 0, 3, 9, 15 (first chain) and 16, 22, 29, 31 (second
    # chain) are checked
    assert eng.plan_cache_stats()["plans"] == 2
    full2 = [t.cpu().numpy() for t in eng.forward(d)]
    for a, b in zip(full, full2):
        assert np.array_equal(a, b)
    stages = {n: eng.get_tensor(n) for n in STAGES}
    for i in (0, 3, 9, 15, 16, 22, 29, 31):
        keep = {}
        ref = oracle_graph.forward(imgs[i:i + 1], Wt, params, keep)
        for a, k in zip(full2, ("boxes", "labels", "scores", "num_boxes")):
            assert np.array_equal(a[i:i + 1], ref[k]), (i, k)
        for n in STAGES:
            got = stages[n][i:i + 1]
            assert np.array_equal(got, keep[n].reshape(got.shape)), (i, n)
    eng.close()


_RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import ssd_amd
from importlib import import_module
d = import_module("ssd_amd.distributed")
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
params = ssd_amd.load_config(os.path.join(sys.argv[1], "tests", "golden", "config_mobilenet.json"))
W = ssd_amd.synthetic_weights(params, seed=3, logits_bias=-5.0)
eng = ssd_amd.Engine(params, W, device=0)
g = torch.Generator().manual_seed(5)
frames = torch.randint(0, 256, (2, 128, 256, 3), dtype=torch.uint8, generator=g).cuda()
out = eng.forward(frames)
rec = d.pack_detections(*out)
got = d.gather_records(rec)            # RCCL all-gather on the records (world 1: must be the identity)
torch.cuda.synchronize()
assert torch.equal(got, rec)
back = d.unpack_detections(got)
assert all(torch.equal(a, b) for a, b in zip(back, out))
full = ssd_amd.detect_sharded(eng, frames)
assert all(torch.equal(a, b) for a, b in zip(full, out))
# the collective path itself at world 1 (force): the engine writes its records into this rank's slice of the receive buffer,
# RCCL gathers in place, the results are views of that buffer -- even shard, then a "global batch" the world does not divide
forced = ssd_amd.detect_sharded(eng, frames, total=2, force=True)
torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(forced, out)) and forced[0].data_ptr() != out[0].data_ptr()
recs = eng.new_records(2, frames.device)
again = eng.forward(frames, records=recs)
assert recs.shape == (2, 12001) and recs.element_size() * recs.shape[1] == 48004        # SURVEY 8e
assert all(torch.equal(a, b) for a, b in zip(again, out)) and again[0].data_ptr() == recs.data_ptr()
assert torch.equal(recs, d.pack_detections(*out))
dense = tuple(torch.empty_like(t.contiguous()) for t in out)
eng.forward(frames, out=dense)                  # four dense tensors of the caller's (ssd_forward)
assert all(torch.equal(a, b) for a, b in zip(dense, out))
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK", int(out[3].sum()))
"""


def test_rccl_all_gather_of_engine_records(cuda, ssd, tmp_path):
    # the N > 1 bench path minus the second GPU: backend 'nccl' (RCCL) process group in a child
    # process, all-gather of the packed detection records the engine just produced
    import subprocess
    script = tmp_path / "rccl_child.py"
    script.write_text(_RCCL_CHILD)
    port = str(29600 + os.getpid() % 2000)
    r = subprocess.run([sys.executable, str(script), os.path.dirname(HERE), port], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])

"""-m gpu: images of many sizes through ONE handle.  The reference's graph takes any image size in one session
(detector/ssd.py:27-31 "the detector supports images of various sizes", create_pb.py:24,40-47 WIDTH, HEIGHT = None, None) and
its accuracy harness feeds val2017's mix of sizes through one Detector (inference/evaluate_on_COCO.ipynb:125-150): the library
keeps one layer plan per NETWORK shape (the size after resize_keeping_aspect_ratio, pipeline.py:138-194), the source size is a
launch argument.  Every result here is compared bit for bit with the oracle and with an engine that has only ever seen that
one size."""
import os

import numpy as np
import pytest

from conftest import TINY_PARAMS
from test_gpu_forward import STAGES, compare_outputs, stage_check

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _frame(h, w, seed=None):
    return np.random.default_rng(h * 1000 + w if seed is None else seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


# (source height, width) -> what the network sees at min_dimension 640 (pipeline.py:160-192):
#   427 x 640 -> 640 x 959 -> pad 1024      480 x 640 -> 640 x 853 -> 896      640 x 480 -> 853 x 640 -> 896 x 640
#   375 x 500 -> 640 x 853 -> 896           256 x 257 -> 640 x 642.5 -> half-to-even 642 -> 768 (the long side lands on x.5)
#   1200 x 1600 -> 640 x 853 -> 896 (a camera frame: DOWN-scaling, source pixels 1.875 apart, 5.8 MB of source)
FULL_SIZES = [(427, 640), (480, 640), (640, 480), (375, 500), (256, 257), (1200, 1600)]


@pytest.mark.parametrize("hw", FULL_SIZES, ids=["%dx%d" % s for s in FULL_SIZES])
def test_any_size_at_the_networks_real_size_vs_oracle(cuda, ssd, oracle_graph, hw):
    """config_mobilenet.json (min_dimension 640) on COCO-typical frames: the up-scaling nearest-neighbour gather fused into the
    first kernel at full width, the zero pad band, box_scaler != 1 -- every retained stage and all four outputs bit-equal."""
    params = ssd.load_config(os.path.join(HERE, "golden", "config_mobilenet.json"))
    Wt = ssd.synthetic_weights(params, seed=0, logits_bias=-4.0)
    h, w = hw
    img = _frame(h, w)[None]
    nh, nw, scaler = ssd.network_input_size(h, w, params["min_dimension"])
    assert (nh, nw) != (h, w) and nh % 128 == 0 and nw % 128 == 0 and (scaler != 1.0).any()
    if hw == (256, 257):
        assert (nh, nw) == (640, 768) and abs(float(scaler[1]) - 642.0 / 768.0) < 1e-7       # tf.round: 642.5 -> 642
    keep = {}
    ref = oracle_graph.forward(img, Wt, params, keep)
    eng = ssd.Engine(params, Wt)
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    assert stage_check(eng, keep, STAGES, "%dx%d" % hw) == 1.0, "stages within tolerance but not bit-identical to the oracle"
    compare_outputs(out, ref, "%dx%d -> %dx%d" % (h, w, nh, nw))
    for a, k in zip(out, ("boxes", "labels", "scores", "num_boxes")):
        assert np.array_equal(a, ref[k]), k
    assert ref["num_boxes"][0] > 50
    st = eng.plan_cache_stats()
    assert st["plans"] == 1 and st["misses"] == 1 and st["last_network_shape"] == [nh, nw]
    eng.close()


@pytest.mark.parametrize("hw", [(427, 640), (640, 480)], ids=["427x640", "640x480"])
def test_any_size_shufflenet_full_size_vs_oracle(cuda, ssd, oracle_graph, hw):
    params = ssd.load_config(os.path.join(HERE, "golden", "config_shufflenet.json"))
    Wt = ssd.synthetic_weights(params, seed=0, logits_bias=-4.0)
    h, w = hw
    img = _frame(h, w)[None]
    nh, nw, scaler = ssd.network_input_size(h, w, params["min_dimension"])
    assert (scaler != 1.0).any()
    keep = {}
    ref = oracle_graph.forward(img, Wt, params, keep)
    eng = ssd.Engine(params, Wt)
    out = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(img).cuda())]
    assert stage_check(eng, keep, STAGES, "shufflenet %dx%d" % hw) == 1.0
    for a, k in zip(out, ("boxes", "labels", "scores", "num_boxes")):
        assert np.array_equal(a, ref[k]), k
    assert ref["num_boxes"][0] > 20
    eng.close()


# (backbone, depth multiplier): the first convolution's 32 physical channels hold 32 (MobileNet 1.0) or ShuffleNet's 24 logical
# ones; MobileNet 2.0 (64 channels) stays on the rounds-1-5 kernel either way
FIRST_CONV_NETS = [("mobilenet", 1.0), ("mobilenet", 2.0), ("shufflenet", 1.0)]


@pytest.mark.parametrize("net", FIRST_CONV_NETS, ids=["%s-%g" % n for n in FIRST_CONV_NETS])
def test_first_convolution_of_resized_frames_both_kernels(cuda, ssd, oracle_graph, net):
    """Frames that are not the network's size take the general first convolution (resize_keeping_aspect_ratio fused into it,
    pipeline.py:138-194): round 6's lane-per-pixel kernel (default) and the rounds-1-5 kernel (option first_conv_px = 0) give the
    same bits as each other and as the oracle -- up-scaling (x3.9) and down-scaling (x0.18: source pixels far apart), odd widths
    (rows that start at any byte, a buffer whose size is no multiple of 4: the last pixel's second dword), the smallest frames,
    batches, and frames of DIFFERENT sizes in one batch (per-frame geometry from the kernel's arguments)."""
    params = dict(TINY_PARAMS, backbone=net[0], depth_multiplier=net[1])
    W = ssd.synthetic_weights(params, seed=11, logits_bias=-3.0)
    # three forms of a resized frame's first layers: the fused launch with the gather in its loads (front.hip GEN: MobileNet 1.0,
    # frames whose width does not shrink -- every size below but 513 x 701 and 255 x 1021), the lane-per-pixel first convolution +
    # Conv2d_1 as its own launch (front_fuse = 0), and the rounds-1-5 first convolution (first_conv_px = 0 on top)
    new, mid, old = ssd.Engine(params, W), ssd.Engine(params, W), ssd.Engine(params, W)
    mid.set_option("front_fuse", 0)
    old.set_option("front_fuse", 0)
    old.set_option("first_conv_px", 0)
    checked = 0
    for h, w, B in [(100, 151, 1), (97, 203, 3), (513, 701, 1), (33, 77, 5), (1, 1, 2), (255, 1021, 1)]:      # (+ 48 random sizes: the sweep below)
        img = np.random.default_rng(h * 7 + w).integers(0, 256, (B, h, w, 3), dtype=np.uint8)
        a = [t.cpu().numpy() for t in new.forward(cuda.from_numpy(img).cuda())]
        for other, what in ((mid, "fused front vs first convolution + Conv2d_1"), (old, "vs the rounds-1-5 first convolution")):
            b = [t.cpu().numpy() for t in other.forward(cuda.from_numpy(img).cuda())]
            for k in range(4):
                assert np.array_equal(a[k], b[k]), (net, h, w, B, k, what)
            for name in ("c3", "p3", "class_predictions"):
                assert np.array_equal(new.get_tensor(name), other.get_tensor(name)), (net, h, w, name, what)
        if B <= 2:                                        # (the oracle costs ~0.1 s per tiny frame)
            ref = oracle_graph.forward(img, W, params)
            for a_k, key in zip(a, ("boxes", "labels", "scores", "num_boxes")):
                assert np.array_equal(a_k, ref[key]), (net, h, w, key, "vs the oracle")
            checked += int(ref["num_boxes"].sum())
    assert checked > 50
    # one batch of frames of different sizes (all -> 128 x 256): both kernels' mixed forms, against each frame alone
    frames = [np.random.default_rng(40 + i).integers(0, 256, (hh, ww, 3), dtype=np.uint8) for i, (hh, ww) in
              enumerate([(100, 151), (128, 200), (64, 100), (97, 193), (33, 65), (128, 256), (101, 202)])]
    assert len({new.network_shape(*f.shape[:2]) for f in frames}) == 1
    ma = [np.array(v) for v in new.detect_host_mixed(frames)]
    mb = [np.array(v) for v in old.detect_host_mixed(frames)]
    for i, f in enumerate(frames):
        alone = [t.cpu().numpy()[0] for t in new.forward(cuda.from_numpy(f[None]).cuda())]      # (alone: the fused launch; in the batch: per-frame geometry)
        for k in range(4):
            assert np.array_equal(ma[k][i], mb[k][i]) and np.array_equal(ma[k][i], alone[k]), (net, "mixed", i, k)
    # one plan, both forms: a frame reduced in width (first convolution + Conv2d_1) between two that are not (fused), same network shape
    seq = [np.random.default_rng(70 + i).integers(0, 256, (1, hh, ww, 3), dtype=np.uint8) for i, (hh, ww) in enumerate([(100, 151), (200, 300), (100, 151)])]
    assert len({new.network_shape(*f.shape[1:3]) for f in seq}) == 1
    before = new.plan_cache_stats()["misses"]
    outs = [[t.cpu().numpy() for t in new.forward(cuda.from_numpy(f).cuda())] for f in seq]
    assert new.plan_cache_stats()["misses"] == before              # (the batch-1 plan of 128 x 256 that served the frames "alone" above)
    refs = [[t.cpu().numpy() for t in old.forward(cuda.from_numpy(f).cuda())] for f in seq]
    for o, r in zip(outs, refs):
        for k in range(4):
            assert np.array_equal(o[k], r[k])
    for e in (new, mid, old):
        e.close()


@pytest.mark.parametrize("backbone", ["mobilenet", "shufflenet"])
def test_first_layers_of_resized_frames_random_sizes(cuda, ssd, oracle_graph, backbone):
    """A seeded sweep of source sizes (1 .. 300 in either direction: every byte alignment of a row's first pixel, spans that end at
    the last pixel of a row or of the buffer, widths that shrink and widths that do not) through the three forms of the first
    layers -- fused launch with the gather (where the width does not shrink), lane-per-pixel first convolution, rounds-1-5 first
    convolution: identical bits; every eighth size against the oracle as well.  Then the same frames as mixed-size batches (the
    fused launch with per-frame geometry) against each frame alone."""
    params = dict(TINY_PARAMS, backbone=backbone)
    W = ssd.synthetic_weights(params, seed=21, logits_bias=-3.0)
    new, mid, old = ssd.Engine(params, W), ssd.Engine(params, W), ssd.Engine(params, W)
    mid.set_option("front_fuse", 0)
    old.set_option("front_fuse", 0)
    old.set_option("first_conv_px", 0)
    rng = np.random.default_rng(2024)
    by_shape, fused_sizes = {}, 0
    for n in range(48):
        h, w = int(rng.integers(1, 301)), int(rng.integers(1, 301))
        img = rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)
        # the fused and the lane-per-pixel forms read the frame through dword loads: give them a base pointer at every offset
        # inside a dword (a caller's slice of a byte buffer; the library aligns its base itself)
        dev = [cuda.from_numpy(img).cuda()]
        for o in (n % 4, (n + 2) % 4):
            flat = cuda.empty((img.size + 8,), dtype=cuda.uint8, device="cuda")
            view = flat[o:o + img.size].view(1, h, w, 3)
            view.copy_(dev[0])
            assert view.data_ptr() % 4 == o
            dev.append(view)
        a = [t.cpu().numpy() for t in new.forward(dev[1])]
        for other, src in ((mid, dev[2]), (old, dev[0])):
            b = [t.cpu().numpy() for t in other.forward(src)]
            for k in range(4):
                assert np.array_equal(a[k], b[k]), (backbone, h, w, k)
            assert np.array_equal(new.get_tensor("c3"), other.get_tensor("c3")), (backbone, h, w)
        if n % 8 == 0:
            ref = oracle_graph.forward(img, W, params)
            for a_k, key in zip(a, ("boxes", "labels", "scores", "num_boxes")):
                assert np.array_equal(a_k, ref[key]), (backbone, h, w, key)
        nh, nw, _ = ssd.network_input_size(h, w, params["min_dimension"])
        fused_sizes += int(w <= nw * 1.0 and (h, w) != (nh, nw))
        by_shape.setdefault((nh, nw), []).append((img[0], [x[0] for x in a]))
    assert fused_sizes > 20 and len(by_shape) >= 3
    checked = 0
    for shape, items in by_shape.items():
        for k0 in range(0, len(items), 7):
            part = items[k0:k0 + 7]
            got = [np.array(v) for v in new.detect_host_mixed([f for f, _ in part])]
            for i, (_f, alone) in enumerate(part):
                for k in range(4):
                    assert np.array_equal(got[k][i], alone[k]), (backbone, shape, i, k)
            checked += len(part)
    assert checked == 48
    for e in (new, mid, old):
        e.close()


def test_cycle_of_sizes_through_one_detector_full_size(cuda, ssd, oracle_graph):
    """A, B, A, C, B, A (+ D, which shares C's network shape with another resize target, + E at the network's own size) through
    ONE Detector: every call equals a fresh single-plan engine's result and the oracle's, bit for bit; the library built one
    plan per network shape and found it again on every later call."""
    params = ssd.load_config(os.path.join(HERE, "golden", "config_mobilenet.json"))
    Wt = ssd.synthetic_weights(params, seed=0, logits_bias=-5.0)
    det = ssd.Detector(Wt, config=params)
    sizes = {"A": (480, 640), "B": (640, 427), "C": (427, 640), "D": (426, 640), "E": (640, 896)}
    net = {k: ssd.network_input_size(h, w, 640)[:2] for k, (h, w) in sizes.items()}
    assert net["C"] == net["D"] == (640, 1024) and net["A"] == net["E"] == (640, 896) and net["B"] == (1024, 640)
    frames = {k: _frame(h, w) for k, (h, w) in sizes.items()}
    want, fresh = {}, {}
    for k, f in frames.items():
        ref = oracle_graph.forward(f[None], Wt, params)
        want[k] = oracle_graph.detector_call(ref, 0.2)
        assert len(want[k][1]) > 10, k
        d1 = ssd.Detector(Wt, config=params)          # an engine that only ever sees this size
        fresh[k] = d1(f, score_threshold=0.2)
        d1.engine.close()
    order = "ABACBADCEAEB"
    for i, k in enumerate(order):
        got = det(frames[k], score_threshold=0.2)
        for a, b, c in zip(got, fresh[k], want[k]):
            assert a.dtype == b.dtype and np.array_equal(a, b), (i, k, "vs a fresh engine")
            assert np.array_equal(a, c), (i, k, "vs the oracle")
    st = det.engine.plan_cache_stats()
    # A / B / C one plan each, D runs C's plan (same network shape, other resize target), E is the identity form of A's shape
    assert st["plans"] == 4 and st["misses"] == 4 and st["hits"] == len(order) - 4 and st["evictions"] == 0, st
    assert 0 < st["arena_bytes"] <= st["budget_bytes"]
    det.engine.close()


def _tiny_detector(ssd, seed=3):
    W = ssd.synthetic_weights(TINY_PARAMS, seed=seed, logits_bias=-3.0)
    return ssd.Detector(W, config=dict(TINY_PARAMS)), W


def test_plan_cache_budget_evicts_least_recently_used(cuda, ssd, oracle_graph):
    """Option plan_cache_mb bounds the cached arenas: with room for about two plans a cycle over four network shapes keeps
    evicting, and every result stays bit-equal to the oracle; raising the budget stops the evictions."""
    det, W = _tiny_detector(ssd)
    shapes = [(128, 128), (128, 200), (260, 128), (128, 300), (100, 151)]        # -> 128x128, 128x256, 384x128, 128x384, 128x256
    frames = [_frame(h, w) for h, w in shapes]
    want = [oracle_graph.detector_call(oracle_graph.forward(f[None], W, dict(TINY_PARAMS)), 0.1) for f in frames]
    for f in frames:
        det(f, score_threshold=0.1)
    st = det.engine.plan_cache_stats()
    assert st["plans"] == 4 and st["evictions"] == 0 and st["misses"] == 4 and st["hits"] == 1, st
    per_plan = st["arena_bytes"] / st["plans"]
    det.engine.set_option("plan_cache_mb", max(1, int(2.5 * per_plan) >> 20))      # does NOT drop the plans: evicts down to the budget
    st = det.engine.plan_cache_stats()
    assert 1 <= st["plans"] < 4 and st["evictions"] >= 1 and st["arena_bytes"] <= st["budget_bytes"], st
    ev0 = st["evictions"]
    for rnd in range(3):
        for f, w in zip(frames, want):
            got = det(f, score_threshold=0.1)
            assert all(np.array_equal(a, b) for a, b in zip(got, w)), rnd
    st = det.engine.plan_cache_stats()
    assert st["evictions"] > ev0 and st["arena_bytes"] <= st["budget_bytes"], st
    det.engine.set_option("plan_cache_mb", 0)               # auto again: a quarter of the device
    for f in frames:
        det(f, score_threshold=0.1)
    ev1 = det.engine.plan_cache_stats()["evictions"]
    for f, w in zip(frames, want):
        assert all(np.array_equal(a, b) for a, b in zip(det(f, score_threshold=0.1), w))
    st = det.engine.plan_cache_stats()
    assert st["evictions"] == ev1 and st["plans"] == 4, st
    # the whole cache dropped on request: the next call of each shape rebuilds
    det.engine.plan_cache_clear()
    assert det.engine.plan_cache_stats()["plans"] == 0
    assert all(np.array_equal(a, b) for a, b in zip(det(frames[1], score_threshold=0.1), want[1]))
    det.engine.close()


def test_batches_of_mixed_shapes_and_batch_sizes_share_one_engine(cuda, ssd, oracle_graph):
    """Engine.forward on device batches whose size AND shape change from call to call, without a host wait in between (the
    results of call k are read after call k + 1 was enqueued): plans of different shapes have arenas of their own."""
    W = ssd.synthetic_weights(TINY_PARAMS, seed=8, logits_bias=-3.0)
    eng = ssd.Engine(dict(TINY_PARAMS), W)
    rng = np.random.default_rng(4)
    shapes = [(3, 128, 128, 3), (1, 100, 151, 3), (5, 128, 256, 3), (2, 200, 128, 3), (4, 140, 128, 3)]
    batches = [rng.integers(0, 256, s, dtype=np.uint8) for s in shapes]
    want = [oracle_graph.forward(b, W, dict(TINY_PARAMS)) for b in batches]
    dev = [cuda.from_numpy(b).cuda() for b in batches]
    order = [0, 1, 2, 1, 3, 0, 4, 2, 3, 4, 0]
    outs = [eng.forward(dev[k]) for k in order]             # enqueued back to back, each with a record block of its own
    cuda.cuda.synchronize()
    for k, o in zip(order, outs):
        for a, name in zip(o, ("boxes", "labels", "scores", "num_boxes")):
            assert np.array_equal(a.cpu().numpy(), want[k][name]), (k, name)
    st = eng.plan_cache_stats()
    assert st["plans"] == 5 and st["misses"] == 5 and st["hits"] == len(order) - 5
    # retained tensors are those of the LAST forward's plan
    eng.forward(dev[2])
    assert eng.get_tensor("p3").shape == (5, 16, 32, 256)
    eng.forward(dev[1])
    assert eng.get_tensor("p3").shape == (1, 16, 32, 256)
    eng.close()


def test_precision_and_option_changes_drop_every_cached_plan(cuda, ssd):
    det, W = _tiny_detector(ssd, seed=4)
    f1, f2 = _frame(128, 128), _frame(128, 200)
    a1, a2 = det(f1, 0.1), det(f2, 0.1)
    assert det.engine.plan_cache_stats()["plans"] == 2
    det.engine.set_option("streams", 1)
    assert det.engine.plan_cache_stats()["plans"] == 0
    b1, b2 = det(f1, 0.1), det(f2, 0.1)
    assert all(np.array_equal(x, y) for x, y in zip(a1 + a2, b1 + b2))
    det.engine.set_precision("f16x3")
    assert det.engine.plan_cache_stats()["plans"] == 0
    c1 = det(f1, 0.1)
    assert np.array_equal(c1[1], a1[1]) and np.abs(c1[2] - a1[2]).max() <= 1e-4
    det.engine.set_precision("f32")
    d2 = det(f2, 0.1)
    assert all(np.array_equal(x, y) for x, y in zip(a2, d2))
    with pytest.raises(ssd.SsdError, match="plan_cache_mb"):
        det.engine.set_option("plan_cache_mb", -1)
    with pytest.raises(ssd.SsdError, match="igemm_tile"):
        det.engine.set_option("igemm_tile", 29)            # a wave tile only the diagnostics build has
    det.engine.close()

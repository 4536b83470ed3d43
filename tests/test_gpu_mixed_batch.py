"""-m gpu: frames of DIFFERENT sizes as one batch (ssd_forward_mixed).  The reference runs one `sess.run` per image because frames of
different sizes do not form a tensor (inference/evaluate_on_COCO.ipynb:125-150); what the network sees is the size after
resize_keeping_aspect_ratio (pipeline.py:138-194), which frames of many source sizes share.  Image b of a mixed batch must be bit
for bit what frame b gives alone -- and what the oracle gives for it."""
import os

import numpy as np
import pytest

from conftest import TINY_PARAMS

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
# at min_dimension 128 all of these become 128 x 256: pad only, up-scale by 1.28 (193 columns), identity, up-scale by 2, one column over
TINY_SIZES = [(128, 200), (100, 151), (128, 256), (64, 100), (128, 129)]


def _frame(h, w, seed=0):
    return np.random.default_rng(h * 1000 + w + seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def _oracle_rows(oracle_graph, W, params, frame):
    ref = oracle_graph.forward(frame[None], W, params)
    return [ref[k][0] for k in ("boxes", "labels", "scores", "num_boxes")]


@pytest.mark.parametrize("backbone", ["mobilenet", "shufflenet"])
def test_mixed_batch_equals_each_frame_alone_and_the_oracle(cuda, ssd, oracle_graph, backbone):
    params = dict(TINY_PARAMS, backbone=backbone)
    W = ssd.synthetic_weights(params, seed=5, logits_bias=-3.0)
    eng = ssd.Engine(params, W)
    frames = [_frame(h, w) for h, w in TINY_SIZES]
    assert len({eng.network_shape(*f.shape[:2]) for f in frames}) == 1 and eng.network_shape(100, 151) == (128, 256)
    alone = [[t.cpu().numpy()[0] for t in eng.forward(cuda.from_numpy(f[None]).cuda())] for f in frames]
    want = [_oracle_rows(oracle_graph, W, params, f) for f in frames]
    assert sum(int(w[3]) for w in want) > 20
    # host-fed form (two backbone chains: 5 frames), twice (the second call finds the plan and the staging buffer)
    for rnd in range(2):
        got = [np.array(v) for v in eng.detect_host_mixed(frames)]
        for b in range(len(frames)):
            for k in range(4):
                assert np.array_equal(got[k][b], alone[b][k]), (backbone, rnd, b, k, "vs the frame alone")
                assert np.array_equal(got[k][b], want[b][k]), (backbone, rnd, b, k, "vs the oracle")
    # device form: a list of device tensors, in another order, and the pre-laid-out form with explicit (unordered, padded) offsets
    order = [3, 0, 4, 2, 1]
    out = [t.cpu().numpy() for t in eng.forward_mixed([cuda.from_numpy(frames[i]).cuda() for i in order])]
    for j, i in enumerate(order):
        for k in range(4):
            assert np.array_equal(out[k][j], alone[i][k]), (backbone, "device list", i, k)
    sizes = [f.size for f in frames]
    offs = [4096 * 50, 0, 4096 * 20, 4096 * 75, 4096 * 90 + 3]            # any byte offsets (not overlapping), any order
    flat = np.zeros(4096 * 110, np.uint8)
    for f, o in zip(frames, offs):
        flat[o:o + f.size] = f.reshape(-1)
    out = [t.cpu().numpy() for t in eng.forward_mixed((cuda.from_numpy(flat).cuda(), [f.shape[:2] for f in frames], offs))]
    for b in range(len(frames)):
        for k in range(4):
            assert np.array_equal(out[k][b], alone[b][k]), (backbone, "offsets", b, k)
    assert sizes[0] != sizes[1]
    # one plan served every mixed batch of this shape and size; the batch-1 runs above used their own
    st = eng.plan_cache_stats()
    assert st["evictions"] == 0 and st["last_network_shape"] == [128, 256]
    eng.close()


def test_mixed_batch_refusals_and_limits(cuda, ssd):
    W = ssd.synthetic_weights(TINY_PARAMS, seed=5, logits_bias=-3.0)
    eng = ssd.Engine(dict(TINY_PARAMS), W)
    with pytest.raises(ssd.SsdError, match="share the network shape"):
        eng.detect_host_mixed([_frame(128, 200), _frame(200, 128)])
    with pytest.raises(ssd.SsdError, match="frames per mixed-size batch"):
        eng.detect_host_mixed([_frame(128, 200)] * 65)
    with pytest.raises(ValueError):
        eng.detect_host_mixed([_frame(128, 200).astype(np.float32)])
    # the largest batch the argument table holds, with sub-batch plans forced (option nsub): every image equals its own run
    frames = [_frame(*TINY_SIZES[i % len(TINY_SIZES)], seed=i) for i in range(64)]
    ref = {}
    for i in range(5):
        ref[i] = [t.cpu().numpy()[0] for t in eng.forward(cuda.from_numpy(frames[i][None]).cuda())]
    full = [np.array(v) for v in eng.detect_host_mixed(frames)]
    eng.set_option("nsub", 3)
    split = [np.array(v) for v in eng.detect_host_mixed(frames)]
    for a, b in zip(full, split):
        assert np.array_equal(a, b)
    for i in range(5):
        for k in range(4):
            assert np.array_equal(full[k][i], ref[i][k]), (i, k)
    # a frame alone through the mixed entry point = the ordinary forward
    one = [np.array(v) for v in eng.detect_host_mixed([frames[1]])]
    for k in range(4):
        assert np.array_equal(one[k][0], ref[1][k])
    eng.close()


def test_mixed_batch_at_the_networks_real_size(cuda, ssd, oracle_graph):
    """config_mobilenet.json (min_dimension 640): 480x640, 375x500, 427x580 and a frame at the network's own size, all -> 640x896."""
    params = ssd.load_config(os.path.join(HERE, "golden", "config_mobilenet.json"))
    W = ssd.synthetic_weights(params, seed=0, logits_bias=-5.0)
    det = ssd.Detector(W, config=params)
    sizes = [(480, 640), (375, 500), (640, 896), (427, 580)]
    frames = [_frame(h, w) for h, w in sizes]
    assert {det.engine.network_shape(h, w) for h, w in sizes} == {(640, 896)}
    got = det.detect_many(frames, score_threshold=0.2)
    for f, g in zip(frames, got):
        ref = oracle_graph.forward(f[None], W, params)
        want = oracle_graph.detector_call(ref, 0.2)
        alone = det(f, score_threshold=0.2)
        assert len(want[1]) > 10
        for a, b, c in zip(g, alone, want):
            assert a.dtype == b.dtype and np.array_equal(a, b) and np.array_equal(a, c), f.shape
    det.engine.close()


def test_detect_many_groups_by_network_shape_and_keeps_the_order(cuda, ssd):
    W = ssd.synthetic_weights(TINY_PARAMS, seed=9, logits_bias=-3.0)
    det = ssd.Detector(W, config=dict(TINY_PARAMS))
    rng = np.random.default_rng(4)
    pool = TINY_SIZES + [(128, 128), (200, 128), (260, 128), (128, 300), (90, 90), (128, 384)]
    imgs = [_frame(*pool[int(rng.integers(len(pool)))], seed=i) for i in range(40)]
    want = [det(im, score_threshold=0.1) for im in imgs]
    assert sum(len(w[2]) for w in want) > 100
    for mb in (32, 3, 1):
        got = det.detect_many(imgs, score_threshold=0.1, max_batch=mb)
        assert len(got) == len(imgs)
        for i, (g, w) in enumerate(zip(got, want)):
            assert all(a.dtype == b.dtype and np.array_equal(a, b) for a, b in zip(g, w)), (mb, i, imgs[i].shape)
    assert det.detect_many([], 0.1) == []
    det.engine.close()

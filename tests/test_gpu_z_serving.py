"""The boundary under serving conditions (inference/detector.py:34,51-52): a Detector shared by host threads the way a
tf.Session may be, host-fed batches pipelined against the compute, and bench.py's own collective path on one GPU."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import TINY_PARAMS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _detector(ssd, precision="f32", seed=3):
    W = ssd.synthetic_weights(TINY_PARAMS, seed=seed, logits_bias=-3.0)
    return ssd.Detector(W, config=dict(TINY_PARAMS), precision=precision)


def test_two_threads_share_one_detector(cuda, ssd):
    """tf.Session.run is thread-safe and the reference's Detector holds one session (inference/detector.py:34,52): two
    Python threads calling ONE Detector on different images get, each, exactly what a single-threaded run returns."""
    det = _detector(ssd)
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, (128, 128, 3), dtype=np.uint8), rng.integers(0, 256, (140, 128, 3), dtype=np.uint8),
            rng.integers(0, 256, (128, 200, 3), dtype=np.uint8)]
    want = [det(im, score_threshold=0.1) for im in imgs]
    assert sum(len(w[2]) for w in want) > 10
    errors = []

    def worker(tid):
        try:
            cuda.cuda.set_device(0)
            with cuda.cuda.stream(cuda.cuda.Stream()):          # each thread on a stream of its own: the library orders the arena
                for it in range(40):
                    k = (tid + it) % len(imgs)
                    got = det(imgs[k], score_threshold=0.1)
                    for a, b in zip(got, want[k]):
                        if not np.array_equal(a, b):
                            errors.append((tid, it, k))
                            return
        except Exception as e:      # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors[:3]
    # ... and batches through detect_batch from two threads
    batch = np.stack([imgs[0], imgs[0][::-1].copy()])
    ref = det.detect_batch(batch)
    res = [None, None]

    def worker2(tid):
        for _ in range(10):
            res[tid] = det.detect_batch(batch)

    threads = [threading.Thread(target=worker2, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    for r in res:
        assert all(np.array_equal(a, b) for a, b in zip(r, ref))


@pytest.mark.parametrize("fence", [0, 1])
def test_forwards_alternating_between_two_streams_without_host_sync(cuda, ssd, fence):
    """One Engine, forwards enqueued alternately on two NON-BLOCKING streams with no host synchronisation in between: the
    library's own ordering (ev_last recorded behind every forward of a handle that has seen two streams, the next forward
    waits for it) is all that keeps forward k + 1 from overwriting the arena forward k is still reading.  Same shape every
    time (one plan, one arena); unrelated long work sits on the other stream in front of its next forward; with the default
    ordering events and after ssd_set_option(event_fence)."""
    W = ssd.synthetic_weights(TINY_PARAMS, seed=6, logits_bias=-3.0)
    eng = ssd.Engine(dict(TINY_PARAMS), W)
    if fence:
        eng.set_option("event_fence", 1)
    rng = np.random.default_rng(31)
    imgs = [cuda.from_numpy(rng.integers(0, 256, (4, 128, 256, 3), dtype=np.uint8)).cuda() for _ in range(6)]
    want = []
    for im in imgs:                                  # single stream, synchronised: the reference results
        want.append([t.cpu().numpy() for t in eng.forward(im)])
    cuda.cuda.synchronize()
    assert sum(int(w[3].sum()) for w in want) > 30
    s = [cuda.cuda.Stream(), cuda.cuda.Stream()]
    big = cuda.randn(4096, 4096, device="cuda")
    recs = [eng.new_records(4, imgs[0].device) for _ in range(24)]
    cuda.cuda.synchronize()
    outs = []
    for k in range(24):
        st = s[k & 1]
        with cuda.cuda.stream(st):
            if k % 5 == 2:                           # unrelated long work queued on this stream in front of the forward
                for _ in range(4):
                    big = big @ big * 1e-4
            outs.append(eng.forward(imgs[k % 6], records=recs[k]))
    cuda.cuda.synchronize()                          # the first host wait
    for k, o in enumerate(outs):
        for a, b in zip(o, want[k % 6]):
            assert np.array_equal(a.cpu().numpy(), b), (k, fence)
    eng.close()


def test_forward_on_a_capturing_stream_after_another_stream_is_refused(cuda, ssd):
    """A handle whose previous forward ran on another stream cannot be captured into a caller's graph (the wait for that
    forward's event would tie the graph to work outside it): the library says so instead of surfacing a HIP capture error
    mid-enqueue; on a handle that only ever ran on the capturing stream, capture works (option streams = 1)."""
    W = ssd.synthetic_weights(TINY_PARAMS, seed=6, logits_bias=-3.0)
    eng = ssd.Engine(dict(TINY_PARAMS), W)
    eng.set_option("streams", 1)
    img = cuda.from_numpy(np.random.default_rng(2).integers(0, 256, (1, 128, 128, 3), dtype=np.uint8)).cuda()
    want = [t.cpu().numpy() for t in eng.forward(img)]
    s1, s2 = cuda.cuda.Stream(), cuda.cuda.Stream()
    rec = eng.new_records(1, img.device)
    with cuda.cuda.stream(s1):
        eng.forward(img, records=rec)
    with cuda.cuda.stream(s2):
        eng.forward(img, records=rec)                # the handle has now seen two streams
    with cuda.cuda.stream(s1):
        eng.forward(img, records=rec)                # the previous forward is on s1
    cuda.cuda.synchronize()
    g = cuda.cuda.CUDAGraph()
    refused = False
    with cuda.cuda.graph(g, stream=s2):
        try:
            eng.forward(img, records=rec)
        except ssd.SsdError as e:
            refused = "capturing" in str(e)
    assert refused
    cuda.cuda.synchronize()
    # ... and nothing is broken afterwards
    got = [t.cpu().numpy() for t in eng.forward(img)]
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    eng.close()


def test_detect_stream_equals_synchronous_calls_in_order(cuda, ssd):
    """Detector.detect_stream (two pinned staging buffers, copy streams: H2D of batch k+1 and D2H of batch k-1 under the
    compute of batch k) returns, in order, exactly what detect_batch returns for each batch -- including across a change of
    the batch shape in mid-stream and for a single batch."""
    det = _detector(ssd, seed=5)
    rng = np.random.default_rng(21)
    shapes = [(3, 128, 128, 3)] * 5 + [(2, 128, 160, 3)] * 3 + [(3, 128, 128, 3)] * 2
    batches = [rng.integers(0, 256, s, dtype=np.uint8) for s in shapes]
    want = [det.detect_batch(b) for b in batches]
    assert sum(int(w[3].sum()) for w in want) > 20
    got = list(det.detect_stream(iter(batches)))
    assert len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        for a, b in zip(g, w):
            assert a.dtype == b.dtype and np.array_equal(a, b), k
    one = list(det.detect_stream([batches[0]]))
    assert len(one) == 1 and all(np.array_equal(a, b) for a, b in zip(one[0], want[0]))
    assert list(det.detect_stream([])) == []
    # the results are copies: a later batch does not overwrite an earlier result
    it = det.detect_stream(iter(batches[:4]))
    first = next(it)
    rest = list(it)
    assert all(np.array_equal(a, b) for a, b in zip(first, want[0])) and len(rest) == 3


def test_detector_call_reads_pinned_views_safely(cuda, ssd):
    """Detector.__call__ filters views of the engine's pinned result block: what it returns must be copies."""
    det = _detector(ssd)
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)
    ra = det(a, score_threshold=0.05)
    keep = [x.copy() for x in ra]
    rb = det(b, score_threshold=0.05)
    assert all(np.array_equal(x, y) for x, y in zip(ra, keep))
    assert not np.array_equal(ra[2], rb[2])


def test_detector_one_call_path_equals_numpy_filter(cuda, ssd):
    """Detector.__call__ in mode f32 is one library call (ssd_detect_host: upload, forward, wait, score filter in C); the
    same frames through detect_host + the numpy filter of inference/detector.py:54-58 give the same arrays -- at several
    thresholds, incl. one that keeps nothing and one that keeps everything."""
    det = _detector(ssd)
    rng = np.random.default_rng(17)
    for shape in [(128, 128, 3), (100, 151, 3)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        for thr in (0.0, 0.05, 0.3, 0.999999):
            det.engine.one_call_detect = True
            a = det(img, score_threshold=thr)
            det.engine.one_call_detect = False
            b = det(img, score_threshold=thr)
            assert len(a) == 3 and all(x.dtype == y.dtype and np.array_equal(x, y) for x, y in zip(a, b)), (shape, thr)
            assert a[0].shape == (len(a[2]), 4) and a[1].dtype == np.int32
        assert len(det(img, score_threshold=0.999999)[2]) == 0
    det.engine.one_call_detect = True


@pytest.mark.parametrize("B", [1, 3, 6])
def test_detect_host_outputs_written_straight_into_pinned_memory(cuda, ssd, B):
    """Up to Engine.zero_copy_max_batch images detect_host hands the PINNED result block to the forward as its output
    pointers (the last kernel writes over PCIe, no device-to-host copy); beyond, a device block and one copy.  Both forms,
    and the plain device-tensor forward, return the same bytes -- every row, the zero padding included."""
    det = _detector(ssd)
    eng = det.engine
    rng = np.random.default_rng(17 + B)
    imgs = rng.integers(0, 256, (B, 128, 160, 3), dtype=np.uint8)
    ref = [t.cpu().numpy() for t in eng.forward(cuda.from_numpy(imgs).cuda())]
    assert int(ref[3].sum()) > 5
    for zc in (4, 0):
        eng.zero_copy_max_batch = zc
        for _ in range(2):           # (twice: the second call finds the previous results in the block)
            got = [np.array(v) for v in eng.detect_host(imgs)]
            for a, b in zip(got, ref):
                assert np.array_equal(a, b), (zc, B)


def _bench(args, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    return json.loads(lines[0])


def test_bench_collective_path_on_rccl_world_1(cuda):
    """bench.py's own distributed code -- init_process_group('nccl', device_id=...), the ranks_seen all-gather, the
    all-gather of the detection records (pack -> RCCL -> unpack), the all-reduce(MAX) of time and status, the final
    barrier -- in a fresh child process on this box's one GPU, so that the driver's 8-GPU run is not its first contact
    with RCCL.  Functional asserts only: no wall-clock ratio is asserted anywhere under tests/ (boxes of the pool differ
    by 1-2 % and a noisy lease must not turn the parity rows behind this file into "untested"); the two throughputs are
    printed side by side by scripts/ab_dist.sh instead."""
    common = ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-latency", "--no-shufflenet", "--no-other-precision", "--sustained-seconds", "0"]
    plain = _bench(common)
    # the form the driver uses for N > 1, at N = 1: torch.distributed.run sets WORLD_SIZE = 1 and the collective path runs
    # (bench.py --force-dist is the same path without the launcher; scripts/ab_dist.sh uses it)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    forced = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert plain["collective_path"] is False and forced["collective_path"] is True
    assert forced["n_gpus"] == 1 and forced["ranks_seen"] == [0] and forced["config"]["shards"] == [[0, 32]]
    assert forced["config"]["detections_per_image"] == plain["config"]["detections_per_image"] > 50
    assert forced["value"] > 0 and plain["value"] > 0
    # the line explains a scaling loss by itself: per-rank step time, compute / all-gather split from events, the queue note
    for line in (plain, forced):
        for k in ("per_rank_ms_per_step", "per_rank_compute_ms_per_step", "per_rank_allgather_ms_per_step"):
            assert isinstance(line[k], list) and len(line[k]) == 1, k
        assert line["compute_ms_per_step"] > 0 and line["allgather_ms_per_step"] >= 0
        assert line["compute_ms_per_step"] + line["allgather_ms_per_step"] < 1.5 * line["ms_per_step"]
    assert forced["engine_first_forward_before_process_group"] is True and plain["engine_first_forward_before_process_group"] is False
    assert forced["allgather_bytes_per_rank"] == 32 * 48004
    # roofline.traffic: the plain run re-measures it (two rocprofv3 --pmc child passes behind the timed legs); the collective run
    # carries the committed measurement.  The counter bytes of a tower launch sit between its algorithmic bytes and 3x that.
    rl = plain["roofline"]
    if "traffic_live_error" in rl:          # no profiler on this box / it timed out: the line says so and carries the committed value
        assert "profiles/traffic.json" in rl["traffic_source"] and rl["traffic"] > 0, rl["traffic_live_error"]
    else:
        assert rl["traffic_source"].startswith("measured in this run")
        assert rl["traffic_algorithmic_bytes"] < rl["traffic"] < 3 * rl["traffic_algorithmic_bytes"] and rl["traffic_launches_averaged"] >= 20
        assert abs(rl["traffic"] / rl["traffic_committed"] - 1) < 0.05
    assert "profiles/traffic.json" in forced["roofline"]["traffic_source"]

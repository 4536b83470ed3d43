#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: images/sec at 896x640 (W x H), MobileNet-v1 RetinaNet,
on N MI355X of one node (weak scaling: a fixed shard of images per GPU), plus the p50
per-image latency of the reference's own batch-1 protocol and, at N = 1, BASELINE config 4
(ShuffleNet-v2 + FPN, 640x640, batch 64) as a second object on the same line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

`--gpus N` with N > 1 starts itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N` (a child process, before anything touches the GPU) unless it already runs
inside such a launch (WORLD_SIZE set), so both `python bench.py --gpus 8` and the explicit
torchrun form work.

One step = one pass of the whole hot path (uint8 frames already resident in HBM -> backbone
-> FPN -> heads -> decode -> per-class NMS -> padded detections; for N > 1 followed by the
RCCL all-gather of the detection records) over one batch of synthetic frames per GPU.
`value` is measured in the reference's arithmetic: every convolution an exact fp32 chain
(precision mode f32, bit-identical to the CPU oracle).  The opt-in mode f16x3 is reported
beside it as `other_precision`, never as `value`.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import ssd_amd  # noqa: E402

H, W = 640, 896                       # inference/just_try_detector.ipynb:111 resize((896, 640))
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# precision f16x3: every fp32 product costs three v_mfma_f32_32x32x16_f16 terms (xh*wh + xh*wl + xl*wh), so the
# algorithmic-FLOP peak of that kernel is the dense F16 MFMA peak (2.5 PFLOP/s = 16 x 157.3) divided by 3
PEAK_F16X3_TFLOPS = 16 * 157.3 / 3.0
HBM_PEAK_TBS = 8.0                    # MI355X_MICROARCH.md: spec; 6.29 measured copy
PARAMS = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80,
          "score_threshold": 0.15, "iou_threshold": 0.6, "max_boxes_per_class": 25,
          "min_dimension": 640}       # config_mobilenet.json:7-12,21
PARAMS_SHUFFLE = dict(PARAMS, backbone="shufflenet")    # config_shufflenet.json
# per-layer max(FLOP / 157.3 TFLOP/s, bytes / 8 TB/s) summed, ms per image
ROOFLINE_MS = {"mobilenet": 1.113, "shufflenet": 0.739}
GFLOP_PER_IMAGE = {"mobilenet": 169.957, "shufflenet": 113.719}
# Random-init heads have no trained sparsity: with the tower activations of the seeded weights the
# logits are ~N(bias, 1.6).  -7.5 puts ~6.5k (anchor, class) scores above score_threshold per
# image and ~190 detections after NMS -- a busy-scene RetinaNet output.  (The reference's own
# init, -log(99), would leave 250k candidates / 1600 detections per image: tests/ use such
# dense settings as an NMS stress, the benchmark does not.)  See DESIGN.md section 6.
LOGITS_BIAS = {"mobilenet": -7.5, "shufflenet": -7.5}

KERNEL_NAMES = {
    "conv3x3_f16x3_tile256": "igemm16_kernel<9> (3x3 convs of the head towers + fpn p3, 256x256 tiles, 3 x f16 MFMA per product)",
    "conv3x3_mfma": "igemm_kernel<...,9> (3x3 convs: FPN outputs + head towers + class/box heads)"}


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (a port of the reference graph, oracle/) on this host's cores, one
    640x896 frame at a time (the reference graph is batch 1), bounded to ~budget_s.  The
    OpenMP thread count is calibrated first (one frame each): on a many-core, multi-socket
    host the batch-1 graph runs fastest well below the full core count."""
    from oracle import graph, ops
    ops.build()
    Wt = ssd_amd.synthetic_weights(PARAMS, seed=0, logits_bias=LOGITS_BIAS["mobilenet"])
    img = np.random.default_rng(0).integers(0, 256, (1, H, W, 3), dtype=np.uint8)

    def frame():
        t0 = time.perf_counter()
        graph.forward(img, Wt, PARAMS)
        return time.perf_counter() - t0

    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu} or {ncpu})
    best_t, best = cands[0], None
    for t in cands:
        ops.set_num_threads(t)
        frame()                                          # warm-up (threads, page faults)
        dt = frame()
        if best is None or dt < best:
            best_t, best = t, dt
        if dt > 3.0:
            break
    ops.set_num_threads(best_t)
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 30 and (len(times) < 3 or time.perf_counter() < t_end):
        times.append(frame())
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "img/s", "cores": best_t, "kind": "port",
            "sample": "%d frames of 640x896, batch 1, C oracle = CPU restatement of the TF1.12 reference path "
                      "(OpenMP %d threads of %d logical CPUs, AVX2 fmaf chains), median %.3f s/frame"
                      % (len(times), best_t, ncpu, med)}


def latency_batch1(detector):
    """inference/just_try_detector.ipynb:149-155, through the boundary class itself: 110 calls of
    `Detector.__call__(image, score_threshold=0.5)` on one host uint8 image (pinned staging + H2D, graph, ONE packed
    D2H, score filter; numpy in -> numpy out), first 10 dropped."""
    img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    runs = []
    for _ in range(3):          # the protocol three times: a busy host (another tenant's burst) shows as one run apart from the others
        times = []
        for _ in range(110):
            t0 = time.perf_counter()
            boxes, labels, scores = detector(img, score_threshold=0.5)
            times.append(time.perf_counter() - t0)
        t = np.array(times[10:]) * 1e3
        runs.append({"p50_ms": float(np.percentile(t, 50)), "mean_ms": float(t.mean()), "std_ms": float(t.std())})
    mid = sorted(runs, key=lambda r: r["p50_ms"])[1]
    return dict(mid, runs_p50_ms=[r["p50_ms"] for r in runs],
                protocol="Detector.__call__, host ndarray in -> filtered numpy out, 110 calls, first 10 dropped; the protocol run "
                         "three times, the run with the median p50 reported (all three p50s in runs_p50_ms)",
                **{"detections_over_0.5": int(len(scores))})


# (height, width) of frequent COCO val2017 frames, portrait and landscape, plus one at the network's own size
# (inference/evaluate_on_COCO.ipynb:125-150 feeds such a mix through ONE Detector)
MIXED_SIZES = [(480, 640), (640, 480), (427, 640), (640, 427), (375, 500), (500, 375), (360, 640), (333, 500),
               (640, 428), (612, 612), (426, 640), (500, 333), (640, 896)]


def latency_mixed_sizes(detector, cycles=8, alone_calls=20):
    """Detector.__call__ over a cyclic mix of image sizes (every call another size than the one before) next to the same
    sizes each timed alone.  The library keeps one layer plan per NETWORK shape (the size after resize_keeping_aspect_ratio),
    so after the first pass no call builds anything: `plan_cache` carries the hits / misses / evictions of the timed part."""
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in MIXED_SIZES]
    eng = detector.engine
    alone = {}
    for (h, w), f in zip(MIXED_SIZES, frames):
        for _ in range(3):
            detector(f, score_threshold=0.5)
        t = []
        for _ in range(alone_calls):
            t0 = time.perf_counter()
            detector(f, score_threshold=0.5)
            t.append((time.perf_counter() - t0) * 1e3)
        nh, nw, _ = ssd_amd.network_input_size(h, w, PARAMS["min_dimension"])
        alone["%dx%d" % (h, w)] = {"network": [nh, nw], "p50_ms": float(np.percentile(t, 50))}
    for f in frames:
        detector(f, score_threshold=0.5)
    s0 = eng.plan_cache_stats()
    t = []
    for _ in range(cycles):
        for f in frames:
            t0 = time.perf_counter()
            detector(f, score_threshold=0.5)
            t.append((time.perf_counter() - t0) * 1e3)
    s1 = eng.plan_cache_stats()
    alone_mean = float(np.mean([v["p50_ms"] for v in alone.values()]))
    return {"p50_ms": float(np.percentile(t, 50)), "p95_ms": float(np.percentile(t, 95)), "mean_ms": float(np.mean(t)), "calls": len(t),
            "source_sizes": len(MIXED_SIZES), "network_shapes_seen": sorted({tuple(v["network"]) for v in alone.values()}),
            "alone_p50_ms": alone, "alone_mean_of_p50_ms": alone_mean,
            # every size appears equally often in the mix: the mean over the mix against the mean of the sizes' own p50s
            "mixed_mean_over_alone_mean": float(np.mean(t)) / alone_mean,
            "plan_cache": {"plans": s1["plans"], "arena_mb": s1["arena_bytes"] / 2 ** 20, "budget_mb": s1["budget_bytes"] / 2 ** 20,
                           "hits_in_timed_part": s1["hits"] - s0["hits"], "misses_in_timed_part": s1["misses"] - s0["misses"],
                           "evictions_in_timed_part": s1["evictions"] - s0["evictions"]},
            "protocol": "Detector.__call__(host ndarray, 0.5); %d source sizes x %d cycles, each call another size than the one before; "
                        "alone = %d calls of one size after 3 warm-up calls" % (len(MIXED_SIZES), cycles, alone_calls)}


def throughput_mixed_sizes(detector, n_images=260, max_batch=32):
    """A stream of `n_images` host frames of the 13 COCO-typical sizes in shuffled order through ONE Detector: one call per image (the
    reference's loop, inference/evaluate_on_COCO.ipynb:125-150) against Detector.detect_many (frames grouped by the size the
    network sees; frames of DIFFERENT source sizes in one batch, ssd_forward_mixed).  Same results, bit for bit (checked here)."""
    rng = np.random.default_rng(1)
    order = rng.permutation(np.repeat(np.arange(len(MIXED_SIZES)), -(-n_images // len(MIXED_SIZES))))[:n_images]
    frames = [rng.integers(0, 256, MIXED_SIZES[i] + (3,), dtype=np.uint8) for i in order]
    for f in frames[:26]:
        detector(f, score_threshold=0.5)
    t0 = time.perf_counter()
    one = [detector(f, score_threshold=0.5) for f in frames]
    t_one = time.perf_counter() - t0
    detector.detect_many(frames, score_threshold=0.5, max_batch=max_batch)      # builds the batched plans
    t0 = time.perf_counter()
    many = detector.detect_many(frames, score_threshold=0.5, max_batch=max_batch)
    t_many = time.perf_counter() - t0
    same = all(np.array_equal(a, b) for x, y in zip(one, many) for a, b in zip(x, y))
    groups = {}
    for f in frames:
        k = ssd_amd.network_input_size(f.shape[0], f.shape[1], PARAMS["min_dimension"])[:2]
        groups[k] = groups.get(k, 0) + 1
    return {"images": n_images, "source_sizes": len(MIXED_SIZES), "one_call_per_image_img_s": n_images / t_one,
            "detect_many_img_s": n_images / t_many, "speedup": t_one / t_many, "max_batch": max_batch,
            "results_identical": bool(same), "frames_per_network_shape": {"%dx%d" % k: v for k, v in sorted(groups.items())},
            "note": "host frames in, filtered numpy detections out, both ways; never `value` (that is 32 resident frames of the network's own size)"}


def latency_segments(detector):
    """Where a batch-1 Detector call spends its time (attribution: every segment followed by a synchronisation, so the
    sum exceeds the pipelined call -- ssd_forward_host stages and uploads the image in pieces, the upload of one under the
    host copy of the next): staging memcpy, upload, forward (results into pinned host memory), score filter."""
    e = detector.engine
    img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    for _ in range(5):
        detector(img, score_threshold=0.5)
    slot = e._out_slot(1)
    _, dev_in, pin_in = e._in_slot((1, H, W, 3), index=3, pinned=True)
    pin_in_np = pin_in.numpy()
    seg = {k: [] for k in ("copyto_pinned", "h2d", "forward_zero_copy_out", "filter")}
    for _ in range(50):
        t0 = time.perf_counter(); np.copyto(pin_in_np, img[None]); t1 = time.perf_counter()
        dev_in.copy_(pin_in, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
        e.forward(dev_in, records=slot["pin_out"]); torch.cuda.synchronize(); t3 = time.perf_counter()
        b, l, s, n = slot["host"]; k = s[0][:n[0]] > 0.5; _ = b[0][:n[0]][k], l[0][:n[0]][k], s[0][:n[0]][k]; t4 = time.perf_counter()
        for name, a, c in zip(seg, (t0, t1, t2, t3), (t1, t2, t3, t4)):
            seg[name].append((c - a) * 1e6)
    return {k: float(np.percentile(v, 50)) for k, v in seg.items()}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv, script):
    """`python bench.py --gpus N` typed as is: re-run this script under torch.distributed.run, one
    rank per GPU, as a CHILD process (never exec: nothing here has touched the GPU yet, and the
    parent never does), and return its exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    return subprocess.call(cmd, env=env)


def dominant(prof):
    # f16x3: the 256x256-tile kernel (8 tower launches + fpn p3 per step); f32: the 3x3 implicit-GEMM class
    k = "conv3x3_f16x3_tile256" if prof.get("conv3x3_f16x3_tile256", {}).get("launches", 0) > 0 else "conv3x3_mfma"
    return k, prof[k]


DOMINANT_F32_KERNEL = "igemm_kernel<2, 2, 2, 2, 9, 0, 0"      # the 128x128-tile 3x3 instance: 8 tower layers + fpn p3 + fpn p4 per step


def measure_traffic_live(config, timeout_s=90):
    """HBM bytes per launch of the dominant kernel from the PMC counters, measured NOW: two child runs of this script (2 timed
    steps, mode f32 only, every other leg off) under `rocprofv3 --pmc <counter> --kernel-trace`, one counter per pass
    (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 -- they do not fit one pass), from /tmp with the
    program itself behind `--`; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the guide's gfx950 correction: FETCH_SIZE reports
    half of a wide coalesced read stream; both counters are in KB).  Returns (bytes per launch, launches averaged, note) or
    (None, 0, reason) -- then the line keeps the committed measurement of profiles/traffic.json."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, 0, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="ssd_traffic_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-latency", "--no-traffic",
             "--sustained-seconds", "0", "--no-other-precision", "--no-shufflenet", "--config", config]
    env = dict(os.environ, TMPDIR="/tmp")
    means = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child
            # (its own process group: a pass that outlives its limit is killed together with the program it profiles)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _, err = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.communicate()
                return None, 0, "the %s pass did not finish within %d s" % (counter, timeout_s)
            r = subprocess.CompletedProcess(cmd, pr.returncode, None, err)
            vals = []
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if DOMINANT_F32_KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, 0, "no %s rows for the dominant kernel (rocprofv3 rc %d: %s)" % (counter, r.returncode, r.stderr.decode("utf-8", "replace")[-200:])
            means[counter] = (sum(vals) / len(vals), len(vals))
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return None, 0, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    n = min(means["FETCH_SIZE"][1], means["WRITE_SIZE"][1])
    return (2.0 * means["FETCH_SIZE"][0] + means["WRITE_SIZE"][0]) * 1024.0, n, \
        "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate, --kernel-trace only) of 3 steps of this " \
        "workload in a child process, FETCH x2 per the gfx950 correction, KB -> bytes, mean over %d launches" % n


def roofline_block(prof, precision, steps, traffic_key=None):
    """roofline of the dominant kernel class: algorithmic FLOP of its launches / the union of their HIP-event
    intervals on the forward's own streams inside the timed region (ssd_profile_read)."""
    name, c = dominant(prof)
    n = max(c["launches"], 1)
    avg_ms = c["ms"] / n
    flop = c["flops"] / n
    ach = flop / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    peak = PEAK_FP32_MFMA_TFLOPS if precision == "f32" else PEAK_F16X3_TFLOPS
    traffic, src, t_alg, t_kernel = None, None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")    # scripts/collect_profiles.sh: separate --pmc passes
    if os.path.exists(tpath):
        t = json.load(open(tpath)).get(traffic_key or precision, {})
        t_alg, t_kernel = t.get("algorithmic_bytes_per_launch"), t.get("kernel")
        traffic, src = t.get("hbm_bytes_per_launch"), "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of " \
            "this command, FETCH x2 per the gfx950 correction; a committed measurement, not re-measured in this run): " + str(t.get("source", ""))
    return {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "traffic": traffic, "traffic_source": src,
            # `traffic` is the counter bytes of ONE kernel instance (the class's dominant one); its own algorithmic bytes beside it
            # (algorithmic_gbyte_per_launch below is the average over ALL launches of the class: not the same denominator)
            "traffic_kernel": t_kernel, "traffic_algorithmic_bytes": t_alg, "kernel": KERNEL_NAMES[name],
            "peak_note": ("dense exact-fp32 MFMA peak (v_mfma_f32_32x32x2_f32)" if precision == "f32"
                          else "dense F16 MFMA peak 2516.8 TFLOP/s / 3 MFMA terms per product"),
            "launches_per_step": c["launches"] / steps, "avg_launch_ms": avg_ms,
            "avg_launch_ms_note": "union of the class's launch intervals / launches (the two head towers run "
                                  "side by side on two streams)",
            "algorithmic_gflop_per_launch": flop / 1e9,
            "algorithmic_gbyte_per_launch": c["bytes"] / n / 1e9}


def kernel_tables(prof, steps):
    ms = {k: v["ms"] / steps for k, v in prof.items()}
    # per kernel class: algorithmic TFLOP/s and TB/s over the union of its launches' intervals
    # (MFMA peak 157.3 TFLOP/s; HBM 8.0 TB/s spec, 6.3 measured copy)
    rates = {k: {"tflops": v["flops"] / max(v["ms"], 1e-9) / 1e9, "tbytes_per_s": v["bytes"] / max(v["ms"], 1e-9) / 1e9,
                 "frac_of_hbm_8tbs": v["bytes"] / max(v["ms"], 1e-9) / 1e9 / HBM_PEAK_TBS}
             for k, v in prof.items() if v["launches"] > 0}
    return ms, rates


class Timed:
    """W warm-up steps, then exactly K steps between two fences (device sync + barrier)."""

    def __init__(self, world, dist, sync, dev, use_dist=None):
        self.world, self.dist, self.sync, self.dev = world, dist, sync, dev
        self.use_dist = world > 1 if use_dist is None else use_dist

    def fence(self):
        self.sync()
        if self.use_dist:
            self.dist.barrier()
            self.sync()

    def run(self, engine, step, steps, warmup):
        out = None
        for _ in range(warmup):
            out = step()
        self.fence()
        engine.profile_reset()
        engine.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        self.fence()
        dt = time.perf_counter() - t0
        engine.profile_enable(False)
        prof = engine.profile_read()
        self.per_rank_s = [dt]
        if self.use_dist:       # MAX over ranks (the contract's time); every rank's own time kept beside it
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            allt = torch.empty((self.world,), dtype=torch.float64, device=self.dev)
            self.dist.all_gather_into_tensor(allt, t)
            self.per_rank_s = [float(v) for v in allt.cpu()]
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out, prof

    def breakdown(self, step_with_mark, steps, stub):
        """A few extra steps (not the timed ones) with a mark between the engine's forward and the all-gather: per rank
        the time from a step's start to the mark (compute) and from the mark to the step's end (the collective, incl. its
        wait for the slowest rank), ms per step; lists over ranks."""
        self.fence()
        marks = []
        if stub:
            now = time.perf_counter
            for _ in range(steps):
                a = now()
                m = []
                step_with_mark(lambda: m.append(now()))
                marks.append((a, m[0], now()))
            comp = sum(b - a for a, b, c in marks) / steps * 1e3
            coll = sum(c - b for a, b, c in marks) / steps * 1e3
        else:
            for _ in range(steps):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                step_with_mark(e[1].record)
                e[2].record()
                marks.append(e)
            self.sync()
            comp = sum(e[0].elapsed_time(e[1]) for e in marks) / steps
            coll = sum(e[1].elapsed_time(e[2]) for e in marks) / steps
        self.fence()
        comp_l, coll_l = [comp], [coll]
        if self.use_dist:
            t = torch.tensor([comp, coll], dtype=torch.float64, device=self.dev)
            allt = torch.empty((self.world * 2,), dtype=torch.float64, device=self.dev)
            self.dist.all_gather_into_tensor(allt, t)
            comp_l = [float(v) for v in allt[0::2].cpu()]
            coll_l = [float(v) for v in allt[1::2].cpu()]
        return comp_l, coll_l


def _sysfs_probe(device):
    """(read_mhz, read_watts) callables on the HIP device's sysfs files (shader clock level marked '*' in pp_dpm_sclk, socket
    power in W), or None where the file is not readable by this user."""
    import glob
    import re
    try:
        pr = torch.cuda.get_device_properties(device)
        base = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None, None

    def first(pattern):
        for q in sorted(glob.glob(pattern)):
            try:
                open(q).read()
                return q
            except OSError:
                pass
        return None
    sclk = first(base + "/pp_dpm_sclk")
    power = first(base + "/hwmon/hwmon*/power1_average") or first(base + "/hwmon/hwmon*/power1_input")

    def mhz():
        m = re.search(r"(\d+)Mhz \*", open(sclk).read())
        return int(m.group(1)) if m else None

    def watts():
        return int(open(power).read()) / 1e6
    return (mhz if sclk else None), (watts if power else None)


def sustained_leg(step, images_per_step, device, seconds, window=5.0):
    """Back-to-back steps for `seconds` of wall clock (the headline's 20 steps are 0.8 s of GPU work on a 1.2 kW part with
    DVFS give-back): img/s over the first and the last `window` seconds from per-step events on the compute stream, shader
    clock / socket power (sysfs, 10 samples per second) over the same two windows."""
    import threading
    read_mhz, read_w = _sysfs_probe(device)
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            t = time.perf_counter()
            try:
                samples.append((t, read_mhz() if read_mhz else None, read_w() if read_w else None))
            except (OSError, ValueError):
                pass
            time.sleep(0.1)
    torch.cuda.synchronize(device)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    evs = [torch.cuda.Event(enable_timing=True)]
    evs[0].record()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        step()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        if len(evs) > 8:
            evs[-8].synchronize()            # bounded queue: the host stays at most 8 steps ahead
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    stop[0] = True
    th.join()
    ts = [0.0] + [evs[0].elapsed_time(e) * 1e-3 for e in evs[1:]]          # seconds since the first event, per finished step
    total_s, n = ts[-1], len(ts) - 1

    def rate(lo, hi):
        k = [i for i in range(1, len(ts)) if lo < ts[i] <= hi]
        if len(k) < 2:
            return None
        return images_per_step * (k[-1] - k[0]) / (ts[k[-1]] - ts[k[0]])

    def med(vals):
        v = sorted(x for x in vals if x is not None)
        return v[len(v) // 2] if v else None

    def sysfs(lo, hi):
        w = [sm for sm in samples if lo <= sm[0] - t0 <= hi]
        return {"sclk_mhz_median": med(x[1] for x in w), "power_w_median": med(x[2] for x in w), "samples": len(w)}
    first, last = rate(0.0, window), rate(total_s - window, total_s)
    return {"seconds": total_s, "steps": n, "img_s": images_per_step * n / total_s,
            "img_s_first_%ds" % window: first, "img_s_last_%ds" % window: last,
            "ratio_last_to_first": (last / first) if first and last else None,
            "sysfs_first_%ds" % window: sysfs(0.0, window), "sysfs_last_%ds" % window: sysfs(t1 - t0 - window, t1 - t0),
            "note": "same step as `value`, enqueued back to back (host at most 8 steps ahead), no profiling; `value` stays the "
                    "K-step number of the contract"}


def shufflenet_leg(local, timed, steps, warmup, batch):
    """BASELINE config 4: ShuffleNet-v2 + FPN (config_shufflenet.json), 640x640, batch 64, one GPU."""
    Wt = ssd_amd.synthetic_weights(PARAMS_SHUFFLE, seed=0, logits_bias=LOGITS_BIAS["shufflenet"])
    eng = ssd_amd.Engine(PARAMS_SHUFFLE, Wt, device=local, precision="f32")
    g = torch.Generator().manual_seed(4321)
    frames = torch.randint(0, 256, (batch, 640, 640, 3), dtype=torch.uint8, generator=g).to("cuda:%d" % local)
    res = {"workload": "ShuffleNet-v2 1.0x + FPN + RetinaNet heads + decode + per-class NMS, 640x640 uint8 frames, "
                       "batch %d (BASELINE config 4)" % batch, "batch": batch}
    outs = {}
    for mode in ("f32", "f16x3"):
        eng.set_precision(mode)
        dt, out, prof = timed.run(eng, lambda: eng.forward(frames), steps, warmup)
        outs[mode] = [t.clone() for t in out]
        ms, rates = kernel_tables(prof, steps)
        leg = {"value": batch * steps / dt, "unit": "img/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
               "roofline": roofline_block(prof, mode, steps, "shufflenet_" + mode), "kernel_ms_per_step": ms, "kernel_rates": rates}
        if mode == "f32":
            leg["whole_net_roofline_frac"] = ROOFLINE_MS["shufflenet"] * batch / (dt / steps * 1e3)
            leg["detections_per_image"] = float(out[3].float().mean().item())
            res.update(leg, dtype="f32", precision="f32")
        else:
            leg["status_word"] = eng.status()
            a, b = outs["f32"], outs["f16x3"]
            leg["agreement_with_f32"] = {
                "num_boxes_identical": bool((a[3] == b[3]).all().item()), "labels_identical": bool((a[1] == b[1]).all().item()),
                "max_abs_score_diff": float((a[2] - b[2]).abs().max().item()),
                "slots_with_box_diff_over_1e-4": int(((a[0] - b[0]).abs().amax(dim=2) > 1e-4).sum().item())}
            res["other_precision"] = dict(leg, precision="f16x3")
    eng.close()
    return res


def main(argv=None, engine_factory=None, backend="nccl", script=None):
    """engine_factory / backend / script are injection points for tests/ (a world-size-2 gloo run of this very
    launcher and step loop with a stand-in engine on CPU); the product run uses none of them."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-numa-bind", action="store_true", help="do not restrict the process to the CPUs of the GPU's NUMA node")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 32 = BASELINE config 5: 256/8; 64 with --config shufflenet)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="total images per step over all GPUs (default --batch x --gpus); a count the GPUs do not divide "
                         "gives uneven contiguous shards")
    ap.add_argument("--precision", choices=["f32", "f16x3"], default="f32",
                    help="arithmetic of the dense convolutions (include/ssd_hip.h SSD_PRECISION_*).  f32 (default) = the "
                         "reference's arithmetic, exact-fp32 MFMA, bit-identical to the oracle.  f16x3 = opt-in: fp32 operands "
                         "carried as split-fp16 pairs, 3 f16 MFMAs per product, fp32 accumulation -- NOT the reference's "
                         "arithmetic (it does not guarantee identical box indices).  The line reports the other mode beside "
                         "`value` (other_precision)")
    ap.add_argument("--config", choices=["mobilenet", "shufflenet"], default="mobilenet",
                    help="mobilenet (default) = BASELINE.json's metric: config_mobilenet.json at 640x896.  shufflenet = BASELINE config 4 "
                         "(config_shufflenet.json, 640x640, default --batch 64) as the line's workload: for profiling that network "
                         "alone; the default run already carries it as the shufflenet_config4 object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not re-measure roofline.traffic with two rocprofv3 --pmc child runs at the end (N = 1, mode f32; ~40 s); the "
                         "line then carries the committed measurement of profiles/traffic.json")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-shufflenet", action="store_true", help="skip the config-4 object (N = 1 only anyway)")
    ap.add_argument("--no-other-precision", action="store_true")
    ap.add_argument("--sustained-seconds", type=float, default=30.0,
                    help="length of the sustained-rate leg (N = 1 only; 0 skips it): back-to-back steps, img/s of the first and last 5 s")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="a kernel / schedule selector of the library for this run (ssd_set_option, include/ssd_hip.h; A/B runs: "
                         "scripts/ab_opt.sh); none changes a result bit in mode f32")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the collective path (init_process_group on RCCL, the ranks_seen gather, the all-gather of the detection "
                         "records, the all-reduces and the final barrier) even at world size 1 -- what a launch under "
                         "`torch.distributed.run --nproc-per-node 1` does too (WORLD_SIZE=1 in the environment)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.global_batch and args.global_batch < args.gpus:
        ap.error("--global-batch (%d) is smaller than --gpus (%d): some rank would get an empty shard" % (args.global_batch, args.gpus))
    argv = list(sys.argv[1:] if argv is None else argv)
    net = args.config
    params = PARAMS if net == "mobilenet" else PARAMS_SHUFFLE
    Hh, Ww = (H, W) if net == "mobilenet" else (640, 640)
    if args.batch is None:
        args.batch = 64 if net == "shufflenet" else 32
    if net == "shufflenet":
        args.no_latency = args.no_shufflenet = args.no_cpu_baseline = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nothing has touched the GPU yet (importing torch / ssd_amd does not)
        sys.exit(self_launch(args.gpus, argv, script or os.path.abspath(__file__)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    stub = engine_factory is not None
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    import torch.distributed as dist
    numa_node = None
    if stub:
        dev, sync = torch.device("cpu"), (lambda: None)
    else:
        torch.cuda.set_device(local)
        dev, sync = torch.device("cuda", local), torch.cuda.synchronize
        # one process per GPU, on the GPU's NUMA node (the host side of a batch-1 call is 1-3 % faster there and no longer
        # depends on where the scheduler put the thread: scripts/numa_probe.py)
        numa_node = None if args.no_numa_bind else ssd_amd.bind_to_gpu_numa_node(local)
    for kv in args.option:
        k, v = kv.split("=", 1)
        ssd_amd.set_option(k, int(v, 0))
    B = args.batch
    total = args.global_batch or B * world
    Wt = ssd_amd.synthetic_weights(params, seed=0, logits_bias=LOGITS_BIAS[net])
    detector = None
    if stub:
        engine = engine_factory(params, Wt, local)
    else:       # the boundary class (inference/detector.py:5-60) owns the engine; the throughput legs drive its engine directly
        detector = ssd_amd.Detector(Wt, visible_device_list=str(local), config=params, precision=args.precision)
        engine = detector.engine
        # The engine creates its internal streams with its first forward; HIP maps a new stream onto the least-loaded of a few
        # hardware queues.  One small forward BEFORE the process group exists gives the engine queues of its own; RCCL's
        # streams, created later, share whichever (they only run between forwards).  The other order measured 788 instead of
        # 822 img/s: the class tower's stream on the caller's queue (csrc/plan.hip ssd_side_stream).
        if world > 1 or args.force_dist or "WORLD_SIZE" in os.environ:
            engine.forward(torch.zeros((1, Hh, Ww, 3), dtype=torch.uint8, device=dev))
            sync()
    first_forward_before_pg = bool(not stub and (world > 1 or args.force_dist or "WORLD_SIZE" in os.environ))
    ranks_seen = [0]
    # inside a torch.distributed.run launch (WORLD_SIZE set, also = 1) or with --force-dist the collective path runs
    use_dist = world > 1 or args.force_dist or "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())
        if stub:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
        ids = torch.empty((world,), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int32, device=dev))
        ranks_seen = [int(v) for v in ids.cpu()]

    # this rank's shard of the global batch, resident in HBM before the timed region
    lo, hi = ssd_amd.shard_range(total, rank, world)
    g = torch.Generator().manual_seed(1234 + rank)
    frames = torch.randint(0, 256, (hi - lo, Hh, Ww, 3), dtype=torch.uint8, generator=g).to(dev)
    timed = Timed(world, dist, sync, dev, use_dist)

    def step():
        return ssd_amd.detect_sharded(engine, frames, total=total, force=use_dist)

    dt, out, prof = timed.run(engine, step, args.steps, args.warmup)
    assert out[0].shape[0] == total, (out[0].shape, total)
    det_per_image = float(out[3].float().mean().item())
    per_rank_ms = [t / args.steps * 1e3 for t in timed.per_rank_s]
    # where a step goes, per rank: compute up to the engine's last kernel, then the all-gather (with its wait for the slowest rank)
    out = [t.clone() for t in out]
    comp_ms, coll_ms = timed.breakdown(lambda mark: ssd_amd.detect_sharded(engine, frames, total=total, force=use_dist, on_forward_done=mark),
                                       min(5, max(2, args.steps)), stub)
    def status_all_ranks():
        """bit 0: an f16x3 activation left the fp16 range on SOME rank since the last call (0 in mode f32)."""
        v = engine.status()
        if use_dist:
            t = torch.tensor([v], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            v = int(t.item())
        return v

    status_value = status_all_ranks()
    sustained = None
    if world == 1 and not use_dist and not stub and args.sustained_seconds > 0:
        sustained = sustained_leg(step, total, dev, args.sustained_seconds)

    # the same workload in the other precision mode, same process, same frames (shorter run)
    other = "f32" if args.precision == "f16x3" else "f16x3"
    other_res = None
    if not args.no_other_precision and not stub:
        engine.set_precision(other)
        n_other = max(3, args.steps // 2)
        out = [t.clone() for t in out]       # (collective path: detect_sharded's results are views of its receive buffers)
        dt_o, out_o, prof_o = timed.run(engine, step, n_other, 2)
        status_other = status_all_ranks()
        # agreement of the two modes on this run's frames (mode f32 is bit-identical to the CPU oracle, tests/)
        agree = {"num_boxes_identical": bool((out_o[3] == out[3]).all().item()),
                 "labels_identical": bool((out_o[1] == out[1]).all().item()),
                 "max_abs_score_diff": float((out_o[2] - out[2]).abs().max().item()),
                 "slots_with_box_diff_over_1e-4": int(((out_o[0] - out[0]).abs().amax(dim=2) > 1e-4).sum().item()),
                 "detections": int(out[3].sum().item())}
        ms_o, rates_o = kernel_tables(prof_o, n_other)
        other_res = {"precision": other, "value": total * n_other / dt_o, "unit": "img/s",
                     "ms_per_step": dt_o / n_other * 1e3, "steps": n_other,
                     "dtype": "f32" if other == "f32" else "f32 carried as split f16 pairs (3 x f16 MFMA, f32 accumulate)",
                     "note": None if other == "f32" else "opt-in mode, narrower than the reference's arithmetic: labels / num_boxes "
                             "are not guaranteed identical to the oracle's (see agreement_with_value_run); never the headline",
                     "status_word": status_other,
                     "roofline": roofline_block(prof_o, other, n_other), "kernel_ms_per_step": ms_o, "kernel_rates": rates_o,
                     "agreement_with_value_run": agree}
        engine.set_precision(args.precision)

    pcie_img_s = None
    if not stub:
        # the same step with the boundary's host buffers in the loop (pinned host frames -> HBM,
        # detections -> host); reported beside `value`, never as `value`
        host_frames = frames.cpu().numpy()
        sync()
        # A warm-up pass first: the first two batches of a batch shape create detect_stream's two buffer sets (pinning 2 x 32 MB of
        # host memory: ~0.15 s each, during which finished results wait to be collected) and, after a precision change, build the
        # plan.  Then the rate between INTERIOR results of a second pass: its first batch fills the pipeline, its last result is
        # delivered with nothing behind it.  (Rounds 1-5 timed "everything after the first result" of ONE pass; when the first
        # result is collected late -- the host busy pinning memory -- that window holds fewer batches than it counts: round 6's
        # full run read 1.09 x the resident rate.  scripts/experiments/bench_pcie_debug.py has the time stamps.)
        n_p = max(6, args.steps)
        list(detector.detect_stream(host_frames for _ in range(3)))
        sync()
        stamps = []
        for o in detector.detect_stream(host_frames for _ in range(n_p + 3)):
            stamps.append(time.perf_counter())
        pcie_img_s = (hi - lo) * n_p / (stamps[n_p + 1] - stamps[1])
        pcie_ok = all(np.array_equal(a, b.cpu().numpy()) for a, b in zip(o, [t[lo:hi] for t in out])) if args.precision == "f32" else None

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        res = {
            "metric": ("images/sec at 896x640, MobileNet-v1 RetinaNet (whole hot path incl. decode + per-class NMS)" if net == "mobilenet"
                       else "images/sec at 640x640, ShuffleNet-v2 RetinaNet (BASELINE config 4; NOT BASELINE.json's headline metric)"),
            "value": total * args.steps / dt, "unit": "img/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f32 carried as split f16 pairs (3 x f16 MFMA, f32 accumulate)",
            "precision": args.precision, "status_word": status_value,
            "data": "synthetic" if not stub else "STAND-IN ENGINE (launcher test, no GPU work)",
            "ranks_seen": ranks_seen, "collective_path": bool(use_dist), "numa_node_bound": numa_node,
            # what a scaling loss would be made of (SCALE_rNN.json is the driver's; this line must explain it): every rank's own
            # time for the K timed steps (ms_per_step is their MAX), and from a few extra steps with an event between the engine's
            # forward and the collective: compute and all-gather per rank.  A rank whose compute is short waits in ITS all-gather
            # for the slowest one: rank skew shows as a spread in per_rank_compute_ms_per_step, the collective's own cost as
            # min(per_rank_allgather_ms_per_step).
            "per_rank_ms_per_step": per_rank_ms,
            "per_rank_compute_ms_per_step": comp_ms, "per_rank_allgather_ms_per_step": coll_ms,
            "compute_ms_per_step": max(comp_ms), "allgather_ms_per_step": max(coll_ms),
            "allgather_ms_per_step_min_over_ranks": min(coll_ms),
            "allgather_bytes_per_rank": (hi - lo) * (6 * params["num_classes"] * params["max_boxes_per_class"] + 1) * 4,
            "engine_first_forward_before_process_group": first_forward_before_pg,
            "hardware_queue_note": "the engine's internal streams are created by its first forward; run before RCCL creates its streams "
                                   "they get hardware queues of their own (788 vs 822 img/s the other way round, csrc/plan.hip)",
            "config": {"workload": ("MobileNet-v1 + FPN + RetinaNet heads + decode + per-class NMS, 640x896 (HxW) "
                                    "uint8 frames, %d per GPU (BASELINE config 5 shard; config 2 = same graph at batch 1, "
                                    "see latency_batch1; config 4 = the shufflenet_config4 object)" % B) if net == "mobilenet" else
                                   ("ShuffleNet-v2 1.0x + FPN + RetinaNet heads + decode + per-class NMS, 640x640 uint8 frames, "
                                    "%d per GPU (BASELINE config 4)" % B),
                       "per_gpu_batch": B, "global_batch": total, "height": Hh, "width": Ww,
                       "shards": [list(ssd_amd.shard_range(total, r, world)) for r in range(world)],
                       "parallelism": "dp%d" % world,
                       "weights": "random-init (seed 0), logits bias %.1f" % LOGITS_BIAS[net],
                       "library_options": args.option,
                       "detections_per_image": det_per_image},
        }
        if not stub:
            ms, rates = kernel_tables(prof, args.steps)
            res["roofline"] = roofline_block(prof, args.precision, args.steps, None if net == "mobilenet" else "shufflenet_" + args.precision)
            res["kernel_ms_per_step"], res["kernel_rates"] = ms, rates
            res["pcie_inclusive_img_s_per_gpu"] = pcie_img_s
            res["pcie_inclusive"] = {"img_s_per_gpu": pcie_img_s, "ratio_to_resident": pcie_img_s * world / (total * args.steps / dt),
                                     "outputs_equal_resident_run": pcie_ok,
                                     "path": "Detector.detect_stream: host uint8 batches -> pinned staging -> H2D on a copy stream, "
                                             "forward, packed D2H on a second copy stream, numpy out; copies of batches k+1 / k-1 under "
                                             "the compute of batch k; rate between interior results of a warmed-up pass"}
            # 1.113 ms/img at the per-layer roofline of the exact-fp32 arithmetic
            res["whole_net_roofline_frac"] = ROOFLINE_MS[net] * (hi - lo) / ms_step if args.precision == "f32" else None
            if sustained:
                res["sustained"] = sustained
            if other_res:
                res["other_precision"] = other_res
            if not args.no_latency:
                lat = {}
                for mode in ("f32", "f16x3"):
                    engine.set_precision(mode)
                    lat[mode] = latency_batch1(detector)
                engine.set_precision(args.precision)
                res["latency_batch1"] = dict(lat[args.precision], precision=args.precision,
                                             roofline_ms=ROOFLINE_MS["mobilenet"], by_precision=lat,
                                             segments_p50_us=latency_segments(detector))
                if args.precision == "f32":
                    res["latency_mixed_sizes"] = latency_mixed_sizes(detector)
                    res["throughput_mixed_sizes"] = throughput_mixed_sizes(detector)
            if world == 1 and not args.no_shufflenet:
                engine.close()
                res["shufflenet_config4"] = shufflenet_leg(local, timed, max(3, args.steps // 2), 2, 64)
            if not args.no_cpu_baseline:          # (rank 0 at any world size: ~12 s of host time while the other ranks wait at the barrier)
                res["cpu_baseline"] = cpu_baseline()
            if world == 1 and not use_dist and args.precision == "f32" and not args.no_traffic:
                # LAST: every timed leg is done; the children profile the same workload while this process sits idle
                try:
                    engine.close()
                except Exception:
                    pass
                t_bytes, t_n, t_note = measure_traffic_live(net)
                rl = res["roofline"]
                if t_bytes is not None:
                    rl["traffic_committed"] = rl["traffic"]
                    rl["traffic"], rl["traffic_source"], rl["traffic_launches_averaged"] = t_bytes, t_note, t_n
                else:
                    rl["traffic_live_error"] = t_note
        print(json.dumps(res))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

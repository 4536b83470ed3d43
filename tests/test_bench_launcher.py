"""`python bench.py --gpus N` must work as typed: the launcher starts torch.distributed.run as a child
process before anything touches the GPU.  Here the same launcher + step loop + all-gather run on CPU
(gloo, world size 2) with a stand-in engine; even and uneven shards."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "helpers", "bench_stub_main.py")


def run(args):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, STUB] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout           # ONE JSON line, from rank 0
    return json.loads(lines[0])


def check_scaling_keys(res, world):
    """The N > 1 line explains itself: every rank's own step time, compute and all-gather per rank from the
    marked extra steps, the hardware-queue note."""
    for k in ("per_rank_ms_per_step", "per_rank_compute_ms_per_step", "per_rank_allgather_ms_per_step"):
        assert isinstance(res[k], list) and len(res[k]) == world and all(v >= 0 for v in res[k]), k
    assert abs(max(res["per_rank_ms_per_step"]) - res["ms_per_step"]) < 1e-6 * max(1.0, res["ms_per_step"])
    assert res["compute_ms_per_step"] == max(res["per_rank_compute_ms_per_step"])
    assert res["allgather_ms_per_step"] == max(res["per_rank_allgather_ms_per_step"])
    assert res["allgather_ms_per_step_min_over_ranks"] == min(res["per_rank_allgather_ms_per_step"])
    assert res["allgather_bytes_per_rank"] > 0 and "engine_first_forward_before_process_group" in res and "hardware_queue_note" in res


@pytest.mark.parametrize("extra,total,shards", [([], 6, [[0, 3], [3, 6]]), (["--global-batch", "5"], 5, [[0, 3], [3, 5]])])
def test_self_launch_world2_gloo(extra, total, shards):
    res = run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3"] + extra)
    assert res["n_gpus"] == 2 and res["ranks_seen"] == [0, 1]
    assert res["config"]["global_batch"] == total and res["config"]["shards"] == shards
    assert res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["value"] > 0 and abs(res["value"] - total * 2 / (res["ms_per_step"] * 2e-3)) < 1e-6 * res["value"]
    assert res["precision"] == "f32" and res["dtype"] == "f32"          # the reference's arithmetic is the headline
    check_scaling_keys(res, 2)


def test_self_launch_world8_gloo_uneven_global_batch():
    """The driver's 8-GPU run (BASELINE config 5) is the first time eight ranks meet on hardware; the launcher, the shard
    table and the padded all-gather meet them here first: world size 8 on gloo, a global batch of 250 that 8 does not
    divide (shards of 32, 32, 31 x 6)."""
    res = run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--global-batch", "250"])
    assert res["n_gpus"] == 8 and res["ranks_seen"] == list(range(8))
    exp, lo = [], 0
    for r in range(8):
        n = 32 if r < 2 else 31
        exp.append([lo, lo + n])
        lo += n
    assert res["config"]["global_batch"] == 250 and res["config"]["shards"] == exp and lo == 250
    assert res["value"] > 0 and abs(res["value"] - 250 * 2 / (res["ms_per_step"] * 2e-3)) < 1e-6 * res["value"]
    check_scaling_keys(res, 8)
    assert res["allgather_bytes_per_rank"] == 32 * 48004          # rank 0's shard of the 250 images
    even = run(["--gpus", "8", "--steps", "1", "--warmup", "0", "--batch", "4"])
    assert even["config"]["global_batch"] == 32 and even["config"]["shards"][7] == [28, 32] and even["ranks_seen"] == list(range(8))


def test_single_rank_needs_no_launcher():
    res = run(["--steps", "1", "--warmup", "0", "--batch", "2"])
    assert res["n_gpus"] == 1 and res["ranks_seen"] == [0] and res["config"]["global_batch"] == 2


def test_force_dist_runs_the_collective_path_at_world_size_1():
    """--force-dist (and WORLD_SIZE=1 from `torch.distributed.run --nproc-per-node 1`): process group, ranks_seen gather,
    the all-gather of the detection records, the all-reduces and the final barrier all run with ONE rank."""
    res = run(["--steps", "2", "--warmup", "1", "--batch", "3", "--force-dist"])
    assert res["n_gpus"] == 1 and res["ranks_seen"] == [0] and res["collective_path"] is True
    assert res["config"]["global_batch"] == 3 and res["config"]["shards"] == [[0, 3]]
    check_scaling_keys(res, 1)
    plain = run(["--steps", "2", "--warmup", "1", "--batch", "3"])
    assert plain["collective_path"] is False


def test_arguments_that_would_leave_a_rank_without_images_are_refused():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, STUB, "--gpus", "2", "--global-batch", "1"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "empty shard" in r.stderr


def test_bench_refuses_without_gpu():
    # the product bench has no CPU path: without a HIP device it must stop, not fall back
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)

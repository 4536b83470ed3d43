"""N > 1 path on CPU: world_size-2 gloo all-gather of detection records."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records(rank, B, T):
    g = torch.Generator().manual_seed(100 + rank)
    boxes = torch.rand((B, T, 4), generator=g)
    scores = torch.rand((B, T), generator=g)
    labels = torch.randint(0, 80, (B, T), generator=g, dtype=torch.int32)
    num = torch.randint(0, T, (B,), generator=g, dtype=torch.int32)
    return boxes, labels, scores, num


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import ssd_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, T = 3, 50
    mine = _records(rank, B, T)
    out = ssd_amd.all_gather_detections(*mine)
    ok = True
    for r in range(world):
        exp = _records(r, B, T)
        for a, b in zip(out, exp):
            ok &= bool(torch.equal(a[r * B:(r + 1) * B], b))
    ok &= out[0].shape == (world * B, T, 4) and out[3].dtype == torch.int32
    # uneven shards (5 images over 2 ranks = 3 + 2): padded to the largest shard for the one all-gather
    lo5, hi5 = ssd_amd.shard_range(5, rank, world)
    mine5 = tuple(t[:hi5 - lo5] for t in _records(rank, B, T))
    out5 = ssd_amd.all_gather_detections(*mine5, total=5)
    ok &= out5[0].shape == (5, T, 4)
    row = 0
    for r in range(world):
        l5, h5 = ssd_amd.shard_range(5, r, world)
        for a, b in zip(out5, _records(r, B, T)):
            ok &= bool(torch.equal(a[row:row + h5 - l5], b[:h5 - l5]))
        row += h5 - l5
    lo, hi = ssd_amd.shard_range(7, rank, world)
    q.put((rank, ok, lo, hi))
    dist.destroy_process_group()


def test_all_gather_detections_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert (res[0][2], res[0][3]) == (0, 4) and (res[1][2], res[1][3]) == (4, 7)


class _RecordEngine:
    """Stand-in with the engine's record interface (Engine.forward(images, records=...), record_words): writes the records
    _records() defines for this rank straight into the block it is handed -- what ssd_forward_records does on the GPU."""

    def __init__(self, rank, T):
        self.rank, self.T, self.record_words = rank, T, 6 * T + 1

    def forward(self, images, out=None, records=None):
        import ssd_amd
        from importlib import import_module
        d = import_module("ssd_amd.distributed")
        mine = tuple(t[:images.shape[0]] for t in _records(self.rank, 3, self.T))
        if records is None:
            return mine
        records.copy_(d.pack_detections(*mine))
        return d.unpack_detections(records)


def _worker_sharded(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import ssd_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = 40
    eng = _RecordEngine(rank, T)
    ok = True
    # even shards: 3 + 3; the engine's records land in this rank's slice of the receive buffer, results are views of it
    out = ssd_amd.detect_sharded(eng, torch.zeros((3, 8, 8, 3), dtype=torch.uint8), total=6)
    for r in range(world):
        for a, b in zip(out, _records(r, 3, T)):
            ok &= bool(torch.equal(a[r * 3:(r + 1) * 3], b))
    # lifetime of the results (views of one of TWO alternating receive buffers): step k survives step k + 1 ...
    keep = [t.clone() for t in out]
    ptr = out[0].data_ptr()
    eng.rank = rank + 10                                # the next steps' records differ from step k's
    out_b = ssd_amd.detect_sharded(eng, torch.zeros((3, 8, 8, 3), dtype=torch.uint8), total=6)
    ok &= all(bool(torch.equal(a, b)) for a, b in zip(out, keep)) and out_b[0].data_ptr() != ptr
    ok &= not torch.equal(out_b[0], keep[0])
    # ... and is overwritten by step k + 2 (documented: clone what must live longer)
    out_c = ssd_amd.detect_sharded(eng, torch.zeros((3, 8, 8, 3), dtype=torch.uint8), total=6)
    ok &= out_c[0].data_ptr() == ptr and bool(torch.equal(out[0], out_c[0]))
    eng.rank = rank
    # uneven shards: 5 = 3 + 2, twice (the buffers are reused)
    for _ in range(2):
        lo, hi = ssd_amd.shard_range(5, rank, world)
        out5 = ssd_amd.detect_sharded(eng, torch.zeros((hi - lo, 8, 8, 3), dtype=torch.uint8), total=5)
        ok &= out5[0].shape == (5, T, 4) and out5[3].shape == (5,)
        row = 0
        for r in range(world):
            l5, h5 = ssd_amd.shard_range(5, r, world)
            for a, b in zip(out5, _records(r, 3, T)):
                ok &= bool(torch.equal(a[row:row + h5 - l5], b[:h5 - l5]))
            row += h5 - l5
    q.put((rank, ok))
    dist.destroy_process_group()


def test_detect_sharded_record_path_gloo_world2():
    """detect_sharded with an engine that writes records (the product path): records -> this rank's slice of the receive
    buffer -> ONE all-gather -> views; even and uneven shards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]


def test_pack_unpack_roundtrip_and_record_size():
    sys.path.insert(0, ROOT)
    import ssd_amd
    from importlib import import_module
    d = import_module("ssd_amd.distributed")
    b, l, s, n = _records(0, 4, 2000)
    rec = d.pack_detections(b, l, s, n)
    assert rec.shape == (4, 12001) and rec.element_size() * rec.shape[1] == 48004
    b2, l2, s2, n2 = d.unpack_detections(rec)
    assert torch.equal(b, b2) and torch.equal(l, l2) and torch.equal(s, s2) and torch.equal(n, n2)
    # world size 1 / uninitialised: identity
    out = ssd_amd.all_gather_detections(b, l, s, n)
    assert out[0] is b
    assert [ssd_amd.shard_range(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]

"""Multi-GPU: images shard over ranks (one process per GPU), detections are exchanged with
ONE all-gather of fixed-size records per batch (RCCL over xGMI with backend 'nccl'; the
same code runs on gloo/CPU tensors in the tests).  The reference has no multi-GPU code;
every image is independent end to end (nms.py:96-101 maps over images), so no other
collective exists on the path.

Record per image, 32-bit words: boxes [T,4] f32 | scores [T] f32 | labels [T] i32 |
num_boxes i32  = 6T+1 words (48 004 B at T = 2000).
"""
import torch
import torch.distributed as dist


def bind_to_gpu_numa_node(device=0):
    """One process per GPU, on the GPU's own NUMA node: restricts this process (and the threads it starts later) to the CPUs
    of the node the device hangs off (sysfs numa_node of its PCI function).  The host side of a small call -- staging memcpy
    into pinned memory, kernel launches, reading results the GPU wrote into pinned memory -- runs measurably faster there:
    batch-1 `Detector.__call__` p50 1.700 ms bound to the GPU's node, 1.743 bound to the other one, 1.71-1.75 unbound
    (scripts/numa_probe.py, profiles/r03_numa_probe.log).  Call it before creating the Detector (pinned buffers are placed by
    first touch).  Returns the node, or None when the topology is not readable (then nothing is changed)."""
    import glob
    import os
    try:
        pr = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = []
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.extend(range(int(a), int(b) + 1))
            elif part:
                cpus.append(int(part))
        cpus = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:            # no GPU, no sysfs, a torch build without the PCI fields: leave the process as it is
        return None


def shard_range(total, rank, world_size):
    """Contiguous split of `total` images: rank r gets [lo, hi)."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_detections(boxes, labels, scores, num):
    B, T = scores.shape
    rec = torch.empty((B, 6 * T + 1), dtype=torch.int32, device=scores.device)
    rec[:, :4 * T] = boxes.reshape(B, 4 * T).view(torch.int32)
    rec[:, 4 * T:5 * T] = scores.view(torch.int32)
    rec[:, 5 * T:6 * T] = labels
    rec[:, 6 * T] = num
    return rec


def unpack_detections(rec):
    """(boxes, labels, scores, num) as VIEWS of a record block [B, 6T+1]: no copy, no launch."""
    B, words = rec.shape
    T = (words - 1) // 6
    boxes = rec[:, :4 * T].view(torch.float32).unflatten(1, (T, 4))
    scores = rec[:, 4 * T:5 * T].view(torch.float32)
    labels = rec[:, 5 * T:6 * T]
    num = rec[:, 6 * T]
    return boxes, labels, scores, num


def gather_records(rec, group=None, out=None):
    """ONE all-gather of the [B_local, 6T+1] int32 records -> [world*B_local, 6T+1], rank order.  `out`: the receive
    buffer; when `rec` is this rank's slice of it the collective runs in place (RCCL: sendbuff == recvbuff + rank * count)."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    # One-buffer form everywhere (RCCL has it, and so does this image's gloo); only a backend that says it does not
    # implement it takes the list form.  Any other error is a real failure and propagates unchanged.
    try:
        dist.all_gather_into_tensor(out, rec, group=group)
    except NotImplementedError:
        parts = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(parts, rec, group=group)
        out = torch.cat(parts, 0)
    return out


def all_gather_detections(boxes, labels, scores, num, group=None, total=None, force=False):
    """Every rank contributes the records of its B_local images and receives all records in
    rank order.  `total` = the global image count when the shards are uneven
    (shard_range(total, rank, world)): the records are padded to the largest shard for the ONE
    all-gather (fixed-size operands) and the pad rows are dropped afterwards.
    A group of one rank needs no exchange and returns its inputs; `force` runs the pack -> all-gather ->
    unpack path even then (bench.py --force-dist: the collective path through RCCL on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return boxes, labels, scores, num
    if dist.get_world_size(group) == 1 and not force:
        return boxes, labels, scores, num
    world = dist.get_world_size(group)
    rec = pack_detections(boxes, labels, scores, num)
    if total is None or total % world == 0:
        return unpack_detections(gather_records(rec, group))
    per = -(-total // world)
    if rec.shape[0] > per:
        raise ValueError("shard of %d images exceeds ceil(%d / %d)" % (rec.shape[0], total, world))
    pad = torch.zeros((per, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    pad[:rec.shape[0]] = rec
    got = gather_records(pad, group).reshape(world, per, rec.shape[1])
    rows = [got[r, :shard_range(total, r, world)[1] - shard_range(total, r, world)[0]] for r in range(world)]
    return unpack_detections(torch.cat(rows, 0))


_gather_buffers = {}


def detect_sharded(engine, images_local, group=None, total=None, force=False, on_forward_done=None):
    """One data-parallel step: this rank's shard through the HIP path, then the all-gather.
    images_local: uint8 CUDA tensor [B_local,H,W,3]; `total`, `force`: see all_gather_detections.
    The engine writes its records straight into this rank's slice of the all-gather's receive buffer (ssd_forward_records),
    the collective runs in place on it and the four results are views of the buffer: no pack / unpack launches, no stream
    users beside RCCL's.  Uneven shards: every rank's slice is the largest shard's size; the rows beyond a rank's own
    shard are dropped by one concatenation of row ranges.

    Lifetime of the results: on the collective path they are VIEWS of a receive buffer this module keeps per (engine,
    world, shard size) -- TWO buffers used alternately, so the results of step k stay intact during step k + 1 (the usual
    "consume the previous step while the next one runs" loop) and are overwritten by step k + 2.  Clone what must live
    longer.  Without a process group (or a group of one rank and no `force`) the results are fresh tensors.

    A failed collective propagates: after an RCCL error the communicator is aborted and a retry on the same process group
    can hang every rank, so nothing is retried here.  Whether the send buffer may sit inside the receive buffer is decided
    once from the backend's name ('nccl' = RCCL: in place; anything else: one copy of this rank's slice).

    `on_forward_done`: called (no arguments) between the engine's forward and the collective -- bench.py records an event
    there to split a step into compute and all-gather."""
    use = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
    if not hasattr(engine, "record_words"):                # a stand-in engine (launcher tests): the generic path
        boxes, labels, scores, num = engine.forward(images_local)
        if on_forward_done is not None:
            on_forward_done()
        return all_gather_detections(boxes, labels, scores, num, group=group, total=total, force=force)
    if not use:
        out = engine.forward(images_local)
        if on_forward_done is not None:
            on_forward_done()
        return out
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = images_local.shape[0]
    per = n if (total is None or total % world == 0) else -(-total // world)
    if n > per:
        raise ValueError("shard of %d images exceeds ceil(%d / %d)" % (n, total, world))
    key = (id(engine), world, per, str(images_local.device))
    slot = _gather_buffers.get(key)
    if slot is None:
        _gather_buffers.clear()
        slot = _gather_buffers[key] = {"bufs": [torch.zeros((world, per, engine.record_words), dtype=torch.int32, device=images_local.device)
                                                for _ in range(2)], "turn": 0}
    buf = slot["bufs"][slot["turn"]]
    slot["turn"] ^= 1
    engine.forward(images_local, records=buf[rank, :n])
    if on_forward_done is not None:
        on_forward_done()
    inplace = dist.get_backend(group) == "nccl"
    got = gather_records(buf[rank] if inplace else buf[rank].clone(), group, out=buf.view(world * per, -1))
    if per == n and (total is None or total % world == 0):
        return unpack_detections(got)
    g3 = got.view(world, per, -1)
    rows = [g3[r, :shard_range(total, r, world)[1] - shard_range(total, r, world)[0]] for r in range(world)]
    return unpack_detections(torch.cat(rows, 0))

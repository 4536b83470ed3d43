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
    B, words = rec.shape
    T = (words - 1) // 6
    boxes = rec[:, :4 * T].contiguous().view(torch.float32).reshape(B, T, 4)
    scores = rec[:, 4 * T:5 * T].contiguous().view(torch.float32)
    labels = rec[:, 5 * T:6 * T].contiguous()
    num = rec[:, 6 * T].contiguous()
    return boxes, labels, scores, num


def gather_records(rec, group=None):
    """ONE all-gather of the [B_local, 6T+1] int32 records -> [world*B_local, 6T+1], rank order."""
    world = dist.get_world_size(group)
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    try:
        dist.all_gather_into_tensor(out, rec, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(parts, rec, group=group)
        out = torch.cat(parts, 0)
    return out


def all_gather_detections(boxes, labels, scores, num, group=None):
    """Every rank contributes the records of its B_local images (equal on all ranks) and
    receives all world_size*B_local records in rank order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return boxes, labels, scores, num
    return unpack_detections(gather_records(pack_detections(boxes, labels, scores, num), group))


def detect_sharded(engine, images_local, group=None):
    """One data-parallel step: this rank's shard through the HIP path, then the all-gather.
    images_local: uint8 CUDA tensor [B_local,H,W,3]."""
    boxes, labels, scores, num = engine.forward(images_local)
    return all_gather_detections(boxes, labels, scores, num, group=group)

"""Frame sharding across the GPUs of a node.

Frames are independent units: rank r of `world` codes frames
r, r + world, ... of the batch, or -- in the weak-scaling benchmark -- its own
fixed-size shard.  Nothing of the data path crosses ranks; the only collective is
the final reduction of per-rank metric sums (RCCL over xGMI on the GPU box, gloo
in the CPU tests): seconds by MAX, everything else by SUM.
"""
import torch
import torch.distributed as dist

# (the last two: seconds per step of bench.py's extra legs -- frames resident in HBM, one frame per call)
METRIC_FIELDS = ("pixels", "bits", "frames", "psnr_sum", "ssim_sum", "resident_s", "one_frame_s")


def shard(total_frames, rank, world):
    """indices of the frames rank `rank` codes"""
    return list(range(rank, total_frames, world))


def reduce_metrics(local, seconds, device="cpu"):
    """local: dict with METRIC_FIELDS (missing = 0).  Returns (totals dict, max seconds)
    identical on every rank."""
    vec = torch.tensor([float(local.get(k, 0.0)) for k in METRIC_FIELDS], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return dict(zip(METRIC_FIELDS, vec.tolist())), float(t.item())

"""Frame sharding across the GPUs of one node.  Frames are independent: rank r of W
decodes a contiguous block of frames with its own handle and stream; there is NO collective on the
data path.  Only the timing (max over ranks) and the error counters are reduced after the last
kernel, a few dozen bytes."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def block_range(total, rank, world):
    """contiguous ceil-sized blocks: [lo, hi) of `total` frames owned by `rank`"""
    per = -(-total // world)
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def frame_seed_offset(frames_per_rank, rank):
    """global index of a rank's first frame under weak scaling (every rank decodes frames_per_rank
    DIFFERENT frames: the channel RNG is keyed by the global frame index)"""
    return rank * frames_per_rank


def reduce_counters(local, world, dist=None, device=None):
    """sum of int64 counters + max of float seconds over ranks.
    local = (seconds, [int counters...]).  With world == 1 no process group is touched."""
    secs, counters = local
    if world == 1 or dist is None:
        return secs, list(counters)
    import torch
    t = torch.tensor([secs], dtype=torch.float64, device=device)
    c = torch.tensor(list(counters), dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [int(x) for x in c.tolist()]

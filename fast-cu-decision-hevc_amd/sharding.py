"""Chain sharding for multi-GPU runs: frames (x QPs) are independent chains, so every rank owns a
disjoint set and no collective touches the data path.  Weak scaling: a rank's share
does not depend on the world size."""


def chains_for_rank(frames_per_gpu, qps, rank):
    """[(frame_seed, qp)] of one rank; seeds are globally unique so ranks never duplicate content."""
    stride = max(1000, frames_per_gpu)                     # ranks own disjoint seed ranges whatever the frame count
    return [(7 + f + stride * rank, qp) for f in range(frames_per_gpu) for qp in qps]


def slices_for_rank(n_ctu, slice_ctus, world, rank):
    """One frame over several GPUs: contiguous runs of whole slices per rank, as
    [(first_ctu, n_ctus)] chain ranges for fcu_chain_set_range.  No rank shares a slice."""
    n_sl = (n_ctu + slice_ctus - 1) // slice_ctus
    per = (n_sl + world - 1) // world
    out = []
    for k in range(rank * per, min(n_sl, (rank + 1) * per)):
        first = k * slice_ctus
        out.append((first, min(slice_ctus, n_ctu - first)))
    return out


def reduce_step_time(dist, local_seconds, device=None):
    """Slowest rank defines the step time (MAX all-reduce); returns local_seconds when single-process."""
    if dist is None:
        return local_seconds
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

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


def slice_owner(n_ctu, slice_ctus, world):
    """owner rank of every slice of a picture under slices_for_rank (contiguous runs of whole slices)"""
    n_sl = (n_ctu + slice_ctus - 1) // slice_ctus
    per = (n_sl + world - 1) // world
    return [min(k // per, world - 1) for k in range(n_sl)]


CTU_HEAD_BYTES = 8192          # fcu_ctu_out up to and including mv / mvd: every per-partition decision array, no coefficients (include/fcu.h)


def _ctu_rects(a, b, w_ctu, width, height):
    """The luma rectangles (y0, y1, x0, x1) that CTUs [a, b) in raster order cover: a partial first CTU row, whole rows, a
    partial last row."""
    out = []
    while a < b:
        cy, cx = divmod(a, w_ctu)
        if cx == 0 and b - a >= w_ctu:
            rows = (b - a) // w_ctu
            out.append((cy * 64, min(height, (cy + rows) * 64), 0, width))
            a += rows * w_ctu
        else:
            n = min(b - a, w_ctu - cx)
            out.append((cy * 64, min(height, cy * 64 + 64), cx * 64, min(width, (cx + n) * 64)))
            a += n
    return out


def merge_picture(dist, planes, out, slice_ctus, ctu_out_bytes, rank=None):
    """The one exchange step of a picture whose slices were decided on different ranks (inter hand-off:
    "reconstructed reference pixels broadcast once per frame").  Every rank sends what it decided and nothing else: the
    reconstruction samples of its own CTUs (at most three rectangles per plane: a partial first CTU row, whole rows, a partial
    last row) and, per CTU, the head of its fcu_ctu_out record -- the decision and motion arrays the loop filters
    (TComLoopFilter.cpp:130) and the next picture's TMVP (TComDataCU.cpp:3175-3242) read, 8 KB of the record's 32.8 KB; the
    coefficient arrays stay on the rank that produced them.  One broadcast per sender and piece (RCCL / xGMI): 12.4 MB of
    samples + 16.7 MB of decisions per 4K picture cross the links once, against the 2 x 80 MB a SUM all-reduce of zero-padded
    planes and whole records moved.  Afterwards every rank runs the loop filters on the whole picture itself (cheap
    data-parallel kernels, replicated instead of a second exchange) and pads its own copy of the reference.
    Order per picture: decide own slices -> merge_picture -> deblock -> SAO -> pad -> next picture.
    `planes`: the three uint8 reconstruction planes (height x width, half size for chroma), `out`: the picture's fcu_ctu_out
    array as a uint8 tensor; both completed in place.  No-op without a process group."""
    if dist is None:
        return
    world = dist.get_world_size()
    rank = dist.get_rank() if rank is None else rank
    height, width = planes[0].shape
    w_ctu = (width + 63) // 64
    n_ctu = w_ctu * ((height + 63) // 64)
    v = out.view(n_ctu, ctu_out_bytes)
    for src in range(world):
        ranges = slices_for_rank(n_ctu, slice_ctus, world, src)
        if not ranges:
            continue
        a, b = ranges[0][0], ranges[-1][0] + ranges[-1][1]     # a rank's slices are contiguous
        head = v[a:b, :CTU_HEAD_BYTES].contiguous()
        dist.broadcast(head, src=src)
        if src != rank:
            v[a:b, :CTU_HEAD_BYTES] = head
        for (y0, y1, x0, x1) in _ctu_rects(a, b, w_ctu, width, height):
            for k, p in enumerate(planes):
                sh = 1 if k else 0
                piece = p[y0 >> sh:y1 >> sh, x0 >> sh:x1 >> sh].contiguous()
                dist.broadcast(piece, src=src)
                if src != rank:
                    p[y0 >> sh:y1 >> sh, x0 >> sh:x1 >> sh] = piece

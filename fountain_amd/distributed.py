"""Multi-GPU decomposition: film tiles are independent units of work (integrator/mod.rs:197-204, one sampler seed and a
disjoint set of pixels per 16x16 tile), so rank r of N renders tiles r, r+N, r+2N, ... of Film::sample_bounds() into a zeroed
Pixel buffer and the buffers are summed ONCE at the end of the frame with a single reduce (RCCL over xGMI on GPUs; gloo in
the CPU tests).  The sum is exact for every pixel that has a single contributor (x + 0 = x); pixels that also received a
"spill" sample from a tile owned by another rank get the same two addends the single-GPU film merge adds.
No other collective exists on this path."""
import numpy as np


def tile_shard(rank, world_size):
    """(first, stride, count) for ftn_tile_range: interleaved so that sky and geometry tiles spread over the ranks."""
    return (int(rank), int(world_size), 0)


def merge_film(pixels, dst=0, group=None):
    """One reduce(sum, f32) of the [h, w, 4] Pixel buffer (xyz + filter_weight_sum) to rank `dst`.
    `pixels` is a torch tensor (cuda -> RCCL, cpu -> gloo); a no-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return pixels
    dist.reduce(pixels, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return pixels


def render_sharded(render_fn, shape, rank, world_size, make_tensor, group=None):
    """render_fn(tiles, out) renders the tile shard into `out` (zeroed [h, w, 4] f32 tensor made by make_tensor);
    returns the merged film on rank 0."""
    out = make_tensor(shape)
    render_fn(tile_shard(rank, world_size), out)
    return merge_film(out, 0, group)


def step_samples(scaling, world_size, spp_per_gpu):
    """Samples per pixel of one bench step.  "weak": every rank renders spp_per_gpu samples of ITS tiles, so a step of the N-rank job
    covers N * spp_per_gpu samples of the film's sample budget (per-GPU work fixed); "strong": a step is spp_per_gpu samples of the
    whole film whatever N is and each rank renders its 1/N of the tiles (total work fixed, per-GPU work shrinks with N)."""
    spp = max(1, int(spp_per_gpu))
    if scaling == "strong":
        return spp
    if scaling != "weak":
        raise ValueError("scaling must be 'weak' or 'strong'")
    return spp * int(world_size)


def rank_spread(values, device="cpu", group=None):
    """{name: x} of this rank -> {name: {"min": .., "max": ..}} over all ranks (two small all-reduces; the identity without a process
    group).  bench.py reports the per-rank device time of a step and the end-of-frame reduce this way, so that a scaling loss can be
    told apart: one slow rank, or the merge."""
    import torch
    import torch.distributed as dist
    names = sorted(values)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return {k: {"min": float(values[k]), "max": float(values[k])} for k in names}
    v = torch.tensor([float(values[k]) for k in names], dtype=torch.float64, device=device)
    lo, hi = v.clone(), v.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return {k: {"min": float(lo[i]), "max": float(hi[i])} for i, k in enumerate(names)}

"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

The path is embarrassingly parallel over rays: the point cloud, grid, MLP weights and
reference-view feature pyramid are replicated on every GPU; rays are split into contiguous scan-line
blocks (neighbouring rays share voxels -> L2 reuse); the only data-path collective is ONE gather of the
rendered colours to rank 0 (3.4 MB for a 620x460 frame) -- the reference's counterpart is the per-chunk
`.cpu()` scatter into the image in run/test_ft.py:185-198.
Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU tensors (tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rays, world_size, rank):
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one ray."""
    base, rem = divmod(int(n_rays), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rays(raydir, world_size, rank):
    lo, hi = shard_bounds(raydir.shape[0], world_size, rank)
    return raydir[lo:hi], (lo, hi)


def gather_rows(local, n_total, dst=0, group=None):
    """Gather per-rank row blocks (possibly of unequal length) to `dst`, in rank order.
    local: [n_local, C].  Returns [n_total, C] on dst, None elsewhere.  One collective."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)]
    pad = max(sizes)
    buf = local
    if local.shape[0] != pad:                      # equal-size buffers keep it a single plain gather
        buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf[:local.shape[0]] = local
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf.contiguous(), outs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0)


def render_sharded(render_fn, raydir, n_total=None, group=None):
    """Each rank renders its block of `raydir` ([R,3], the SAME full tensor on every rank) with
    render_fn(rays) -> [n_local, C]; rank 0 gets the assembled [R, C] image rows."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    rays, (lo, hi) = shard_rays(raydir, world, rank)
    local = render_fn(rays)
    return gather_rows(local, raydir.shape[0] if n_total is None else n_total, group=group)

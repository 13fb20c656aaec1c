"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

The path is embarrassingly parallel over rays: the point cloud, grid, MLP weights and
reference-view feature pyramid are replicated on every GPU; the only data-path collective is ONE gather of the
rendered colours to rank 0 (3.4 MB for a 620x460 frame) -- the reference's counterpart is the per-chunk
`.cpu()` scatter into the image in run/test_ft.py:185-198.

Two ways to split a frame's rays:
  shard_bounds  N contiguous blocks of scan lines (the reference's chunk order).  The work per ray is NOT uniform over an image -- on
                the bench frame the busiest of 8 blocks has 1.68x the mean number of neighbour rows (tools/shard_balance.py), which
                caps strong scaling at 60 %;
  shard_lines   whole scan lines dealt round-robin (rank r renders lines r, r + N, ...): rays of a line stay together (neighbouring
                rays share voxels -> L2 reuse) and every rank sees every part of the image: busiest rank 1.01x the mean at N = 8.
Every ray is rendered independently of its launch mates (tests: chunked == whole frame, bit for bit), so both reassemble the
single-GPU image exactly.
Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU tensors (tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rays, world_size, rank):
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one ray."""
    base, rem = divmod(int(n_rays), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_lines(n_rays, line, world_size, rank):
    """Indices (ascending, int64, CPU) of the rays of rank `rank` when the scan lines -- runs of `line` consecutive rays, the last one
    possibly shorter -- are dealt round-robin to the ranks."""
    idx = torch.arange(int(n_rays), dtype=torch.int64)
    return idx[(idx // int(line)) % int(world_size) == int(rank)]


def gather_indexed(local, index_of_rank, n_total, dst=0, group=None):
    """Reassemble rows rendered under an index sharding: rank r holds the rows index_of_rank(r) of the [n_total, C] result.
    ONE gather of equal-size buffers (padded to the largest shard), then a scatter into place on `dst`.  None on the other ranks."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    idx = [index_of_rank(r) for r in range(world)]
    pad = max(int(i.numel()) for i in idx)
    buf = local
    if local.shape[0] != pad:
        buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf[:local.shape[0]] = local
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf.contiguous(), outs, dst=dst, group=group)
    if rank != dst:
        return None
    full = torch.empty((int(n_total),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for o, i in zip(outs, idx):
        full[i.to(full.device)] = o[:i.numel()]
    return full


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


def render_sharded(render_fn, raydir, n_total=None, group=None, line=None):
    """Each rank renders its share of `raydir` ([R,3], the SAME full tensor on every rank) with
    render_fn(rays) -> [n_local, C]; rank 0 gets the assembled [R, C] image rows.  line = image width: scan lines dealt round-robin
    (balanced, see the module docstring); line = None: contiguous blocks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if line is not None:
        R = raydir.shape[0]
        mine = shard_lines(R, line, world, rank).to(raydir.device)
        local = render_fn(raydir.index_select(0, mine))
        return gather_indexed(local, lambda r: shard_lines(R, line, world, r), R if n_total is None else n_total, group=group)
    rays, (lo, hi) = shard_rays(raydir, world, rank)
    local = render_fn(rays)
    return gather_rows(local, raydir.shape[0] if n_total is None else n_total, group=group)


# ---------------------------------------------------------------------------------------------------------------
# Training: the batch is sharded by WHOLE dilated patches, every rank runs forward +
# backward on its rays with the replicated cloud and weights, and the gradients are summed so that both optimisers
# (models/mvs_points_volumetric_model.py:94-104 in the reference) step identically on every rank.  The reference has no
# counterpart: its DataParallel wrapper always runs on gpu_ids[0] only.

def shard_patches(patch_num, patch_size, world_size, rank):
    """Whole dilated patches per rank (49 patches on 8 ranks -> 7,6,6,6,6,6,6,6), so the per-patch blur / argmin of the shell needs
    no halo.  The batch is a (patch_num*patch_size)^2 grid of rays in row-major order (data/scannet_ft_dataset.py:899-949), so a
    patch is NOT a contiguous ray range.  Returns (patch ids [n_local], ray indices [n_local * patch_size^2]) with the rays
    packed patch-major (patch, y, x) -- the layout blur.blur_update_output(..., layout="patch_major") expects."""
    lo, hi = shard_bounds(patch_num * patch_num, world_size, rank)
    ids = torch.arange(lo, hi)
    S = patch_num * patch_size
    pi, pj = ids // patch_num, ids % patch_num
    y = torch.arange(patch_size)
    rows = (pi[:, None, None] * patch_size + y[None, :, None]) * S + (pj[:, None, None] * patch_size + y[None, None, :])
    return ids, rows.reshape(-1)


def global_drop_flags(patch_num, patch_size, drop_ratio):
    """[S*S] uint8: the batch-wide image-feature drop pattern of the reference (drop_patch_rays, point_aggregators.py:14-23) by
    GLOBAL ray index; a rank passes its slice to render_train(ray_drop=...).  (Single-GPU training indexes the pattern by
    valid-ray row like the reference; the two agree whenever every ray of the batch finds neighbours.)"""
    S = patch_num * patch_size
    flag = torch.zeros((S, S), dtype=torch.uint8)
    n = int(patch_num * patch_num * drop_ratio)
    row, col = n // patch_num, n % patch_num
    flag[0:row * patch_size, :] = 1
    flag[row * patch_size:row * patch_size + patch_size, 0:col * patch_size] = 1
    return flag.reshape(-1)


def loss_scale(n_local, n_total):
    """A loss that is a MEAN over rays: each rank back-propagates mean_local * n_local / n_total and the summed gradients
    equal those of the global mean."""
    return float(n_local) / float(max(n_total, 1))


def allreduce_gradients(tensors, group=None, bucket_bytes=64 << 20):
    """In-place SUM over ranks of a list of gradient tensors, packed into a few large flat buckets: the network's 449 381
    parameters are one 1.8 MB message (latency-bound); the dense point-buffer gradients (N x 39 floats) go as 64 MB ring
    segments, which is what the point-to-point xGMI links like (per-link bound)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    tensors = [t for t in tensors if t is not None]
    i = 0
    while i < len(tensors):
        j, size = i, 0
        while j < len(tensors) and (j == i or size + tensors[j].numel() * 4 <= bucket_bytes) and tensors[j].dtype == tensors[i].dtype:
            size += tensors[j].numel() * 4
            j += 1
        if j == i + 1:
            t = tensors[i]
            if t.is_contiguous():
                dist.all_reduce(t, group=group)
            else:
                c = t.contiguous()
                dist.all_reduce(c, group=group)
                t.copy_(c)
        else:
            flat = torch.cat([t.reshape(-1) for t in tensors[i:j]])
            dist.all_reduce(flat, group=group)
            off = 0
            for t in tensors[i:j]:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        i = j


def allreduce_point_gradients_sparse(grad, touched, group=None):
    """SUM over ranks of a point-buffer gradient [N, C] that is non-zero only on the rows `touched` (int64 ids, the points
    this rank's batch referenced -- at most ~600 k of N = 2-4 M): all-gather (id, row) pairs and scatter-add locally instead
    of moving N x C floats around the ring.  Returns the summed dense gradient (new tensor)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return grad
    world = dist.get_world_size(group)
    n_local = torch.tensor([touched.numel()], dtype=torch.int64, device=grad.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    cap = int(max(int(c.item()) for c in counts))
    C = grad.shape[1]
    ids = torch.full((cap,), -1, dtype=torch.int64, device=grad.device)
    rows = torch.zeros((cap, C), dtype=grad.dtype, device=grad.device)
    ids[:touched.numel()] = touched
    rows[:touched.numel()] = grad.index_select(0, touched)
    all_ids = [torch.empty_like(ids) for _ in range(world)]
    all_rows = [torch.empty_like(rows) for _ in range(world)]
    dist.all_gather(all_ids, ids, group=group)
    dist.all_gather(all_rows, rows, group=group)
    out = torch.zeros_like(grad)
    for i, r, c in zip(all_ids, all_rows, counts):
        k = int(c.item())
        if k:
            out.index_add_(0, i[:k], r[:k])
    return out


def allreduce_point_buffers_sparse(grads, touched, group=None):
    """SUM over ranks of ALL the point-buffer gradients of a training step -- embeddings [N,32], conf [N,1], dir [N,3], colour [N,3] (any list of
    [N, C_i], [1, N, C_i] or [N] tensors, the first one 2-D or 3-D) -- that are non-zero only on the rows `touched` (int64 ids of the points this rank's rays referenced): the
    touched rows of all buffers are packed side by side into ONE [n, sum C_i] matrix, so the step's point gradients cost one all-gather of ids and one
    of rows (~12 k points x 164 B per rank for a 6-7-patch share of the C5 batch) and a local scatter-add, instead of a dense all-reduce of
    N x 39 floats (312 MB at 2 M points; the dense conf / dir / colour all-reduce alone was 56 MB).  Returns the summed dense gradients (new tensors,
    shaped like the inputs).  Deterministic: every rank adds the ranks' rows in rank order."""
    shapes = [g.shape for g in grads]
    n_rows = grads[0].reshape(-1, grads[0].shape[-1]).shape[0]             # the first buffer is [N, C] / [1, N, C]; the others may be [N] (conf)
    flat = [g.reshape(n_rows, -1) for g in grads]
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [g for g in grads]
    widths = [f.shape[1] for f in flat]
    packed = torch.cat([f.index_select(0, touched) for f in flat], dim=1) if touched.numel() else torch.zeros((0, sum(widths)), dtype=flat[0].dtype, device=flat[0].device)
    world = dist.get_world_size(group)
    n_local = torch.tensor([touched.numel()], dtype=torch.int64, device=packed.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    cap = max(1, int(max(int(c.item()) for c in counts)))
    ids = torch.full((cap,), -1, dtype=torch.int64, device=packed.device)
    rows = torch.zeros((cap, packed.shape[1]), dtype=packed.dtype, device=packed.device)
    ids[:touched.numel()] = touched
    rows[:touched.numel()] = packed
    all_ids = [torch.empty_like(ids) for _ in range(world)]
    all_rows = [torch.empty_like(rows) for _ in range(world)]
    dist.all_gather(all_ids, ids, group=group)
    dist.all_gather(all_rows, rows, group=group)
    outs = [torch.zeros_like(f) for f in flat]
    for i, r, c in zip(all_ids, all_rows, counts):
        k = int(c.item())
        off = 0
        for o, w in zip(outs, widths):
            if k:
                o.index_add_(0, i[:k], r[:k, off:off + w])
            off += w
    return [o.reshape(sh) for o, sh in zip(outs, shapes)]

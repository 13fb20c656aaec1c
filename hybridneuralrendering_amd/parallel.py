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


def _with_point_zero(touched):
    """Point 0 always travels (round-4 advice): the empty neighbour slots' share of d conf_coefficient lands on it through the reference's index clamp
    (neural_points.py:711) whether or not the batch touched it, so a rank's gradient can be non-zero there outside `touched`."""
    if touched.numel() and int(touched.min()) == 0:
        return touched
    return torch.cat([torch.zeros((1,), dtype=touched.dtype, device=touched.device), touched])


def allreduce_point_gradients_sparse(grad, touched, group=None):
    """SUM over ranks of a point-buffer gradient [N, C] that is non-zero only on the rows `touched` (int64 ids, the points
    this rank's batch referenced -- at most ~600 k of N = 2-4 M): all-gather (id, row) pairs and scatter-add locally instead
    of moving N x C floats around the ring.  Returns the summed dense gradient (new tensor).  (The host-synchronising form of round 4; the
    training step uses PointGradExchange below.)"""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return grad
    world = dist.get_world_size(group)
    touched = _with_point_zero(touched)
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
    touched = _with_point_zero(touched)
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



# ---------------------------------------------------------------------------------------------------------------
# The step's two gradient collectives without a host read (round-4 verdict / advice): fixed-capacity buffers, device-side counts.

def allreduce_weight_grads(flat, n_valid, n_payload, group=None):
    """The network's gradients as ONE all-reduce, weighted for a loss that is a mean over the batch's VALID rays: rank r holds the gradient of ITS mean
    (over its n_r valid rays); the gradient of the global mean is sum_r (n_r / n) g_r.  `flat` = TrainPath's flat gradient buffer (all parameters'
    gradients, 256-byte aligned slices) with at least one spare float behind the `n_payload` used ones; n_valid = a one-element device tensor (the loss
    kernel's count of valid rays).  In place: flat[:n_payload] <- sum_r n_r g_r / sum_r n_r.  No host read, one collective.  Returns the global number
    of valid rays (device tensor)."""
    if flat.numel() <= n_payload:
        raise ValueError("allreduce_weight_grads: flat needs a spare element behind the payload")
    nv = n_valid.reshape(-1)[:1].to(flat.dtype)
    flat[:n_payload].mul_(nv)
    flat[n_payload:n_payload + 1].copy_(nv)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat[:n_payload + 1], group=group)
    tot = flat[n_payload:n_payload + 1].clone()
    flat[:n_payload].div_(torch.clamp(tot, min=1.0))
    return tot


class PointGradExchange:
    """SUM over ranks of the point-buffer gradients of a training step as ONE fixed-capacity all-gather of packed records -- no counts exchange, no
    `.item()`, no torch.unique, no dense temporaries ("sparse: all-gather (unique point id, 39-float grad) for touched points only").

    A rank's batch touches a few thousand of the N = 2-4 M points; the forward call leaves their ids (ascending) and their number on the device
    (TrainPath.touched_points).  pack() copies the touched rows of all buffers side by side into `rec` [capacity + 2, 1 + sum C_i] floats:
      row 0              header {number of records, this rank's number of valid rays, overflow flag (records > capacity)}
      rows 1..capacity   {point id (int32 bits), the point's gradient row}; unused slots: id -1, zeros
      row capacity + 1   point 0 when the batch did not touch it: the empty neighbour slots' share of d conf_coefficient lands there through the
                         reference's index clamp (neural_points.py:711) whenever the loss reads conf_coefficient (round-4 advice)
    exchange() is ONE all_gather_into_tensor; apply() rewrites the dense gradients in place as sum_r (n_r / n) g_r with the ranks' rows added in rank
    order on every rank (bit-identical results on all ranks), touching only rows some rank touched."""

    def __init__(self, capacity, widths=(32, 1, 3, 3), group=None):
        self.cap, self.widths, self.group = int(capacity), tuple(int(w) for w in widths), group
        self.W = 1 + sum(self.widths)
        if self.W < 3:
            raise ValueError("PointGradExchange: needs at least two gradient columns (the header holds three values)")

    def _hip(self, flat, ids=None, count=None, n_valid=None):
        """the HIP kernels serve the training step's own layout: fp32 [N,32] [N,1] [N,3] [N,3] on the GPU, int32 ids, an int64 count (HNR_EXCHANGE_TORCH=1: torch ops)"""
        import os
        if os.environ.get("HNR_EXCHANGE_TORCH") == "1" or self.widths != (32, 1, 3, 3):
            return False
        ok = all(f.is_cuda and f.dtype == torch.float32 and f.is_contiguous() for f in flat)
        if ids is not None:
            ok = ok and ids.is_cuda and ids.dtype == torch.int32 and ids.is_contiguous() and count.is_cuda and count.dtype == torch.int64 and ids.numel() >= 1
        if n_valid is not None:
            ok = ok and n_valid.is_cuda
        return ok

    def _flat(self, grads):
        n_rows = grads[0].reshape(-1, grads[0].shape[-1]).shape[0]
        return [g.reshape(n_rows, -1) for g in grads], n_rows

    def pack(self, grads, ids, count, n_valid):
        """grads: the dense gradients ([N, C_i] / [1, N, C_i] / [N]); ids int32 [>= n] ascending touched point ids, count = their number (device tensor);
        n_valid = this rank's number of valid rays (device tensor).  Returns rec (a new tensor)."""
        flat, N = self._flat(grads)
        dev = flat[0].device
        cap = self.cap
        if self._hip(flat, ids, count, n_valid):
            # one launch of libhnr_hip.so (csrc/exchange.hip) instead of ~25 tiny torch kernels; same bits (tests/test_train_gpu.py)
            from . import _lib
            rec = torch.empty((cap + 2, self.W), dtype=torch.float32, device=dev)
            nv = n_valid.reshape(-1)[:1].to(torch.float32).contiguous()
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().hnr_point_grad_pack(_lib.ptr(ids), _lib.ptr(count), cap, _lib.ptr(nv), _lib.ptr(flat[0]), _lib.ptr(flat[1]), _lib.ptr(flat[2]),
                                                          _lib.ptr(flat[3]), _lib.ptr(rec), _lib.stream()), "hnr_point_grad_pack")
            return rec
        cnt = count.reshape(-1)[:1].to(torch.int64)
        n = torch.clamp(cnt, max=cap)
        slot = torch.arange(cap, device=dev)
        live = slot < n
        idc = ids[:cap].to(torch.int64) if ids.numel() >= cap else torch.cat([ids.to(torch.int64), torch.zeros((cap - ids.numel(),), dtype=torch.int64, device=dev)])
        idc = torch.where(live, idc, slot % N)                            # padded slots read distinct rows (their values are masked below)
        rec = torch.zeros((cap + 2, self.W), dtype=flat[0].dtype, device=dev)
        body = torch.cat([f.index_select(0, idc) for f in flat], dim=1)
        body = torch.where(live[:, None], body, torch.zeros_like(body))     # (+0.0 in the unused slots: x * 0 would keep the sign of x)
        rec[1:cap + 1, 1:] = body
        rec[1:cap + 1, 0] = torch.where(live, idc, torch.full_like(idc, -1)).to(torch.int32).view(torch.float32)
        zero_in = (n > 0) & (idc[:1] == 0)                                  # ascending ids: point 0 is touched iff it comes first
        row0 = torch.cat([f[0] for f in flat])
        rec[cap + 1, 1:] = torch.where(zero_in, torch.zeros_like(row0), row0)
        rec[cap + 1, 0] = torch.where(zero_in, torch.full((1,), -1, dtype=torch.int32, device=dev), torch.zeros((1,), dtype=torch.int32, device=dev)).view(torch.float32)[0]
        rec[0, 0] = n.to(flat[0].dtype)[0]
        rec[0, 1] = n_valid.reshape(-1)[0].to(flat[0].dtype)
        rec[0, 2] = (cnt > cap).to(flat[0].dtype)[0]
        return rec

    def raise_on_overflow(self, flag):
        """Reads the (running maximum of the) overflow flag `apply` returns -- a host synchronisation -- and raises when it is set."""
        if float(flag) != 0.0:
            from ._lib import HnrError
            raise HnrError("PointGradExchange: a rank touched more than the %d points the exchange was sized for; its extra gradient rows were not "
                           "exchanged (replicas diverge) -- rebuild with a larger capacity" % self.cap)

    def exchange(self, rec):
        """[world, capacity + 2, W]: every rank's records.  ONE collective (none at world size 1)."""
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return rec[None]
        world = dist.get_world_size(self.group)
        out = torch.empty((world,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
        if rec.is_cuda:
            dist.all_gather_into_tensor(out, rec.contiguous(), group=self.group)
        else:                                                               # gloo (the CPU tests): the list form of the same collective
            parts = [torch.empty_like(rec) for _ in range(world)]
            dist.all_gather(parts, rec.contiguous(), group=self.group)
            out = torch.stack(parts)
        return out

    def apply(self, all_rec, grads, own_rank):
        """In place: every dense gradient becomes sum_r (n_r / n) g_r.  Returns (global number of valid rays, overflow flag) as device tensors.
        OVERFLOW (flag != 0: some rank touched more points than `capacity`) is an error the caller must act on: that rank's extra rows were not sent
        and stay in ITS dense gradient as local, unscaled values, so the replicated point buffers would diverge if the optimiser stepped on them.
        Keep the running maximum of the flag on the device and pass it to `raise_on_overflow` at a point where a host read is acceptable (once per
        logging interval), then rebuild the exchange with a larger capacity and redo the affected steps."""
        flat, N = self._flat(grads)
        cap = self.cap
        if self._hip(flat) and all_rec.is_cuda and all_rec.is_contiguous():
            from . import _lib
            out2 = torch.empty((2,), dtype=torch.float32, device=all_rec.device)
            with torch.cuda.device(all_rec.device):
                _lib.check(_lib.lib().hnr_point_grad_apply(_lib.ptr(all_rec), int(all_rec.shape[0]), cap, _lib.ptr(flat[0]), _lib.ptr(flat[1]), _lib.ptr(flat[2]),
                                                           _lib.ptr(flat[3]), N, _lib.ptr(out2), _lib.stream()), "hnr_point_grad_apply")
            return out2[0], out2[1]
        n_r = all_rec[:, 0, 1]
        tot = torch.clamp(n_r.sum(), min=1.0)
        scale = n_r / tot
        ids = all_rec[:, 1:, 0].contiguous().view(torch.int32).to(torch.int64)          # [world, cap + 1]
        live = ids >= 0
        slot = torch.arange(cap + 1, device=all_rec.device)
        ids = torch.where(live, ids, (slot % N)[None, :])                                # padded slots add exact zeros to distinct rows
        body = all_rec[:, 1:, 1:]
        order = [(own_rank, -1.0)] + [(r, None) for r in range(all_rec.shape[0])]
        for r, sgn in order:
            off = 0
            for f, w in zip(flat, self.widths):
                rows = body[r, :, off:off + w]
                rows = rows * sgn if sgn is not None else rows * scale[r]            # first: minus the rank's own rows (x - x = 0 exactly), then all ranks in rank order
                f.index_add_(0, ids[r], rows.to(f.dtype))
                off += w
        return tot, all_rec[:, 0, 2].max()

"""Batch axis of the hot path: independent frames sharded over the GPUs of one node.

The reference's only parallel axis is independent images: ``cv::parallel_for_(Range(0, N), body)``
with one SteerableFiltersG2 per file (example/steer.cpp:69-124,169).  Here that axis maps onto
one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm):

  * frame f of F goes to rank floor(f * G / F) -- contiguous blocks, SURVEY.md 8(e);
  * the per-frame work needs no cross-frame exchange, so the data path has NO collective;
  * RCCL moves data only at the edges: `scatter_frames` (root -> ranks, grouped send/recv =
    one point-to-point xGMI link per peer) and `gather_planes` (ranks -> root).

The collective plumbing is backend-agnostic (tests run it on CPU with gloo, world_size 2); the
per-frame compute is the HIP engine and nothing else -- `process_frames` takes the engine's
bound method, there is no CPU implementation in this package.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, world, rank):
    """contiguous block [lo, hi) of rank: item f belongs to rank floor(f*world/n_items)"""
    if n_items <= 0:
        return 0, 0
    lo = (rank * n_items + world - 1) // world
    hi = ((rank + 1) * n_items + world - 1) // world
    return lo, hi


def shard_counts(n_items, world):
    return [shard_range(n_items, world, r)[1] - shard_range(n_items, world, r)[0] for r in range(world)]


def _group_info(group):
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def scatter_frames(frames, n_frames, frame_shape, device, root=0, group=None, dtype=torch.float32):
    """Root holds `frames` [n_frames, H, W]; every rank returns its contiguous block [n_local, H, W].

    Implemented as grouped point-to-point sends (ncclSend/ncclRecv under RCCL): each peer's block
    crosses exactly one xGMI link, root's own block is a local copy."""
    world, rank = _group_info(group)
    lo, hi = shard_range(n_frames, world, rank)
    local = torch.empty((hi - lo,) + tuple(frame_shape), dtype=dtype, device=device)
    if world == 1:
        local.copy_(frames[lo:hi])
        return local
    ops = []
    if rank == root:
        for r in range(world):
            rlo, rhi = shard_range(n_frames, world, r)
            if rhi == rlo:
                continue
            if r == root:
                local.copy_(frames[rlo:rhi])
            else:
                ops.append(dist.P2POp(dist.isend, frames[rlo:rhi].contiguous(), r, group))
    elif hi > lo:
        ops.append(dist.P2POp(dist.irecv, local, root, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return local


def gather_planes(local, n_frames, root=0, group=None):
    """Every rank holds `local` [n_local, K, H, W] (its block, in frame order); root returns
    [n_frames, K, H, W], other ranks return None.  Grouped point-to-point, like scatter_frames."""
    world, rank = _group_info(group)
    if world == 1:
        return local
    out = None
    ops = []
    if rank == root:
        out = torch.empty((n_frames,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        for r in range(world):
            rlo, rhi = shard_range(n_frames, world, r)
            if rhi == rlo:
                continue
            if r == root:
                out[rlo:rhi].copy_(local)
            else:
                ops.append(dist.P2POp(dist.irecv, out[rlo:rhi], r, group))
    elif local.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, local.contiguous(), root, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def process_frames(frames, frame_fn, n_outputs):
    """Run `frame_fn(frame, outs)` on every local frame; outs = list of n_outputs [H, W] views into
    the result [n_local, n_outputs, H, W].  frame_fn is an engine method, e.g.
    ``lambda img, outs: engine.pipeline(img, out=outs)`` (8 outputs)."""
    n, h, w = frames.shape
    result = torch.empty((n, n_outputs, h, w), dtype=torch.float32, device=frames.device)
    for i in range(n):
        frame_fn(frames[i], [result[i, k] for k in range(n_outputs)])
    return result


def run_sharded(frames_on_root, n_frames, frame_shape, device, frame_fn, n_outputs, root=0, group=None, gather=True):
    """scatter -> per-frame engine work -> (optional) gather.  Returns (local_result, gathered)."""
    local = scatter_frames(frames_on_root, n_frames, frame_shape, device, root, group)
    result = process_frames(local, frame_fn, n_outputs)
    return result, (gather_planes(result, n_frames, root, group) if gather else None)

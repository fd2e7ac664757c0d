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


# ----------------------------------------------------------------------------------------------------------------
# Native path: a binding of cvs_batch_* (cvsteer_amd/csrc/cvs_batch.cpp).  Sharding, staging, the per-rank launch
# and the RCCL calls (ncclSend / ncclRecv groups, ncclBroadcast) all happen in the library; this class only hands
# over pointers.  The torch.distributed functions above are the same plan expressed with torch collectives -- they
# exist so that the N > 1 bookkeeping can be tested on CPU (gloo); on a GPU box the native path is the product.
# ----------------------------------------------------------------------------------------------------------------
import ctypes as _C

import numpy as _np

from . import _lib as _L
from ._lib import CvsError as _CvsError


class NativeBatch:
    """The batch axis of example/steer.cpp:169 over the GPUs of one node, through the C ABI.

    NativeBatch.local(devices)            -- this process drives all listed devices (ncclCommInitAll)
    NativeBatch.from_torch_distributed()  -- one process per GPU: rank 0 makes the RCCL id, torch.distributed
                                             (already initialised) carries the 128 bytes to the other ranks
    """

    def __init__(self, handle, world, rank, device, devices=None):
        self._b, self.world, self.rank, self.device = handle, world, rank, device
        self.devices = list(devices) if devices is not None else [device]   # local ranks' devices, rank order
        self._conv = {}

    # -- construction --
    @classmethod
    def local(cls, devices=(0,), kind=_L.KIND_G2, width=4, spacing=0.67):
        devs = (_C.c_int * len(devices))(*[int(d) for d in devices])
        h = _C.c_void_p()
        rc = _L.lib().cvs_batch_create_local(kind, width, spacing, len(devices), devs, _C.byref(h))
        nb = cls._made(rc, h, len(devices), 0, int(devices[0]), "cvs_batch_create_local")
        nb.devices = [int(d) for d in devices]
        return nb

    @classmethod
    def from_torch_distributed(cls, device, kind=_L.KIND_G2, width=4, spacing=0.67, group=None):
        world, rank = _group_info(group)
        ident = (_C.c_char * _L.BATCH_ID_BYTES)()
        if rank == 0:
            rc = _L.lib().cvs_batch_unique_id(ident)
            if rc:
                raise _CvsError(rc, "cvs_batch_unique_id", "RCCL could not be loaded")
        box = [bytes(ident)]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        ident = (_C.c_char * _L.BATCH_ID_BYTES).from_buffer_copy(box[0])
        h = _C.c_void_p()
        rc = _L.lib().cvs_batch_create_rank(kind, width, spacing, ident, world, rank, int(device), _C.byref(h))
        return cls._made(rc, h, world, rank, int(device), "cvs_batch_create_rank")

    @classmethod
    def _made(cls, rc, h, world, rank, device, where):
        if rc:
            msg = _L.lib().cvs_batch_last_error(h).decode() if h else ""
            if h:
                _L.lib().cvs_batch_destroy(h)
            raise _CvsError(rc, where, msg)
        return cls(h, world, rank, device)

    def close(self):
        h, self._b = getattr(self, "_b", None), None
        if h:
            _L.lib().cvs_batch_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: module globals may be gone already
            pass

    def _check(self, rc, where):
        if rc:
            raise _CvsError(rc, where, _L.lib().cvs_batch_last_error(self._b).decode())

    @property
    def transport(self):
        w, n, t = _C.c_int(), _C.c_int(), _C.c_int()
        self._check(_L.lib().cvs_batch_info(self._b, _C.byref(w), _C.byref(n), _C.byref(t)), "cvs_batch_info")
        return {_L.TRANSPORT_NONE: "none", _L.TRANSPORT_RCCL: "rccl", _L.TRANSPORT_COPY: "device copies (rehearsal)"}[t.value]

    def set_persist(self, on):
        self._check(_L.lib().cvs_batch_set_option(self._b, _L.OPT_PERSIST_STATE, 1 if on else 0), "cvs_batch_set_option")

    # -- config 4 --
    def run(self, frames, n_frames, frame_shape, outputs=(5, 6, 7), out=None, root=0, gather=True, self_via_transport=False):
        """frames: [n_frames, H, W] float32 CUDA tensor on the root (None elsewhere), or a numpy array of that shape
        (host planes: every rank must be local to this process; each GPU pulls its frames over its own link).  Returns
        (out [n_frames, len(outputs), H, W] on the root or None, {'scatter','compute','gather'} milliseconds)."""
        rows, cols = (int(v) for v in frame_shape)
        sel = [int(k) for k in outputs]
        cfg = _L.BatchCfg(rows, cols, int(n_frames), sum(1 << k for k in sel), int(root), 1 if gather else 0, 1 if self_via_transport else 0)
        imgs = outs = None
        is_root = frames is not None
        if is_root and isinstance(frames, _np.ndarray):
            from .api import _PLANE_DTYPE
            item = frames.dtype.itemsize
            assert frames.dtype in (_np.float32, _np.uint8) and frames.shape == (n_frames, rows, cols) and frames.strides[2] == item and frames.strides[1] >= cols * item
            imgs = _np.zeros(n_frames, _PLANE_DTYPE)
            imgs["data"] = frames.ctypes.data + _np.arange(n_frames, dtype=_np.uint64) * _np.uint64(frames.strides[0])
            imgs["rows"], imgs["cols"], imgs["step"] = rows, cols, frames.strides[1]
            imgs["mem"] = _L.MEM_HOST | (_L.DEPTH_U8 if frames.dtype == _np.uint8 else 0)   # 8-bit frames cross the link as bytes
            if gather:
                if out is None:
                    out = _np.empty((n_frames, len(sel), rows, cols), _np.float32)
                # a uint8 `out` asks for the maps as bytes (normalised / converted on the GPUs, cvs_batch_set_u8_gain)
                assert isinstance(out, _np.ndarray) and out.dtype in (_np.float32, _np.uint8) and out.flags.c_contiguous and out.shape == (n_frames, len(sel), rows, cols)
                osz = out.dtype.itemsize
                outs = _np.zeros((n_frames, 8), _PLANE_DTYPE)
                off = _np.arange(n_frames, dtype=_np.uint64) * _np.uint64(len(sel) * rows * cols * osz)
                for j, k in enumerate(sel):
                    outs["data"][:, k] = out.ctypes.data + off + _np.uint64(j * rows * cols * osz)
                    outs["rows"][:, k], outs["cols"][:, k], outs["step"][:, k] = rows, cols, cols * osz
                    outs["mem"][:, k] = _L.MEM_HOST | (_L.DEPTH_U8 if osz == 1 else 0)
        elif is_root:
            assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous() and tuple(frames.shape) == (n_frames, rows, cols)
            from .api import _PLANE_DTYPE
            imgs = _np.zeros(n_frames, _PLANE_DTYPE)
            imgs["data"] = frames.data_ptr() + _np.arange(n_frames, dtype=_np.uint64) * _np.uint64(rows * cols * 4)
            imgs["rows"], imgs["cols"], imgs["step"], imgs["mem"] = rows, cols, cols * 4, _L.MEM_DEVICE
            if gather:
                if out is None:
                    out = torch.empty((n_frames, len(sel), rows, cols), dtype=torch.float32, device=frames.device)
                assert out.is_cuda and out.is_contiguous() and tuple(out.shape) == (n_frames, len(sel), rows, cols)
                outs = _np.zeros((n_frames, 8), _PLANE_DTYPE)
                off = _np.arange(n_frames, dtype=_np.uint64) * _np.uint64(len(sel) * rows * cols * 4)
                for j, k in enumerate(sel):
                    outs["data"][:, k] = out.data_ptr() + off + _np.uint64(j * rows * cols * 4)
                    outs["rows"][:, k], outs["cols"][:, k], outs["step"][:, k], outs["mem"][:, k] = rows, cols, cols * 4, _L.MEM_DEVICE
            torch.cuda.current_stream(frames.device).synchronize()  # the library works on its own streams
        t = _L.BatchTiming()
        rc = _L.lib().cvs_batch_run(self._b, _C.byref(cfg), imgs.ctypes.data_as(_L._PP) if imgs is not None else None,
                                    outs.ctypes.data_as(_L._PP) if outs is not None else None, _C.byref(t))
        self._check(rc, "cvs_batch_run")
        return (out if (is_root and gather) else None), {"scatter": t.scatter_ms, "compute": t.compute_ms, "gather": t.gather_ms}

    def run_to_u8(self, frames, outputs=(5, 6, 7), gain=0.0, out=None):
        """The flow of example/steer.cpp:69-122 for one batch of equally sized HOST images (numpy [n, H, W], uint8 or
        float32), every rank local to this process, as ONE native call: host planes up, the caller pipeline, every map
        normalised (gain = 0: normalize(0, 255, MINMAX)) or converted (convertTo(gain)) on its GPU, bytes down --
        chunk by chunk, the upload and launches of chunk c+1 overlapping the download of chunk c.
        -> numpy uint8 [n, len(outputs), H, W]."""
        n, rows, cols = (int(v) for v in frames.shape)
        sel = [int(k) for k in outputs]
        if out is None:
            out = _np.empty((n, len(sel), rows, cols), _np.uint8)
        assert isinstance(out, _np.ndarray) and out.dtype == _np.uint8
        self._check(_L.lib().cvs_batch_set_u8_gain(self._b, float(gain)), "cvs_batch_set_u8_gain")
        return self.run(frames, n, (rows, cols), outputs=sel, out=out)

    def run_to_u8_two_step(self, frames, outputs=(5, 6, 7), gain=0.0, out=None):
        """the same result in two steps (round 2's first form, kept as the reference of the one-call flow): cvs_batch_run
        with the maps kept on the GPUs, then cvs_*_u8_batch on each GPU (one launch pair, one copy, one sync per rank)"""
        n, rows, cols = (int(v) for v in frames.shape)
        sel = [int(k) for k in outputs]
        _, t = self.run(frames, n, (rows, cols), outputs=sel, gather=False)
        if out is None:
            out = _np.empty((n, len(sel), rows, cols), _np.uint8)
        assert isinstance(out, _np.ndarray) and out.dtype == _np.uint8 and out.flags.c_contiguous and out.shape == (n, len(sel), rows, cols)
        lib = _L.lib()
        f0 = 0
        for r, dev in enumerate(self.devices):
            data, nf, npl, rr, cc = _L._FP(), _C.c_int(), _C.c_int(), _C.c_int(), _C.c_int()
            self._check(lib.cvs_batch_local_result(self._b, r, _C.byref(data), _C.byref(nf), _C.byref(npl), _C.byref(rr), _C.byref(cc)), "cvs_batch_local_result")
            m = nf.value * npl.value
            if not m:
                continue
            base = _C.cast(data, _C.c_void_p).value
            if r not in self._conv:
                from .api import SteerableFiltersG2
                self._conv[r] = SteerableFiltersG2(None, 4, 0.67, device=dev)
            h = self._conv[r]._h
            planes = (_L.Plane * m)()
            dst = (_C.c_void_p * m)()
            for i in range(m):
                planes[i] = _L.Plane(base + i * rows * cols * 4, rows, cols, cols * 4, _L.MEM_DEVICE)
                dst[i] = out.ctypes.data + (f0 * len(sel) + i) * rows * cols
            if gain > 0:
                rc = lib.cvs_convert_u8_batch(h, planes, m, float(gain), 0.0, dst, cols, _L.MEM_HOST)
            else:
                rc = lib.cvs_normalize_u8_batch(h, planes, m, dst, cols, _L.MEM_HOST)
            if rc:
                raise _CvsError(rc, "cvs_*_u8_batch", lib.cvs_last_error(h).decode())
            f0 += nf.value
        assert f0 == n
        return out, t

    # -- config 3 --
    def pyramid_setup(self, image, rows, cols, levels, flags=1, root=0):
        """image: [H, W] float32 CUDA tensor on the root (None elsewhere).  Returns timing in milliseconds."""
        pl = None
        if image is not None:
            assert image.is_cuda and image.dtype == torch.float32 and image.is_contiguous() and tuple(image.shape) == (rows, cols)
            pl = _L.Plane(image.data_ptr(), rows, cols, cols * 4, _L.MEM_DEVICE)
            torch.cuda.current_stream(image.device).synchronize()
        t = _L.BatchTiming()
        rc = _L.lib().cvs_batch_pyramid_setup(self._b, _C.byref(pl) if pl is not None else None, int(rows), int(cols), int(levels),
                                              int(flags), int(root), _C.byref(t))
        self._check(rc, "cvs_batch_pyramid_setup")
        self._keep = image
        return {"broadcast": t.scatter_ms, "compute": t.compute_ms, "gather": t.gather_ms}

    def level_plane(self, level, which):
        """root only: state plane `which` (cvs_lib PLANE_*) of pyramid level `level`, copied into a new CUDA tensor"""
        h = _C.c_void_p()
        self._check(_L.lib().cvs_batch_level(self._b, int(level), _C.byref(h), None), "cvs_batch_level")
        v = _L.Plane()
        rc = _L.lib().cvs_state_plane(h, int(which), _C.byref(v))
        if rc:
            raise _CvsError(rc, "cvs_state_plane", _L.lib().cvs_last_error(h).decode())
        out = torch.empty((v.rows, v.cols), dtype=torch.float32, device=torch.device("cuda", self.device))
        dst = _L.Plane(out.data_ptr(), v.rows, v.cols, v.cols * 4, _L.MEM_DEVICE)
        rc = _L.lib().cvs_read_state(h, int(which), _C.byref(dst))
        if rc:
            raise _CvsError(rc, "cvs_read_state", _L.lib().cvs_last_error(h).decode())
        _L.lib().cvs_sync(h)
        return out

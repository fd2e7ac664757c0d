"""CPU tests (-m "not gpu"): the N>1 batch path -- sharding + scatter/gather -- with gloo, world_size 2.

The collective plumbing in cvsteer_amd/batch.py is backend-agnostic; on the GPU box it runs over
RCCL.  The per-frame compute injected here is the CPU oracle (tests may use it); the product's
own frame function is the HIP engine."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cvsteer_amd import batch


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 5, 32, 256, 257):
        for world in (1, 2, 3, 8):
            ranges = [batch.shard_range(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for a, b in zip(ranges, ranges[1:]):
                assert a[1] == b[0]
            counts = batch.shard_counts(n, world)
            assert sum(counts) == n and max(counts) - min(counts) <= 1
    # BASELINE config 4: 256 frames over 8 GPUs = 32 each
    assert batch.shard_counts(256, 8) == [32] * 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle as ora
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape = (24, 40)
        frames = None
        if rank == 0:
            frames = torch.from_numpy(np.random.default_rng(5).random((n_frames,) + shape, dtype=np.float32))

        def frame_fn(img, outs):  # CPU stand-in with the engine's signature: 7 basis planes of one frame
            b = ora.basis(ora.KIND_G2, img.numpy(), 4, 0.67)
            for k in range(7):
                outs[k].copy_(torch.from_numpy(b[k]))

        local, gathered = batch.run_sharded(frames, n_frames, shape, torch.device("cpu"), frame_fn, 7)
        lo, hi = batch.shard_range(n_frames, world, rank)
        assert local.shape == (hi - lo, 7) + shape
        if rank == 0:
            want = np.stack([ora.basis(ora.KIND_G2, frames[i].numpy(), 4, 0.67) for i in range(n_frames)])
            assert gathered.shape == want.shape
            assert np.array_equal(gathered.numpy(), want)  # sharded == unsharded, bit for bit
            q.put("ok")
        else:
            assert gathered is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 2, 1])
def test_scatter_process_gather_world2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"

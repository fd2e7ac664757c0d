#!/usr/bin/env python3
"""tools/bench_batch_e2e.py -- BASELINE config 4 end to end: F frames of 1080x1920 on rank 0 ->
scatter over the ranks (RCCL send/recv) -> fused pipeline per rank -> gather the feature maps on rank 0.
Reports scatter / compute / gather separately (the gather, not the compute, is the bound).

    python tools/bench_batch_e2e.py --frames 32                               # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 tools/bench_batch_e2e.py --frames 256
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import cvsteer_amd as cv
from cvsteer_amd import batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    ws, rank, lr = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    if ws > 1:
        dist.init_process_group("nccl", device_id=dev)
    shape = (args.rows, args.cols)
    frames = torch.rand((args.frames,) + shape, device=dev) if rank == 0 else None
    eng = cv.SteerableFiltersG2(None, device=lr)
    eng.set_persist(False)

    def sync():
        torch.cuda.synchronize()
        if ws > 1:
            dist.barrier()

    t = {"scatter": 0.0, "compute": 0.0, "gather": 0.0}
    for rep in range(args.reps + 1):
        sync(); t0 = time.perf_counter()
        local = batch.scatter_frames(frames, args.frames, shape, dev)
        sync(); t1 = time.perf_counter()
        out = eng.pipeline_batch(local, outputs=(5, 6, 7)) if local.shape[0] else torch.empty((0, 3) + shape, device=dev)
        sync(); t2 = time.perf_counter()
        full = batch.gather_planes(out, args.frames)
        sync(); t3 = time.perf_counter()
        if rep:  # first repetition warms up allocations / RCCL channels
            t["scatter"] += t1 - t0; t["compute"] += t2 - t1; t["gather"] += t3 - t2
    if rank == 0:
        pix = args.frames * args.rows * args.cols
        res = {k: round(v / args.reps * 1e3, 3) for k, v in t.items()}
        tot = sum(t.values()) / args.reps
        print(json.dumps({"n_gpus": ws, "frames": args.frames, "ms": res, "compute_only_Mpix/s": round(pix / (t["compute"] / args.reps) / 1e6, 1),
                          "end_to_end_Mpix/s": round(pix / tot / 1e6, 1), "gathered_shape": list(full.shape)}))
    if ws > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""tools/r06_host_probe.py -- the batch driver's shape (8-bit host frames in, three 8-bit maps out, one native call) against the host link's
roof, for the number of chunks the environment names (CVS_BATCH_HOST_CHUNKS; round-6 experiment)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from cvsteer_amd import batch
torch.cuda.init()
pin = torch.empty(64 << 20, dtype=torch.float32).pin_memory()
dbuf = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
def rate(dst, src):
    best = 0.0
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        best = max(best, src.numel() * 4 / (time.perf_counter() - t0) / 1e9)
    return best
h2d, d2h = rate(dbuf, pin), rate(pin, dbuf)
del pin, dbuf
hb = batch.NativeBatch.local((0,))
hb.set_persist(False)
n = 32
u8 = np.random.default_rng(77).integers(0, 256, (n, 1080, 1920), dtype=np.uint8)
q8 = np.zeros((n, 3, 1080, 1920), np.uint8)
ts = []
for rep in range(8):
    t0 = time.perf_counter(); hb.run_to_u8(u8, out=q8); ts.append(time.perf_counter() - t0)
best = min(ts[1:]); pix = n * 1080 * 1920
roof = max(pix / (h2d * 1e9), 3 * pix / (d2h * 1e9))
print("chunks %s: h2d %.1f d2h %.1f GB/s | u8 -> 3 x u8: best %.3f ms, median %.3f ms, %.2f Gpix/s, %.3f of the link roof (%.3f ms)" %
      (os.environ.get("CVS_BATCH_HOST_CHUNKS", "4 (default)"), h2d, d2h, best * 1e3, sorted(ts[1:])[len(ts[1:]) // 2] * 1e3, pix / best / 1e9, roof / best, roof * 1e3))
f32 = np.random.default_rng(78).random((n, 1080, 1920), dtype=np.float32)
o32 = np.zeros((n, 3, 1080, 1920), np.float32)
ts = []
for rep in range(5):
    t0 = time.perf_counter(); hb.run(f32, n, (1080, 1920), outputs=(5, 6, 7), out=o32); ts.append(time.perf_counter() - t0)
best = min(ts[1:]); roof = max(4 * pix / (h2d * 1e9), 12 * pix / (d2h * 1e9))
print("            f32 -> 3 x f32: best %.3f ms, %.2f Gpix/s, %.3f of the link roof (%.3f ms)" % (best * 1e3, pix / best / 1e9, roof / best, roof * 1e3))

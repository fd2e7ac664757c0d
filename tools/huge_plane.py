"""Time the 7-basis + steer pass on an image whose planes exceed 2 GiB (row-banded launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv

rows, cols = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (30000, 40000)))
img = torch.rand((rows, cols), device="cuda")
f = cv.SteerableFiltersG2(None)
g = torch.empty_like(img); h = torch.empty_like(img)
for _ in range(3):
    f.setup_steer(img, 0.3, out=(g, h))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for _ in range(n):
    f.setup_steer(img, 0.3, out=(g, h))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print("%dx%d (%.1f GB/plane): %.3f ms, %.0f Mpix/s, %.1f%% of 8 TB/s at 40 B/pix" %
      (rows, cols, rows * cols * 4 / 1e9, ms, rows * cols / ms / 1e3, rows * cols * 40 / ms / 1e6 / 8e3 * 100 / 1e0))

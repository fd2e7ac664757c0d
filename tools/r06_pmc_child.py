"""tools/r06_pmc_child.py -- the single-image caller pipeline at 4096^2 (M5) and the 32 x 1080p state-kept batch (config 4), tuner off, a few
launches each: the target of the counter-only rocprofv3 passes of tools/r06_pmc.sh (why does the batch run 15-20 % below the single image?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVS_OPTS"] = "autotune=0"
import torch
import cvsteer_amd as cv
n = 4096
img = torch.rand((n, n), device="cuda")
f = cv.SteerableFiltersG2(None, 4, 0.67)
outs8 = cv.alloc_planes(8, n, n, device="cuda")
for _ in range(5):
    f.pipeline(img, out=outs8)
frames = [torch.rand((32, 1080, 1920), device="cuda") for _ in range(2)]
ff = cv.SteerableFiltersG2(None, 4, 0.67)
fo8 = torch.empty((32, 8, 1080, 1920), device="cuda")
for i in range(6):
    ff.pipeline_batch(frames[i & 1], out=fo8)
torch.cuda.synchronize()

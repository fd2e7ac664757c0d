"""tools/pmc_child.py -- the two epilogue-heavy launches (single-image caller pipeline; 32 x 1080p three maps only) three times each, tuner off: the
target of counter-only rocprofv3 passes that compare two library builds (CVSTEER_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVS_OPTS"] = "autotune=0"
import torch
import cvsteer_amd as cv
n = 4096
img = torch.rand((n, n), device="cuda")
f = cv.SteerableFiltersG2(None, 4, 0.67)
outs8 = cv.alloc_planes(8, n, n, device="cuda")
for _ in range(3):
    f.pipeline(img, out=outs8)
frames = torch.rand((32, 1080, 1920), device="cuda")
ff = cv.SteerableFiltersG2(None, 4, 0.67)
ff.set_persist(False)
fo3 = torch.empty((32, 3, 1080, 1920), device="cuda")
for _ in range(3):
    ff.pipeline_batch(frames, out=fo3, outputs=(5, 6, 7))
f4 = cv.SteerableFiltersG4(None, 6, 0.5)
for _ in range(3):
    f4.setup(img)
torch.cuda.synchronize()

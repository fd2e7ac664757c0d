"""tools/valu_child.py -- every timed kernel of bench.py once per size it is timed at, tuner off, so that a counter-only pass
(`rocprofv3 --pmc SQ_INSTS_VALU -- python3 tools/valu_child.py`) gives the vector instructions per launch of each; tools/collect_valu.py
turns them into instructions per output pixel (profiles/valu_insts.json, read by bench.py for the VALU roof)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVS_OPTS"] = "autotune=0"
import torch
import cvsteer_amd as cv
n = 4096
img = torch.rand((n, n), device="cuda")
g, h = cv.alloc_planes(2, n, n, device="cuda")
f = cv.SteerableFiltersG2(None, 4, 0.67)
outs8 = cv.alloc_planes(8, n, n, device="cuda")
for _ in range(3):   # first call of a handle = new image (warmed); the later ones = the resident form the bench times
    f.setup(img, flags=cv.SETUP_BASIS)
for _ in range(3):
    f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
for _ in range(3):
    f.setup(img, flags=cv.SETUP_FULL)
for _ in range(3):
    f.pipeline(img, out=outs8)
f4 = cv.SteerableFiltersG4(None, 6, 0.5)
for _ in range(3):
    f4.setup(img)
for _ in range(3):
    f4.setup_steer(img, 0.3, out=(g, h))
frames = torch.rand((32, 1080, 1920), device="cuda")
ff = cv.SteerableFiltersG2(None, 4, 0.67)
fo3 = torch.empty((32, 3, 1080, 1920), device="cuda")
fo8 = torch.empty((32, 8, 1080, 1920), device="cuda")
for _ in range(3):
    ff.pipeline_batch(frames, out=fo8)
ff.set_persist(False)
for _ in range(3):
    ff.pipeline_batch(frames, out=fo3, outputs=(5, 6, 7))
torch.cuda.synchronize()

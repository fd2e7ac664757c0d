"""tools/r06_pmc_c4_child.py -- the 32 x 1080p state-kept batch (config 4) a few times, allocations as bench.py makes them before that leg (so that the
state block lands where it lands in a bench process): counter-pass target of tools/r06_pmc_c4.sh (is the 0.66 <-> 0.72 process lottery cycles or clock?)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVS_OPTS"] = "autotune=0"
import torch
import cvsteer_amd as cv
# a different allocation history per process: some blocks of random size come and go first
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
junk = [torch.empty(rnd.randrange(1 << 26, 1 << 29), dtype=torch.uint8, device="cuda") for _ in range(rnd.randrange(2, 8))]
for i in sorted(rnd.sample(range(len(junk)), len(junk) // 2), reverse=True):
    del junk[i]
frames = [torch.rand((32, 1080, 1920), device="cuda") for _ in range(2)]
ff = cv.SteerableFiltersG2(None, 4, 0.67)
fo8 = torch.empty((32, 8, 1080, 1920), device="cuda")
for i in range(40):
    ff.pipeline_batch(frames[i & 1], out=fo8)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    ff.pipeline_batch(frames[i & 1], out=fo8)
e1.record()
torch.cuda.synchronize()
print("C4_MS %.4f" % (e0.elapsed_time(e1) / 20), flush=True)

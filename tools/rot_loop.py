#!/usr/bin/env python3
"""tools/rot_loop.py -- nothing but the headline kernel on 8 rotating 4096x4096 inputs (every input read comes from
HBM); the target of the `pmc_rot_*` counter passes of tools/profile.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
n = 4096
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])
f = cv.SteerableFiltersG2(None)
if os.environ.get("ROT_BLOCK_ORDER"):   # e.g. 1000000 = XCD-owned column ranges (diagnostic order)
    from cvsteer_amd import _lib as L
    f.set_option(L.OPT_BLOCK_ORDER, int(os.environ["ROT_BLOCK_ORDER"]))
for i in range(12):
    f.setup_steer(imgs[i & 7], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
torch.cuda.synchronize()

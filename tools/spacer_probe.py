import os, sys
sys.path.insert(0, os.getcwd())
import torch, cvsteer_amd as cv
from cvsteer_amd import _lib as L
img = torch.rand((4096, 4096), device="cuda")
g, h = torch.empty_like(img), torch.empty_like(img)
f = cv.SteerableFiltersG2(None)
f.set_option(L.OPT_PLACEMENT_SEARCH, 1)
f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
torch.cuda.synchronize()
print(f.launch_info())

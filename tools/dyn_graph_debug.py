"""round-5 debug: one dynamic-tail launch in a HIP graph, replays and eager launches mixed -- which rows come out wrong?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
shape = (1400, 1800)
imgs = [torch.rand(shape, device="cuda") for _ in range(2)]
ref = cv.SteerableFiltersG2(None); ref.set_option(L.OPT_BLOCK_ORDER, 0)
want = []
for im in imgs:
    g_, h_ = ref.setup_steer(im, 0.3)
    want.append([g_.clone(), h_.clone()] + [ref.basis(p).clone() for p in range(7)])
f = cv.SteerableFiltersG2(None); f.set_option(L.OPT_BLOCK_ORDER, 2000000); f.set_option(L.OPT_AUTOTUNE, 0)
buf = torch.empty(shape, device="cuda"); g, h = torch.empty_like(buf), torch.empty_like(buf)
run = lambda: f.setup_steer(buf, 0.3, out=(g, h))
def poison():
    g.fill_(-7.0); h.fill_(-7.0)
    for p in range(7): f.basis(p).fill_(-7.0)
def check(which, tag):
    torch.cuda.synchronize()
    cur = [g, h] + [f.basis(p) for p in range(7)]
    bad = [(k, int((a_ != b_).any(dim=1).sum()), (a_ != b_).any(dim=1).nonzero().flatten()[:3].tolist(), (a_ != b_).any(dim=1).nonzero().flatten()[-3:].tolist(),
            int((a_ == -7.0).sum())) for k, (a_, b_) in enumerate(zip(cur, want[which])) if not torch.equal(a_, b_)]
    print(tag, "OK" if not bad else bad[:3], flush=True)
buf.copy_(imgs[0]); run(); check(0, "first eager")
graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
with torch.cuda.stream(side):
    run(); torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        run()
torch.cuda.synchronize()
for n, what in enumerate(["replay", "eager", "replay", "eager", "eager", "eager", "replay", "replay", "eager"]):
    which = n & 1
    buf.copy_(imgs[which]); poison(); torch.cuda.synchronize()
    graph.replay() if what == "replay" else run()
    check(which, (n, what))

#!/usr/bin/env python3
"""tools/ab_c4.py -- A/B of BUILDS (tools/ablibs/<name>.so) on the config-4 legs: 32 x 1080p frames, two alternating
frame sets, whole pipeline with state kept (84 B/pix) and feature maps only (16 B/pix).  Fresh process per build,
alternating.  usage: ab_c4.py cur NEW [rounds]      (child mode: AB_C4_CHILD=1)"""
import os, subprocess, sys, statistics, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("AB_C4_CHILD"):
    sys.path.insert(0, ROOT)
    import torch
    import cvsteer_amd as cv
    nfr = 32
    fs = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
    out = torch.empty((nfr, 8, 1080, 1920), device="cuda")
    out3 = torch.empty((nfr, 3, 1080, 1920), device="cuda")
    pix = nfr * 1080 * 1920
    def run(fn, steps=20):
        for i in range(6): fn(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(steps): fn(i)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / steps
    f = cv.SteerableFiltersG2(None)
    ms = min(run(lambda i: f.pipeline_batch(fs[i & 1], out=out)) for _ in range(3))
    print("C4 state %.1f" % (84 * pix / ms / 1e6 / 80))
    f.set_persist(False)
    ms = min(run(lambda i: f.pipeline_batch(fs[i & 1], out=out3, outputs=(5, 6, 7))) for _ in range(3))
    print("C4 maps-only-Gpix/s %.1f" % (pix / ms / 1e6))
    sys.exit(0)
names = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [3])[0]
res = {}
for r in range(rounds):
    for nm in names:
        env = dict(os.environ, CVSTEER_HIP_LIB=os.path.join(ROOT, "tools", "ablibs", nm + ".so"), AB_C4_CHILD="1")
        o = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
        if o.returncode:
            print(nm, "FAILED", o.stderr[-300:]); continue
        for line in o.stdout.splitlines():
            m = re.match(r"(C4 \S+) ([\d.]+)", line)
            if m: res.setdefault((m.group(1), nm), []).append(float(m.group(2)))
for leg in sorted({k[0] for k in res}):
    print("%-22s " % leg + " | ".join("%s %s (median %.1f)" % (nm, " ".join("%.1f" % v for v in res[(leg, nm)]), statistics.median(res[(leg, nm)])) for nm in names if (leg, nm) in res), flush=True)

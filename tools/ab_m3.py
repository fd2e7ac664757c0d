#!/usr/bin/env python3
"""tools/ab_m3.py -- A/B of BUILDS (tools/ablibs/<name>.so) on the per-pixel legs at 4096^2: scalar steer (36 B/pix),
theta-map steer with e / magnitude / phase (64 B/pix), magnitude + phase (16), find* (20).  Fresh process per build,
alternating.  usage: ab_m3.py cur NEW [rounds]      (child mode: AB_M3_CHILD=1)"""
import os, subprocess, sys, statistics, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("AB_M3_CHILD"):
    sys.path.insert(0, ROOT)
    import torch
    import cvsteer_amd as cv
    n = 4096
    img = torch.rand((n, n), device="cuda")
    f = cv.SteerableFiltersG2(None)
    f.setup(img, flags=cv.SETUP_FULL)
    outs = [torch.empty_like(img) for _ in range(5)]
    def run(fn, steps=30):
        for i in range(5): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(steps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / steps
    f.steer(None, full=True, out=outs)
    legs = [("M3 steer_scalar", lambda: f.steer(0.3, out=outs[:2]), 36), ("M3 steer_map_full", lambda: f.steer(None, full=True, out=outs), 64),
            ("M3 mag_phase", lambda: f.computeMagnitudeAndPhase(outs[0], outs[1]), 16),
            ("M3 find", lambda: f.find(outs[3], outs[4]), 20)]
    for name, fn, bpp in legs:
        try:
            ms = min(run(fn) for _ in range(3))
            print("%s %.1f" % (name, bpp * n * n / ms / 1e6 / 80))
        except Exception as ex:
            print("# %s failed: %s" % (name, ex))
    sys.exit(0)
names = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [3])[0]
res = {}
for r in range(rounds):
    for nm in names:
        env = dict(os.environ, CVSTEER_HIP_LIB=os.path.join(ROOT, "tools", "ablibs", nm + ".so"), AB_M3_CHILD="1")
        o = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
        if o.returncode:
            print(nm, "FAILED", o.stderr[-300:]); continue
        for line in o.stdout.splitlines():
            if line.startswith("#"): print(nm, line)
            m = re.match(r"(M3 \S+) ([\d.]+)", line)
            if m: res.setdefault((m.group(1), nm), []).append(float(m.group(2)))
for leg in sorted({k[0] for k in res}):
    print("%-20s " % leg + " | ".join("%s %s (median %.1f)" % (nm, " ".join("%.1f" % v for v in res[(leg, nm)]), statistics.median(res[(leg, nm)])) for nm in names if (leg, nm) in res), flush=True)

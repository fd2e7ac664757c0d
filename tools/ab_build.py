#!/usr/bin/env python3
"""tools/ab_build.py -- A/B of two BUILDS of libcvsteer_hip.so (tools/ablibs/<name>.so), each run in fresh processes,
alternating, with the allocation-time placement probe and the launch autotuner off (plain hipMalloc state, default launch
configuration), so that what differs is the kernel code.  tools/make_probe_libs.sh builds the probe variants.
usage: ab_build.py cur NOSTORE [more ...] [rounds]"""
import os, subprocess, sys, re, statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = [a for a in sys.argv[1:] if not a.isdigit()] or ["base", "new"]
rounds = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [3])[0]
res = {}
for r in range(rounds):
    for nm in names:
        for kind in ("2", "4"):
            env = dict(os.environ, CVSTEER_HIP_LIB=os.path.join(ROOT, "tools", "ablibs", nm + ".so"), AB_KIND=kind, AB_HANDLES="1")
            spec = "8=1" if kind == "2" else "8=0"
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_same.py"), spec], env=env, capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                print(nm, kind, "FAILED", out.stderr[-400:], flush=True)
                continue
            for line in out.stdout.splitlines():
                m = re.match(r"\s+(M\d \S+)\s+.*\(([\d.]+)%\)", line)
                if m:
                    res.setdefault((m.group(1), nm), []).append(float(m.group(2)))
legs = sorted({k[0] for k in res})
for leg in legs:
    print("%-12s " % leg + " | ".join("%s %s (median %.1f)" % (nm, " ".join("%.1f" % v for v in res.get((leg, nm), [])),
                                                                 statistics.median(res.get((leg, nm), [0]))) for nm in names), flush=True)

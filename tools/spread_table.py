#!/usr/bin/env python3
"""tools/spread_table.py DIR -- min / median / max of every leg over the bench lines in DIR/b*.json (one run per fresh box)"""
import glob, json, os, statistics, sys
d = sys.argv[1]
runs = []
for f in sorted(glob.glob(os.path.join(d, "b*.json"))):
    lines = [l for l in open(f) if l.startswith("{")]
    if lines:
        runs.append(json.loads(lines[-1]))
print("%d runs" % len(runs))
rows = {"headline M2": [r["roofline"]["frac"] for r in runs], "roofline_m1": [r["roofline_m1"]["frac"] for r in runs]}
for r in runs:
    for k, v in r.get("extra", {}).items():
        for key in ("frac_hbm", "whole_frac_hbm"):
            if key in v:
                rows.setdefault(k, []).append(v[key])
        if k.startswith("C4_32x1080p") and "Mpix/s" in v:
            rows.setdefault(k + " Gpix/s", []).append(v["Mpix/s"] / 1000)
for k, v in rows.items():
    print("%-45s n=%d  min %.3f  median %.3f  max %.3f   %s" % (k, len(v), min(v), statistics.median(v), max(v), " ".join("%.3f" % x for x in v)))
cfg = [(r["config"]["launch"]["block_order"], r["config"]["launch"]["strip_rows"]) for r in runs]
print("headline configurations kept by the tuner:", cfg)

#!/usr/bin/env python3
"""tools/bench_picks.py FILE... -- from bench lines: every G2 leg beside the same handle with the tuner off, and what the tuner kept"""
import json, sys
for fn in sys.argv[1:]:
    d = json.loads([l for l in open(fn) if l.startswith("{")][-1])
    e = d["extra"]
    rows = (("M1", d["roofline_m1"]["frac"], d["roofline_m1"]["launch"], e.get("M1_default_same_handle")), ("M2", d["roofline"]["frac"], d["config"]["launch"], e.get("M2_default_same_handle")),
            ("M4", e["M4_full_setup"]["frac_hbm"], e["M4_full_setup"]["launch"], e.get("M4_default_same_handle")), ("M5", e["M5_pipeline"]["frac_hbm"], e["M5_pipeline"]["launch"], e.get("M5_default_same_handle")))
    print(fn)
    for name, fr, li, dflt in rows:
        print("  %s tuned %.3f | default, same handle %s | kept order %d xcd %d strip %d layout %d wg/cu %d" %
              (name, fr, "%.3f (%+.1f %%)" % (dflt["frac_hbm"], 100 * (fr / dflt["frac_hbm"] - 1)) if dflt else "-", li["block_order"], li["xcd_weights"], li["strip_rows"], li["state_layout"], li["wg_per_cu"]))
    print("  other legs:", {k: e[k].get("frac_hbm", e[k].get("whole_frac_hbm")) for k in ("M2_untuned", "M2_rotating_8_inputs", "M6_g4_basis", "C3_pyramid_8192_5_levels", "C4_32x1080p_pipeline_batch")})

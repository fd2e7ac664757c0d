#!/usr/bin/env python3
"""tools/design_table.py label=path.json ... -- the "Measured" table of DESIGN.md section 3 from bench.py lines
(a driver record BENCH_rNN.json is read through its `parsed` field)."""
import json, sys

def load(path):
    txt = open(path).read().strip()
    try:
        d = json.loads(txt)
    except Exception:
        d = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
    if "parsed" in d:
        p = d["parsed"]
        if "extra" not in p:   # the driver keeps the full line in the run's stdout tail
            tail = d["run"]["stdout_tail"]
            p = json.loads([l for l in tail.splitlines() if l.startswith("{")][-1])
        return p
    return d

cols = [(a.split("=", 1)[0], load(a.split("=", 1)[1])) for a in sys.argv[1:]]
def headline(p):
    pl = p.get("config", {}).get("placement")
    tag = "" if not pl else (" (search %s%s)" % ("on" if pl["mode"] else "off", (", window %s, %.1f ms" % ("found" if pl["window_found"] else "not found", pl["probe_ms"])) if pl["mode"] else ""))
    return "%.1f Gpix/s, **%.1f %%**%s" % (p["value"] / 1e3, 100 * p["roofline"]["frac"], tag)
rows = [("**M2 filter+steer (headline `value`)**", 40, headline)]
def leg(name, key="frac_hbm", fmt=None):
    def f(p):
        e = p.get("extra", {}).get(name)
        if not e or "error" in e:
            return "—"
        if fmt:
            return fmt(e)
        return "%.1f %%" % (100 * e[key])
    return f
rows += [
    ("M2, 8 rotating inputs (every input read from HBM)", 40, leg("M2_rotating_8_inputs")),
    ("M2, fresh handle, tuner off (`M2_untuned`)", 40, leg("M2_untuned")),
    ("M2, one new handle per image (`M2_first_call`)", 40, leg("M2_first_call", fmt=lambda e: "%.1f %%, %.2f ms per object" % (100 * e["frac_hbm"], e["ms_object"]))),
    ("the headline loop with the placement search the other way round (`M2_plain_block` = off / `M2_placement_search` = on)", 40,
     lambda p: leg("M2_plain_block", fmt=lambda e: "off: %.1f %%" % (100 * e["frac_hbm"]))(p) if "M2_plain_block" in p.get("extra", {}) else
               leg("M2_placement_search", fmt=lambda e: "on: %.1f %% (window %s, probe %.1f ms)" % (100 * e["frac_hbm"], "found" if e.get("window_found") else "not found", e.get("probe_ms", 0)))(p)),
    ("M1 basis only", 32, leg("M1_basis_only")),
    ("M4 full setup (12 planes)", 52, leg("M4_full_setup")),
    ("M5 whole caller pipeline (20 planes)", 84, leg("M5_pipeline")),
    ("M3 steer scalar / θ-map full", "36 / 64", lambda p: "%s / %s" % (leg("M3_steer_scalar")(p), leg("M3_steer_map_full")(p))),
    ("M6 G4+H4 basis / + scalar steer", "48 / 56", lambda p: "%s / %s" % (leg("M6_g4_basis")(p), leg("M6_g4_filter_steer")(p))),
    ("M1 / M2 on one 8192² image, re-filtered (Infinity-Cache resident)", "32 / 40", lambda p: "%s / %s" % (leg("M1_basis_only_8192")(p), leg("M2_filter_steer_8192")(p))),
    ("M2 on two alternating 8192² images", 40, leg("M2_filter_steer_8192_rotating_2_inputs")),
    ("C4 32 × (1080×1920) pipeline, one launch, state kept", 84, leg("C4_32x1080p_pipeline_batch")),
    ("C4 same frames, feature maps only, no state kept", 16, leg("C4_32x1080p_feature_maps_only", fmt=lambda e: "%.1f Gpix/s" % (e["Mpix/s"] / 1e3))),
    ("C4 end to end through `cvs_batch_run` (1 rank), compute / wall", "—", leg("C4_e2e", fmt=lambda e: "%.3f / %.3f ms" % (e["ms"]["compute"], e["ms_wall"]))),
    ("C4 from HOST planes (f32 in, 3 f32 maps out; PCIe-inclusive)", "—", leg("C4_e2e_host_planes", fmt=lambda e: "%.2f ms (floor %.2f)" % (e["ms_wall"], e["link_floor_ms"]))),
    ("C4 bytes end to end (8-bit in, 3 8-bit maps out; PCIe-inclusive)", "—", leg("C4_e2e_bytes", fmt=lambda e: "%.2f ms (floor %.2f)" % (e["ms_wall"], e["link_floor_ms"]))),
    ("C3 8192² → 5 levels, WHOLE configuration (5 launches, two alternating images)", "33 / 32", leg("C3_pyramid_8192_5_levels", fmt=lambda e: "%.3f ms, %.1f %%" % (e["whole_ms"], 100 * e["whole_frac_hbm"]))),
    ("C3 the five filter launches alone on a prebuilt pyramid", 32, leg("C3_pyramid_8192_5_levels", fmt=lambda e: "%.1f %%" % (100 * e["filter_frac_hbm"]))),
    ("M2 with HOST planes, stream of 8 images (PCIe-inclusive)", "—", leg("M2_host_planes_pcie_inclusive", fmt=lambda e: "%.2f Gpix/s overlapped (%.2f sequential)" % (e["Mpix/s"] / 1e3, e["sequential_Mpix/s"] / 1e3))),
    ("CPU baseline: oracle port, 1 thread / threads × one frame each / one image row-parallel", "—",
     lambda p: "—" if not p.get("cpu_baseline") else "%.3f / %.2f / %.2f Gpix/s" % (p["cpu_baseline"]["value"] / 1e3, p["cpu_baseline"]["one_image_per_thread"]["value"] / 1e3, p["cpu_baseline"]["one_image_row_parallel"]["value"] / 1e3)),
]
print("| leg | B/pix | " + " | ".join(c[0] for c in cols) + " |")
print("|---|---|" + "---|" * len(cols))
for name, bpp, fn in rows:
    print("| %s | %s | " % (name, bpp) + " | ".join(fn(c[1]) for c in cols) + " |")

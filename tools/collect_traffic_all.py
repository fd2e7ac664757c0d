#!/usr/bin/env python3
"""tools/collect_traffic_all.py <round> -- per (kernel instance, grid size): HBM bytes read and written per launch, from the two
counter passes of tools/profile_traffic_all.sh (FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md, calibrated in
profiles/<round>_pmc_traffic.json; WRITE_SIZE exact), next to the algorithmic bytes of the launch where the leg is known.
Writes profiles/<round>_pmc_traffic_all_legs.json."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
G = os.path.join(ROOT, "gpurun_out")

def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(cvs::\w+Args.*$", "", n)

def load(tag):
    files = sorted(glob.glob(os.path.join(G, "pmcall_%s" % tag, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if r["Counter_Name"] == tag and "cvs::" in r["Kernel_Name"]:
            agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]) * 1024.0)
    return agg

f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
P4, P8, F = 4096 * 4096, 8192 * 8192, 32 * 1080 * 1920
# (kernel instance prefix, written MB of the launch to recognise it by) -> leg, algorithmic bytes per launch
known = [
    ("cvs::k_basis<cvs::BankG2, 2, true, 0, true, 4", 9 * 4 * P4, "M2 filter + steer 4096^2", 40 * P4),
    ("cvs::k_basis<cvs::BankG2, 0, true, 0, true, 4", 7 * 4 * P4, "M1 basis 4096^2", 32 * P4),
    ("cvs::k_basis<cvs::BankG2, 0, true, 0, true, 4", 7 * 4 * P8, "M1 basis 8192^2", 32 * P8),
    ("cvs::k_basis<cvs::BankG2, 2, true, 0, true, 4", 9 * 4 * P8, "M2 filter + steer 8192^2", 40 * P8),
    ("cvs::k_basis<cvs::BankG2, 1, true, 0, true, 4", 12 * 4 * P4, "M4 full setup 4096^2", 52 * P4),
    ("cvs::k_basis_lit<cvs::BankG2, 5, true, 0", 20 * 4 * P4, "M5 pipeline 4096^2", 84 * P4),
    ("cvs::k_basis_lit<cvs::BankG2, 5, true, 2", 20 * 4 * F, "C4 32 x 1080p, state kept", 84 * F),
    ("cvs::k_basis_lit<cvs::BankG2, 77, true, 2", 3 * 4 * F, "C4 32 x 1080p, three feature maps", 16 * F),
    ("cvs::k_basis_pair<cvs::BankG4G, cvs::BankG4H, 0, true, true", 11 * 4 * P4, "M6 G4 basis 4096^2 (both half banks: z = 2)", 48 * P4),
    ("cvs::k_basis_pair<cvs::BankG4G, cvs::BankG4H, 2, true, true", 13 * 4 * P4, "M6 G4 basis + steer 4096^2", 56 * P4),
    ("cvs::k_point<(cvs::PointOp)1, 4, true, true", 2 * 4 * P4, "M3 steer scalar 4096^2", 36 * P4),
    ("cvs::k_point<(cvs::PointOp)2, 4, true, true", 5 * 4 * P4, "M3 steer map full 4096^2", 64 * P4),
    ("cvs::k_basis<cvs::BankG2, 16, true, 0, false, 4", 7 * 4 * P8 + P8, "C3 level 0 (8192^2): 7 planes + the next level", 33 * P8),
]
rows = []
for key in sorted(w):
    wr = sum(w[key]) / len(w[key])
    fr = 2.0 * sum(f.get(key, [0.0])) / max(1, len(f.get(key, [])))
    row = {"kernel": key[0], "grid_size": key[1], "launches": len(w[key]), "read_bytes_x2": round(fr), "write_bytes": round(wr), "hbm_bytes": round(fr + wr)}
    for pref, wexp, leg, alg in known:
        if key[0].startswith(pref) and abs(wr - wexp) <= 0.03 * wexp:
            row.update({"leg": leg, "algorithmic_bytes": alg, "hbm_over_algorithmic": round((fr + wr) / alg, 3)})
    rows.append(row)
out = {"method": "see tools/profile_traffic_all.sh (counter-only rocprofv3 passes over bench.py --all-legs, library defaults), resident images unless the leg says otherwise: "
                 "a resident 4096^2 / 1080p input is served by the Infinity Cache, so read_bytes can be BELOW the input's size",
       "rows": rows}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc_traffic_all_legs.json" % rnd), "w"), indent=1)
for r in rows:
    if "leg" in r:
        print("%-48s read %7.1f MB  write %8.1f MB  hbm / algorithmic %.3f  (%d launches)" % (r["leg"], r["read_bytes_x2"] / 1e6, r["write_bytes"] / 1e6, r["hbm_over_algorithmic"], r["launches"]))

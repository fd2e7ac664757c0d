#!/usr/bin/env python3
"""tools/collect_valu.py ROUND -- gpurun_out/pmc_valu (rocprofv3 --pmc SQ_INSTS_VALU over tools/valu_child.py) and gpurun_out/pmc_valu_cal
(the same counter over tools/valu_rate, a kernel of KNOWN vector-instruction count) -> profiles/valu_insts.json:
vector (wave64) instructions per OUTPUT pixel of every timed kernel, halo rows included, corrected by the calibration factor."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
def rows(sub):
    out = []
    fs = sorted(glob.glob(os.path.join(G, sub, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for fn in fs[-1:]:
        out += [r for r in csv.DictReader(open(fn)) if r["Counter_Name"] == "SQ_INSTS_VALU"]
    return out
cal = [float(r["Counter_Value"]) for r in rows("pmc_valu_cal") if "k<0>" in r["Kernel_Name"]]
# tools/valu_rate.hip k<0> (inline-asm v_fma_f32, nothing for the compiler to pack) as tools/profile_all.sh runs it (4096 iterations):
# 2048 workgroups x 4 waves, 4096 iterations x 16 vector instructions per wave (+ ~50 of set-up / final sum)
known = 2048 * 4 * (4096 * 16 + 50)
factor = known / (sum(cal) / len(cal)) if cal else 1.0
pix4096, pixc4 = 4096 * 4096, 32 * 1080 * 1920
legs = {"M1": ("k_basis<cvs::BankG2, 0, true, 0, true", pix4096), "M2": ("k_basis<cvs::BankG2, 2, true, 0, true", pix4096),
        "M4": ("k_basis<cvs::BankG2, 1, true, 0, true", pix4096), "M5": ("k_basis_lit<cvs::BankG2, 5, true, 0", pix4096),
        "M6": ("k_basis_pair<cvs::BankG4G, cvs::BankG4H, 0, true, true", pix4096), "M6s": ("k_basis_pair<cvs::BankG4G, cvs::BankG4H, 2, true, true", pix4096),
        "C4_state": ("k_basis_lit<cvs::BankG2, 5, true, 2", pixc4), "C4_feat3": ("k_basis_lit<cvs::BankG2, 77, true, 2", pixc4)}   # (pipeline variants: the literal-tap instances a default handle runs)
agg = collections.defaultdict(list)
for r in rows("pmc_valu"):
    agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
# cycles per instruction of each kernel's instruction mix (tools/valu_model.py --json on the current ISA; see its header for the model)
try:
    cpi_tab = json.load(open(os.path.join(G, "valu_cpi.json")))
except Exception:
    cpi_tab = {}
out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU -- python3 tools/valu_child.py (counter-only pass), %s; calibration on tools/valu_rate (known %d wave instructions): counter x %.4f" % (rnd, known, factor),
       "unit": "wave64 vector instructions per output pixel (halo rows and window priming included)", "calibration_factor": round(factor, 4), "per_pixel": {}, "per_launch": {},
       "cycles_per_inst_model": "4.4 cycles x max(packed + other, (2 packed + other + simple) / 2) / instructions, classes and costs in tools/valu_model.py / profiles/r05_valu_rate.txt",
       "cycles_per_inst": {}}
for leg, (pat, pix) in legs.items():
    vals = [v for k, vs in agg.items() if pat in k for v in vs]
    if vals:
        v = sorted(vals)[len(vals) // 2] * factor     # the median launch (a handle's first call also requests the image ahead)
        out["per_launch"][leg] = round(v)
        out["per_pixel"][leg] = round(v / pix, 5)
        cp = [c for k, c in cpi_tab.items() if pat in k]
        if cp:
            out["cycles_per_inst"][leg] = cp[0]
json.dump(out, open(os.path.join(P, "valu_insts.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

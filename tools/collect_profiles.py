#!/usr/bin/env python3
"""tools/collect_profiles.py ROUND -- condense gpurun_out/ (scratch) rocprofv3 output of tools/profile.sh
into profiles/ (tracked): kernel-stats CSV, the bench line of the profiled run, the PMC traffic
summary with the gfx950 corrections, and profiles/traffic.json (read by bench.py)."""
import collections, csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs(P, exist_ok=True)

def short(n):
    n = re.sub(r"\((cvs::PointArgs|cvs::BasisArgs|float const\*|float\*|float4|int\*).*", "", n)
    return n.replace("void ", "")

# 1. kernel stats of `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu`
def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: keep only the most recent run's file"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []

stats = newest(os.path.join(G, "prof_stats", "*", "*_kernel_stats.csv"))
if stats:
    rows = [r for r in csv.DictReader(open(stats[0])) if "cvs::" in r["Name"]]
    with open(os.path.join(P, "%s_kernel_stats.csv" % rnd), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    # the same kernel over the TIMED region only: the trace also holds the engine's one-off tuning launches (other strip
    # heights / orders, ~60 launches) and the warm-ups; bench.py's roofline.avg_launch_ms covers the last `steps` launches
    trace = newest(os.path.join(G, "prof_stats", "*", "*_kernel_trace.csv"))
    line = [l for l in open(os.path.join(G, "prof_stats.log")) if l.startswith("{")]
    if trace and line:
        jl = json.loads(line[-1])
        # the timed regions are the bursts of exactly `steps` launches: every region is led into by untimed launches of the same
        # kernel and separated from them by the synchronize + barrier (a gap of tens of microseconds in the trace)
        K = jl["steps"]
        rec = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace[0]))
                     if "k_basis<cvs::BankG2, 2" in r["Kernel_Name"] or "k_basisINS_6BankG2ELi2" in r["Kernel_Name"])
        bursts, cur = [], []
        for st, en in rec:
            if cur and st - cur[-1][1] > 15000:
                bursts.append(cur)
                cur = []
            cur.append((st, en))
        if cur:
            bursts.append(cur)
        regions = [b_ for b_ in bursts if len(b_) == K]
        durs = [en - st for b_ in regions for st, en in b_]
        if durs:
            l2l = [(b_[-1][0] - b_[0][0]) / max(1, len(b_) - 1) for b_ in regions]
            with open(os.path.join(P, "%s_kernel_stats.csv" % rnd), "a") as f:
                w = csv.writer(f)
                w.writerow(["# timed regions: %d bursts of exactly %d launches of the headline kernel (lead-in bursts excluded)" % (len(regions), K), len(durs), sum(durs),
                            "%.1f" % (sum(durs) / len(durs)), min(durs), max(durs),
                            "launch-to-launch %.1f ns (median over the regions); bench.py avg_launch_ms %.5f" % (sorted(l2l)[len(l2l) // 2], jl["roofline"]["avg_launch_ms"])])
    if line:
        open(os.path.join(P, "%s_bench_under_rocprof.json" % rnd), "w").write(line[-1])
# the same headline loop on a plain hipMalloc block (bench.py --placement 0): timed region only
ptrace = newest(os.path.join(G, "prof_stats_plain", "*", "*_kernel_trace.csv"))
pline = [l for l in open(os.path.join(G, "prof_stats_plain.log")) if l.startswith("{")] if os.path.exists(os.path.join(G, "prof_stats_plain.log")) else []
if ptrace and pline:
    jl = json.loads(pline[-1])
    steps = jl["steps"] * int(jl.get("repeats", {}).get("repeats", 1))
    durs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(ptrace[0])) if "k_basis<cvs::BankG2, 2" in r["Kernel_Name"]]
    durs = [d for _, d in sorted(durs)][-steps:]
    with open(os.path.join(P, "%s_kernel_stats_plain_block.csv" % rnd), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "note"])
        w.writerow(["cvs::k_basis<cvs::BankG2, 2, true, 0, true, 4> (timed region, bench.py --placement 0)", len(durs), sum(durs), "%.1f" % (sum(durs) / max(1, len(durs))),
                    min(durs), max(durs), "bench avg_launch_ms %.5f, frac %.4f" % (jl["roofline"]["avg_launch_ms"], jl["roofline"]["frac"])])
    open(os.path.join(P, "%s_bench_under_rocprof_plain_block.json" % rnd), "w").write(pline[-1])
allst = newest(os.path.join(G, "prof_stats_all", "*", "*_kernel_stats.csv"))
if allst:
    rows = [r for r in csv.DictReader(open(allst[0])) if "cvs::" in r["Name"]]
    with open(os.path.join(P, "%s_kernel_stats_all_legs.csv" % rnd), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])

# 2. PMC passes (separate runs, counters only): FETCH_SIZE / WRITE_SIZE per kernel, in KB
def pmc(tag, sub):
    agg = collections.defaultdict(list)
    for fn in newest(os.path.join(G, sub % tag, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == tag:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in agg.items()}

fetch, write = pmc("FETCH_SIZE", "pmc_%s"), pmc("WRITE_SIZE", "pmc_%s")
cal_f, cal_w = pmc("FETCH_SIZE", "pmc_cal_%s"), pmc("WRITE_SIZE", "pmc_cal_%s")
plane = 4096 * 4096 * 4
summary = {
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `python3 bench.py --steps 3 --warmup 1 --no-cpu` "
              "(no tracing combined); counters are KB per dispatch, averaged over dispatches of the kernel. "
              "gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of streamed read bytes -> x2; "
              "WRITE_SIZE is exact.  Both factors re-calibrated here on tools/membench kernels of KNOWN traffic in the same "
              "access shapes (dword/lane strips, float4 linear; plain and nontemporal stores).",
    "calibration": {k: {"FETCH_SIZE_bytes": cal_f.get(k), "WRITE_SIZE_bytes": cal_w.get(k),
                        "known_read_bytes": 0 if "wonly" in k else plane,
                        "known_write_bytes": plane if "copy" in k else 7 * plane} for k in sorted(cal_w)},
    "kernels": {},
}
for k in sorted(write):
    if "cvs::" not in k:
        continue
    f2 = 2.0 * fetch.get(k, 0.0)
    summary["kernels"][k] = {"FETCH_SIZE_bytes_raw": fetch.get(k), "read_bytes_corrected_x2": f2,
                             "WRITE_SIZE_bytes": write[k], "hbm_bytes_per_launch": f2 + write[k]}
# the headline kernel on 8 rotating inputs (tools/rot_loop.py): strips of 10 rows, plain order
rf, rw = pmc("FETCH_SIZE", "pmc_rot_%s"), pmc("WRITE_SIZE", "pmc_rot_%s")
summary["rotating_inputs"] = {k: {"FETCH_SIZE_bytes_raw": rf.get(k), "read_bytes_corrected_x2": 2.0 * rf.get(k, 0.0), "WRITE_SIZE_bytes": rw[k],
                                  "read_amplification_vs_image": 2.0 * rf.get(k, 0.0) / plane} for k in sorted(rw) if "cvs::" in k}
json.dump(summary, open(os.path.join(P, "%s_pmc_traffic.json" % rnd), "w"), indent=1)

# the headline kernel: G2 bank, F_STEER (2), streaming stores, not batched
keys = [k for k in summary["kernels"] if k.startswith("cvs::k_basis<cvs::BankG2, 2, true, 0, true")]
key = keys[0] if keys else None
if key:
    t = summary["kernels"][key]
    json.dump({"k_basis_g2_steer_4096": {"hbm_bytes_per_launch": round(t["hbm_bytes_per_launch"]),
                                         "read_bytes": round(t["read_bytes_corrected_x2"]), "write_bytes": round(t["WRITE_SIZE_bytes"]),
                                         "algorithmic_bytes_per_launch": 40 * 4096 * 4096,
                                         "source": "profiles/%s_pmc_traffic.json" % rnd}},
              open(os.path.join(P, "traffic.json"), "w"), indent=1)
# 3. SQ issue/occupancy counters (own pass)
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in newest(os.path.join(G, "pmc_SQ", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(fn)):
        k = short(r["Kernel_Name"])
        if "cvs::" in k:
            sq[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
if sq:
    with open(os.path.join(P, "%s_pmc_sq.csv" % rnd), "w") as f:
        w = csv.writer(f)
        names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"]
        w.writerow(["Kernel"] + names + ["wait_any_frac", "wait_inst_frac", "active_frac", "valu_frac"])
        for k in sorted(sq):
            m = {n: (sum(sq[k][n]) / len(sq[k][n]) if sq[k][n] else 0.0) for n in names}
            wc = m["SQ_WAVE_CYCLES"] or 1.0
            w.writerow([k] + ["%.6g" % m[n] for n in names] + ["%.3f" % (m[x] / wc) for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU")])
print(open(os.path.join(P, "%s_kernel_stats.csv" % rnd)).read())
print(json.dumps(summary["kernels"], indent=1)[:1500])

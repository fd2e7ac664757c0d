#!/bin/bash
# eight processes with different allocation histories: first the launch's sustained time WITHOUT a profiler, then one counter-only pass each
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06_pmc_c4
rm -rf $O; mkdir -p $O
for s in 1 2 3 4 5 6 7 8; do
  timeout -k 10 120 python3 tools/r06_pmc_c4_child.py $s 2>/dev/null | grep C4_MS > $O/plain_$s.txt
  (cd /tmp && timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_EA0_WRREQ_STALL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES TCC_EA0_WRREQ_LEVEL --output-format csv -d $O/p$s -- python3 $OLDPWD/tools/r06_pmc_c4_child.py $s > $O/p$s.log 2>&1) || echo "pass $s failed"
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.getcwd(), "gpurun_out", "r06_pmc_c4")
print("seed  ms/launch (no profiler)  ms (under counters)  GRBM cycles/8     WRREQ_STALL   WAIT_INST_ANY     WAVE_CYCLES    WRREQ_LEVEL   -> GHz (no profiler)")
for s in range(1, 9):
    plain = float(open(os.path.join(O, "plain_%d.txt" % s)).read().split()[1]) if os.path.getsize(os.path.join(O, "plain_%d.txt" % s)) else float("nan")
    under = [l.split()[1] for l in open(os.path.join(O, "p%d.log" % s)) if l.startswith("C4_MS")]
    agg = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(O, "p%d" % s, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_basis" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v[10:]) / max(1, len(v[10:])) for k, v in agg.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print("%4d  %10.4f  %20s  %14.0f  %14.0f  %14.0f  %14.0f  %14.0f   %.3f" % (s, plain, under[0] if under else "-", cyc, m.get("TCC_EA0_WRREQ_STALL", 0), m.get("SQ_WAIT_INST_ANY", 0),
          m.get("SQ_WAVE_CYCLES", 0), m.get("TCC_EA0_WRREQ_LEVEL", 0), cyc / (plain * 1e-3) / 1e9 if plain == plain else 0))
PY

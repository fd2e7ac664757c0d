#!/bin/bash
# counter-only passes (never combined with tracing) over tools/r06_pmc_child.py; the program itself directly after `--`
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06_pmc
rm -rf $O; mkdir -p $O
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CYCLES" \
            "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_LEVEL" "TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_REQ" \
            "GRBM_GUI_ACTIVE" "TCC_HIT TCC_MISS TCC_WRITEBACK TCC_EA0_WR_UNCACHED_32B" "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 120 rocprofv3 --pmc $ctrs --output-format csv -d $O/p$i -- python3 $OLDPWD/tools/r06_pmc_child.py > $O/p$i.log 2>&1) || echo "pass $i ($ctrs) failed"
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.getcwd(), "gpurun_out", "r06_pmc")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(os.path.join(O, "p*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "k_basis" not in k: continue
        leg = "C4 batch (BATCH=2)" if ", 2, true, 4" in k else "M5 single image"
        agg[leg][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for l in agg.values() for c in l})
with open(os.path.join(O, "summary.txt"), "w") as f:
    f.write("%-36s %18s %18s %10s\n" % ("counter (mean per launch, first 2 launches dropped)", *sorted(agg), "C4/M5 per pixel"))
    pix = {"C4 batch (BATCH=2)": 32 * 1080 * 1920, "M5 single image": 4096 * 4096}
    for c in names:
        vals = {l: (sum(agg[l][c][2:]) / max(1, len(agg[l][c][2:]))) for l in sorted(agg)}
        ls = sorted(agg)
        ratio = (vals[ls[0]] / pix[ls[0]]) / (vals[ls[1]] / pix[ls[1]]) if len(ls) == 2 and vals[ls[1]] else float("nan")
        f.write("%-36s %18.0f %18.0f %10.3f\n" % (c, vals.get(ls[0], 0), vals.get(ls[1], 0) if len(ls) > 1 else 0, ratio))
print(open(os.path.join(O, "summary.txt")).read())
PY

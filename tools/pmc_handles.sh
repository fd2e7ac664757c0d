#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
echo "--- unprofiled speeds of 10 fresh handles (same allocation sequence)"
timeout 120 python3 tools/alloc_modes.py 10 2>&1 | grep "^M4"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_handles_$C -- python3 tools/pmc_handles.py > $O/pmc_handles_$C.log 2>&1
  python3 - $O/pmc_handles_$C $C <<'PY'
import csv, glob, sys, collections
fs = sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))
rows = [r for r in csv.DictReader(open(fs[-1])) if "k_basis" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
disp = collections.OrderedDict()
for r in rows: disp[r["Dispatch_Id"]] = disp.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
vals = list(disp.values())[-50:]
per = [sum(vals[i * 5 + 1:i * 5 + 5]) / 4 for i in range(10)]
print(sys.argv[2], "per launch, by handle (raw counter units):", " ".join("%.0f" % v for v in per))
PY
done

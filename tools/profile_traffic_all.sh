#!/bin/bash
# tools/profile_traffic_all.sh -- HBM traffic of EVERY leg's kernels (not only the headline's): two counter-only passes
# (FETCH_SIZE, WRITE_SIZE; never combined with tracing) over bench.py with its secondary legs, on the library's default
# allocation.  tools/collect_traffic_all.py condenses them into profiles/<round>_pmc_traffic_all_legs.json.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $C --output-format csv -d $O/pmcall_$C -- python3 bench.py --no-live-traffic --steps 3 --warmup 1 --repeats 1 --leg-repeats 1 --lead-ms 2 --no-cpu > $O/pmcall_$C.log 2>&1
  echo "pmcall $C rc=$?" >> $O/pmcall_$C.log
done

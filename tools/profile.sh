#!/bin/bash
# tools/profile.sh -- run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes.
# Summaries land in gpurun_out/prof_*; copy what should be judged into profiles/.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu > $O/prof_stats.log 2>&1
echo "stats rc=$?" >> $O/prof_stats.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_$C.log 2>&1
  echo "pmc $C rc=$?" >> $O/pmc_$C.log
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_cal_$C -- ./tools/membench > $O/pmc_cal_$C.log 2>&1
done
# SQ occupancy / issue counters (own pass): where do the waves spend their cycles
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/pmc_SQ -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_SQ.log 2>&1
echo "pmc SQ rc=$?" >> $O/pmc_SQ.log
find $O -name "*.csv" | head -40

#!/bin/bash
# tools/profile.sh -- run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes.
# Raw output lands in gpurun_out/ (scratch); tools/collect_profiles.py condenses it into profiles/.
#   pass A  kernel-trace + stats of the HEADLINE loop only (bench.py --no-extra): the dominant
#           kernel's average duration here is what bench.py's roofline.avg_launch_ms must agree with
#   pass B  kernel-trace + stats with every secondary leg
#   pass C  PMC FETCH_SIZE / WRITE_SIZE of the headline loop (own runs, counters only) + the same on tools/membench
#           kernels of known traffic (calibration)
#   pass D  SQ issue / occupancy counters (own run)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --no-live-traffic --steps 100 --warmup 10 --repeats 5 --no-cpu --no-extra > $O/prof_stats.log 2>&1
echo "stats rc=$?" >> $O/prof_stats.log
# (since round 4 the headline IS the library's default allocation: no separate plain-block pass)
[ -x tools/membench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench > $O/membench_build.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_all -- python3 bench.py --all-legs --no-live-traffic --steps 50 --warmup 5 --repeats 3 --leg-repeats 1 --no-cpu > $O/prof_stats_all.log 2>&1
# (counter passes: the library's default, a plain state block -- few launches in all, the probe's would be among them)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -- python3 bench.py --no-live-traffic --steps 5 --warmup 3 --repeats 1 --lead-ms 2 --no-cpu --no-extra > $O/pmc_$C.log 2>&1
  echo "pmc $C rc=$?" >> $O/pmc_$C.log
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_cal_$C -- ./tools/membench > $O/pmc_cal_$C.log 2>&1
done
# the same counters for a stream of fresh images (8 rotating inputs: every input read comes from HBM)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_rot_$C -- python3 tools/rot_loop.py > $O/pmc_rot_$C.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/pmc_SQ -- python3 bench.py --no-live-traffic --steps 3 --warmup 1 --repeats 1 --leg-repeats 1 --lead-ms 2 --no-cpu > $O/pmc_SQ.log 2>&1
echo "pmc SQ rc=$?" >> $O/pmc_SQ.log

#!/bin/bash
# tools/profile_all.sh -- one GPU call: kernel-trace stats (headline; all legs), PMC traffic (headline; all legs), SQ counters,
# vector-instruction counts with their calibration.  Raw output in gpurun_out/; tools/collect_*.py condense it into profiles/.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
bash tools/profile.sh
bash tools/profile_traffic_all.sh
[ -x tools/valu_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate > $O/valu_rate_build.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $O/pmc_valu_cal -- ./tools/valu_rate 4096 > $O/pmc_valu_cal.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $O/pmc_valu -- python3 tools/valu_child.py > $O/pmc_valu.log 2>&1
echo "valu rc=$?" >> $O/pmc_valu.log
ls $O | head -50

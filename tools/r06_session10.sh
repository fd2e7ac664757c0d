#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s10
mkdir -p $O
for i in 1 2 3; do timeout -k 10 300 python tools/r06_probe.py lit > $O/lit_$i.txt 2>&1 || echo "lit $i failed"; done
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_canary.py > $O/gpu_pytest.txt 2>&1; echo "gpu pytest rc $?"
tail -n 3 $O/gpu_pytest.txt; cat $O/lit_1.txt

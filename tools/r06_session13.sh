#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s13
mkdir -p $O
for i in 1 2 3; do CVS_OPTS=verbose=1 timeout -k 10 400 python tools/r06_probe.py tune > $O/tune_$i.txt 2>&1 || echo "tune $i failed"; done
timeout -k 10 600 python -m pytest tests/test_gpu_tuner.py -x -q > $O/gpu_pytest.txt 2>&1; echo "gpu pytest rc $?"
for i in 1 2; do CVS_OPTS=verbose=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench$i.json 2> $O/bench$i.err; echo "bench rc $?"; done
tail -n 3 $O/gpu_pytest.txt; grep -h "tuned " $O/tune_*.txt | grep -v cvsteer

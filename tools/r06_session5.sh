#!/bin/bash
# round 6, GPU session 5: the refactored kernels and final defaults -- GPU suite, host path with growing chunks, bench, tuner value probe
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s5
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_pytest.txt 2>&1; echo "gpu pytest rc $?"
for i in 1 2 3; do timeout -k 10 200 python tools/r06_host_probe.py > $O/host_$i.txt 2>&1 || echo "host $i failed"; done
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for i in 1 2; do timeout -k 10 300 python tools/r06_probe.py tune > $O/tune_$i.txt 2>&1 || echo "tune $i failed"; done
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc $?"
tail -n 4 $O/gpu_pytest.txt; grep chunks -A1 $O/host_*.txt

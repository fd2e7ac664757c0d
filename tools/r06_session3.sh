#!/bin/bash
# round 6, GPU session 3: the whole GPU suite on the new defaults, the bench line, C3 localisation, counters for C4 vs M5
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s3
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_pytest.txt 2>&1; echo "gpu pytest rc $?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for i in 1 2; do timeout -k 10 300 python tools/r06_probe.py c3x > $O/c3x_$i.txt 2>&1 || echo "c3x $i failed"; done
bash tools/r06_pmc.sh > $O/pmc.txt 2>&1 || echo "pmc failed"
tail -n 6 $O/gpu_pytest.txt
tail -c 1500 $O/bench.json

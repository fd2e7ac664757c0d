#!/bin/bash
# round 6, GPU session 6: strip heights of mid-size images, the tuner's adoption rule, host path, bench
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s6
mkdir -p $O
for i in 1 2; do timeout -k 10 400 python tools/r06_probe.py strips > $O/strips_$i.txt 2>&1 || echo "strips $i failed"; done
for i in 1 2 3; do timeout -k 10 300 python tools/r06_probe.py tune > $O/tune_$i.txt 2>&1 || echo "tune $i failed"; done
timeout -k 10 200 python tools/r06_host_probe.py > $O/host_1.txt 2>&1 || echo "host failed"
timeout -k 10 600 python -m pytest tests/test_gpu_tuner.py tests/test_gpu_batch.py -x -q > $O/gpu_pytest.txt 2>&1; echo "gpu pytest rc $?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -n 3 $O/gpu_pytest.txt; grep -h "tuned" $O/tune_*.txt | grep -v cvsteer; grep chunks -A1 $O/host_1.txt

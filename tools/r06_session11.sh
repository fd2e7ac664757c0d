#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s11
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_pytest.txt 2>&1; echo "gpu pytest rc $?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench1.json 2> $O/bench1.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench2.json 2> $O/bench2.err; echo "bench rc $?"
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"
tail -n 3 $O/gpu_pytest.txt; tail -2 $O/smoke.txt

#!/bin/bash
# round 6, GPU session 4: C4 merged per frame + read-ahead, host chunking, the GPU suite again, bench
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s4
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_pytest.txt 2>&1; echo "gpu pytest rc $?"
for i in 1 2 3; do timeout -k 10 200 python tools/r06_probe.py c4m > $O/c4m_$i.txt 2>&1 || echo "c4m $i failed"; done
for ch in 4 8 16; do CVS_BATCH_HOST_CHUNKS=$ch timeout -k 10 200 python tools/r06_host_probe.py > $O/host_chunks_$ch.txt 2>&1 || echo "host $ch failed"; done
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -n 4 $O/gpu_pytest.txt; cat $O/c4m_1.txt; cat $O/host_chunks_*.txt

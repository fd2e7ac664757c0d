#!/bin/bash
# round 6, GPU session 2: canary twins, the new tuner, C3 per level, C4 with frames requested ahead
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s2
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_canary.py -x -q > $O/canary_pytest.txt 2>&1; echo "canary rc $?"
timeout -k 10 600 python -m pytest tests/test_gpu_tuner.py -x -q > $O/tuner_pytest.txt 2>&1; echo "tuner rc $?"
for i in 1 2; do timeout -k 10 300 python tools/r06_probe.py c3lv > $O/c3lv_$i.txt 2>&1 || echo "c3lv $i failed"; done
for i in 1 2; do timeout -k 10 200 python tools/r06_probe.py c4w > $O/c4w_$i.txt 2>&1 || echo "c4w $i failed"; done
for i in 1 2 3; do timeout -k 10 300 python tools/r06_probe.py tune > $O/tune_$i.txt 2>&1 || echo "tune $i failed"; done
tail -n 5 $O/canary_pytest.txt $O/tuner_pytest.txt

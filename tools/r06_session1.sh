#!/bin/bash
# round 6, GPU session 1: the A/B probes, several processes each (allocation history differs per process)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s1
mkdir -p $O
for i in 1 2 3 4; do timeout -k 10 200 python tools/r06_probe.py c4 > $O/c4_$i.txt 2>&1 || echo "c4 $i failed" ; done
for i in 1 2 3 4 5 6; do timeout -k 10 200 python tools/r06_probe.py m4 > $O/m4_$i.txt 2>&1 || echo "m4 $i failed" ; done
for i in 1 2; do timeout -k 10 300 python tools/r06_probe.py c3 > $O/c3_$i.txt 2>&1 || echo "c3 $i failed" ; done
for i in 1 2; do timeout -k 10 300 python tools/r06_probe.py fresh > $O/fresh_$i.txt 2>&1 || echo "fresh $i failed" ; done
tail -n 30 $O/c4_1.txt

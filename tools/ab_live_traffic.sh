# tools/ab_live_traffic.sh -- headline / M1 with and without the two rocprofv3 --pmc child runs of bench.py (`roofline.traffic` measured by the run
# itself).  With the children IN FRONT of the timed part 2 of 4 runs read 3-6 % slow (gpurun_out/ab_live.txt); bench.py now starts them after it.
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu --no-live-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nolive', d['roofline']['frac'], d['roofline_m1']['frac'])"
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('live  ', d['roofline']['frac'], d['roofline_m1']['frac'], d['roofline']['traffic'])"
done

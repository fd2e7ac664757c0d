# A/B of builds (tools/ablibs/<name>.so) on bench.py's headline loop (placement search on, tuner on), fresh processes, interleaved
for i in 1 2 3 4; do
  for L in "$@"; do
    CVS_TUNE_VERBOSE=1 CVSTEER_HIP_LIB=$PWD/tools/ablibs/$L.so python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>/tmp/ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L run $i: %.4f  (window %s, %.1f ms)  launch %s' % (d['roofline']['frac'], d['config']['placement']['window_found'], d['config']['placement']['probe_ms'], {k:v for k,v in d['config'].get('launch',{}).items() if k!='note'}))"
    grep "placement probe" /tmp/ab_err.txt | cut -c1-900
  done
done

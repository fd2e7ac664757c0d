export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
for try in 1 2 3; do
  rm -rf $O/prof_stats
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 100 --warmup 10 --repeats 5 --no-cpu --no-extra > $O/prof_stats.log 2>&1
  echo "stats rc=$?" >> $O/prof_stats.log
  if grep -q '"window_found": true' $O/prof_stats.log; then echo "window found on try $try"; break; fi
done

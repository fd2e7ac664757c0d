for i in 1 2 3; do
  for L in g4old g4new; do
    echo "== $L run $i"
    CVSTEER_HIP_LIB=$PWD/tools/ablibs/$L.so python tools/r3_probe.py g4 2>&1 | grep -E "split 2"
  done
done

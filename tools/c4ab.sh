# A/B of two builds of the library on the config-4 legs (separate processes, three runs each, interleaved)
A=${1:-c4cur}; B=${2:-c4w5}
for i in 1 2 3; do
  for L in $A $B; do
    echo "== $L run $i"
    CVSTEER_HIP_LIB=$PWD/tools/ablibs/$L.so python tools/r3_probe.py c4strips 2>&1 | grep -E "strip_rows=  0"
  done
done

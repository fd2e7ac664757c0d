# A/B of two builds of the library on the G4 legs (separate processes, three runs each, interleaved)
A=${1:-g4cur}; B=${2:-g4w4}
for i in 1 2 3; do
  for L in $A $B; do
    echo "== $L run $i"
    CVSTEER_HIP_LIB=$PWD/tools/ablibs/$L.so python tools/r3_probe.py g4 2>&1 | grep -E "split 2 strip 40"
  done
done

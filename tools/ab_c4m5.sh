# A/B of two builds on the pipeline legs: 32 x 1080p batch (state kept / three maps) and M1 / M2 / M4 / M5 at 4096^2 (placement search on)
for i in 1 2 3; do
  for L in "$@"; do
    echo "== $L run $i"
    CVSTEER_HIP_LIB=$PWD/tools/ablibs/$L.so python tools/r3_probe.py c4strips 2>&1 | grep -E "strip_rows=  0"
    CVS_PLACEMENT_SEARCH=1 CVSTEER_HIP_LIB=$PWD/tools/ablibs/$L.so AB_HANDLES=1 python tools/ab_same.py "8=1" 2>&1 | grep -E "M[0-9] " | cut -c1-100
  done
done

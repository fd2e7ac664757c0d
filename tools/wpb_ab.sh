# A/B of workgroup sizes (tools/make_probe_libs.sh WPB1 WPB2): resident legs through ab_build.py, fresh images through ab_rot.py
python tools/ab_build.py cur WPB1 WPB2 3
for L in cur WPB1 WPB2 cur WPB1 WPB2; do echo "== $L"; CVSTEER_HIP_LIB=$PWD/tools/ablibs/$L.so python tools/ab_rot.py 2>&1 | cut -c1-200; done

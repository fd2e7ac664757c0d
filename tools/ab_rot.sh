#!/bin/bash
# tools/ab_rot.sh -- builds (tools/ablibs/<name>.so) compared on a stream of FRESH images: 8 rotating 4096^2 inputs, fresh processes, alternating
for r in 1 2 3; do for nm in "$@"; do echo "== $nm"; CVSTEER_HIP_LIB=$PWD/tools/ablibs/$nm.so AB_ROT=1 AB_HANDLES=1 python tools/ab_same.py "8=0" 2>&1 | grep rotating; done; done

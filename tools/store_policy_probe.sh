#!/bin/bash
# tools/store_policy_probe.sh -- the streaming stores' cache policy: the product's `nt` against twins built with `nt sc0` (aux 3) and
# `nt sc1` (aux 18) (`make -C cvsteer_amd/csrc storepolicy`), one process each, three rounds taking turns on one box.  The legs that
# write one plane group (M1, M2) differ little from block to block, so differences between the libraries show through the block lottery.
R=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2 3; do
  for lib in "" tools/libcvsteer_hip_aux3.so tools/libcvsteer_hip_aux18.so; do
    if [ -n "$lib" ]; then export CVSTEER_HIP_LIB=$R/$lib; else unset CVSTEER_HIP_LIB; fi
    printf "%-34s " "${lib:-product (nt)}"
    python3 tools/region_probe.py 0 2>&1 | grep dummy | sed 's/dummy   0.0 GiB: //; s/state at.*//'
  done
done

#!/bin/bash
# Runs scripts/box_stage_ab.py once per library variant (mpc_amd/variants/lib_<name>.so) so that
# builds of the box second stage can be compared in one GPU call.
#   scripts/box_variants.sh <name> [<name> ...]
set -e
mkdir -p gpurun_out/r03/box
for v in "$@"; do
  echo "== $v"
  MPC_AMD_LIB=$PWD/mpc_amd/variants/lib_$v.so timeout -k 10 120 python scripts/box_stage_ab.py 2>/dev/null | tee gpurun_out/r03/box/ab_$v.txt
done

#!/bin/bash
# A/B of library variants (mpc_amd/variants/lib_<name>.so): digest of every output of the box X axis
# (scripts/box_stage_ab.py) and the headline figures of the given workloads, one variant after the other.
#   scripts/ab_variants.sh "<workloads>" <name> [<name> ...]
set -e
WLS=$1
shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  echo "== $v"
  MPC_AMD_LIB=$PWD/mpc_amd/variants/lib_$v.so timeout -k 10 120 python3 scripts/box_stage_ab.py
done
bash scripts/variant_sweep.sh "$WLS" "$@"

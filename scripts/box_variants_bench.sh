#!/bin/bash
# The decoupled bench.py line once per library variant (mpc_amd/variants/lib_<name>.so): ms per step.
#   scripts/box_variants_bench.sh <name> [<name> ...]
set -e
mkdir -p gpurun_out/r03/box
for v in "$@"; do
  MPC_AMD_LIB=$PWD/mpc_amd/variants/lib_$v.so timeout -k 10 200 python bench.py --workload decoupled --steps 20 --warmup 5 > gpurun_out/r03/box/bench_$v.json 2>/dev/null
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
d=json.loads(open('gpurun_out/r03/box/bench_%s.json'%v).read().strip().splitlines()[-1])
print(v,'ms_per_step %.3f value %.4g'%(d['ms_per_step'],d['value']))
PY
done

#!/bin/bash
# Times library variants (mpc_amd/variants/lib_<name>.so, selected through MPC_AMD_LIB) on the
# given workloads: one bench.py run each, headline figures only.
#   scripts/variant_sweep.sh "<workloads>" <name> [<name> ...]
set -e
WLS=$1
shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in $WLS; do
  for v in "$@"; do
    MPC_AMD_LIB=$PWD/mpc_amd/variants/lib_$v.so timeout -k 10 120 python3 bench.py --workload $w --host-steps 0 --cpu-sample 0 > gpurun_out/var_${w}_$v.log 2>&1
    python3 - "$w" "$v" <<'P'
import json, sys
w, v = sys.argv[1:3]
d = json.loads(open('gpurun_out/var_%s_%s.log' % (w, v)).read().strip().splitlines()[-1])
o = d.get('longest_first_dispatch') or {}
print('%-13s %-8s %.4g  ms %.4f  second %.4g  hinted %.4g' % (w, v, d['value'], d['ms_per_step'], d['second_pass']['value'], o.get('value', 0)), flush=True)
P
  done
done

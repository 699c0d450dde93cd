#!/bin/bash
# Quick A/B numbers for kernel work: headline workload (and optionally others), no CPU baseline.
# usage: scripts/quick_bench.sh tag [workload ...]
tag=$1; shift
wls=${@:-coupled_vel}
mkdir -p gpurun_out/r03/quick
for w in $wls; do
  timeout -k 10 200 python bench.py --workload $w --steps 50 --warmup 5 --cpu-sample 0 --host-steps 0 > gpurun_out/r03/quick/${tag}_$w.log 2>&1 || exit 1
  python - "$w" gpurun_out/r03/quick/${tag}_$w.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-13s value %.4g  ms/step %.4f  second_pass %.4g  pipelined %s  iters_mean %.3f ws_max %d status %s" % (
            sys.argv[1], d["value"], d["ms_per_step"], d.get("second_pass", {}).get("value", 0),
            ("%.4g" % d["pipelined_sub_batches"]["value"]) if "pipelined_sub_batches" in d else "-",
            d["solver"]["outer_iterations_mean"], d["solver"]["working_set_max"], d["solver"]["status_counts"]))
PY
done

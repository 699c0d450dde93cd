# The committed bench lines of a round: every bench.py workload at --steps 20 --warmup 5 (plus the
# variants the design notes quote) into gpurun_out/r03/final, one JSON file each.
set -e
mkdir -p gpurun_out/r03/final
for w in coupled_vel coupled_jerk coupled_ref replanning decoupled; do
  timeout -k 10 250 python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/r03/final/bench_$w.json 2> gpurun_out/r03/final/bench_$w.err
  echo "$w done"
done
timeout -k 10 250 python bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 > gpurun_out/r03/final/bench_gloo2.json 2> gpurun_out/r03/final/bench_gloo2.err
echo gloo2 done

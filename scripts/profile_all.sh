set -e
for wl in coupled_vel decoupled coupled_jerk coupled_ref replanning; do
  bash scripts/profile_gpu.sh $wl r2_prof_$wl > gpurun_out/r2_prof_$wl.out 2>&1
  echo "$wl profiled"
done

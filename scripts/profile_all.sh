set -e
# Profiles every bench.py workload (kernel trace + counter passes, scripts/profile_gpu.sh) into
# gpurun_out/${PREFIX}<workload>; scripts/refresh_profiles.py condenses them into profiles/rNN.
PREFIX=${1:-r3_prof_}
for wl in coupled_vel decoupled coupled_jerk coupled_ref replanning; do
  bash scripts/profile_gpu.sh $wl ${PREFIX}$wl > gpurun_out/${PREFIX}$wl.out 2>&1
  echo "$wl profiled"
done

"""The two remaining benchmark workloads on the GPU against the CPU restatement, at (or near) their
benchmark sizes: the ZMP-reference form on the full 8,192-instance batch, and the receding-horizon
sweep (8 dependent replans, Gaussian weights, CoM step correction, disturbance) on the first
`rollouts` rollouts of the benchmark's 16,384.

    python scripts/check_ref_replanning_full.py [rollouts=1024]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from mpc_amd import batches, planner as P  # noqa: E402
from oracle import oracle as O  # noqa: E402

N, steps, stages = 200, 20, 8
TRAJ = ("jerk_x", "jerk_y", "com_x", "com_y", "zmp_x", "zmp_y")


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


# ---- ZMP-reference form, the bench's weights (R = 0.01, Q = 1, Qs = 100) ----------------------
B = 8192
bt = batches.coupled_batch(B)
t = time.time()
ref = O.coupled_batch(O.params(N, R=0.01, Q=1.0, Qs=100.0, height=0.5), O.COUPLED_REF, steps, 1000, bt["x0"], bt["y0"],
                      bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"], nthreads=16)
print("cpu restatement (ZMP-reference form): %.1f s" % (time.time() - t), flush=True)
pl = P.BatchedPlanner(N, formulation=P.COUPLED_REF, weight_R=0.01, weight_Q=1.0, weight_Qs=100.0, number_of_steps=steps)
o = {k: v.cpu().numpy() for k, v in pl.solve_coupled(**bt).items()}
bad_ws = sum(list(o["active_idx"][i, :o["n_active"][i]]) != list(ref["active"][i, :ref["n_active"][i]]) for i in range(B))
print("COUPLED_REF, %d instances: status differs on %d, iteration count on %d, ordered working set on %d; worst relative "
      "trajectory error %.2e, objective %.2e; handed over %d" %
      (B, int((o["status"] != ref["status"]).sum()), int((o["iters"] != ref["iters"]).sum()), bad_ws,
       max(rel(o[k], ref[k]) for k in TRAJ), float((np.abs(o["cost"] - ref["cost"]) / np.abs(ref["cost"])).max()),
       pl.last_handover_count()), flush=True)

# ---- receding-horizon sweep ---------------------------------------------------------------------
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bt = batches.coupled_batch(R)
dist = batches.replanning_disturbance(R)
hip = batches.NOMINAL_HIP_OFFSETS
t = time.time()
ref = O.replanning_batch(O.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000), steps, stages, bt["x0"], bt["y0"],
                         bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"], dist, hip=hip, nthreads=16)
print("cpu restatement (replanning, %d x %d solves): %.1f s" % (R, stages, time.time() - t), flush=True)
pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, replanning_window=N // steps,
                      number_of_steps=steps, hip_offsets=hip)
res, stage, st = pl.replan_sweep(pl.prepare_replanning(disturbance=dist, **bt), stages)
sc = stage["cost"].cpu().numpy()
print("replanning, %d rollouts x %d stages: per-stage status differs on %d solves, working-set size on %d; worst relative "
      "error of a stage's objective %.2e; final states: CoM %.2e, feet %.2e (absolute), swing leg differs on %d; "
      "last stage's jerk %.2e" %
      (R, stages, int((stage["status"].cpu().numpy() != ref["stage_status"]).sum()),
       int((stage["n_active"].cpu().numpy() != ref["stage_nactive"]).sum()),
       float((np.abs(sc - ref["stage_cost"]) / np.abs(ref["stage_cost"])).max()),
       max(float(np.abs(st[k].cpu().numpy() - ref[k]).max()) for k in ("x0", "y0")),
       max(float(np.abs(st[k].cpu().numpy() - ref[k]).max()) for k in ("feet_x", "feet_y")),
       int((st["swing0"].cpu().numpy() != ref["swing0"]).sum()),
       max(rel(res[k].cpu().numpy(), ref[k]) for k in ("jerk_x", "jerk_y") if k in ref)), flush=True)

"""Replanning sweep of 24 stages: index order against stages ordered by the one a gait cycle before."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mpc_amd import batches, planner as P
B, N, STAGES = 16384, 200, 24
inputs = batches.coupled_batch(B)
dist = batches.replanning_disturbance(B)
pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, replanning_window=10,
                      number_of_steps=20, hip_offsets=batches.NOMINAL_HIP_OFFSETS)
out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
for hint in (0, 4, 0, 4, 1):
    pl.set_dispatch_hint(hint)
    ts = []
    for rep in range(3):
        prepared = pl.prepare_replanning(disturbance=dist, **inputs)
        torch.cuda.synchronize()
        t = time.perf_counter()
        pl.replan_sweep(prepared, STAGES, out=out, record=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    print("hint %d: %d stages in %s ms" % (hint, STAGES, " ".join("%.2f" % x for x in ts)))

"""The 512-instance coupled batch at the constructor's default weight_R = 1e-6 (working sets of 20-53
rows: more than half of the instances outgrow the first edition of the wave-per-QP kernel and are
finished by its wide edition): ms per solve, how many instances took which path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mpc_amd import batches, planner as P  # noqa: E402

for R in (1e-6, 1e-4):
    B = 512
    bt = batches.coupled_batch(B)
    pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL, weight_R=R, weight_Q=1.0, number_of_steps=20)
    prep = pl.prepare_coupled(**bt)
    out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
    for _ in range(5):
        pl.solve_coupled_prepared(prep, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        pl.solve_coupled_prepared(prep, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 20 * 1e3
    o = out[0]
    print("weight_R %g: %.3f ms per solve of %d instances; wide edition %d, dense stage %d; iterations sum %d, "
          "working sets up to %d rows, status %s" %
          (R, ms, B, pl.last_wide_count(), pl.last_handover_count(), int(o["iters"].sum().item()),
           int(o["n_active"].max().item()), torch.bincount(o["status"]).tolist()))

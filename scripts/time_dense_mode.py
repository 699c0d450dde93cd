"""Times the dense second-stage kernel alone on the coupled shape (solver_mode = dense)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_amd import batches, planner as P

for B in (1, 16, 256):
    bt = batches.coupled_batch(B)
    pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=20,
                          solver_mode=1)
    prep = pl.prepare_coupled(**bt)
    out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
    pl.solve_coupled_prepared(prep, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        pl.solve_coupled_prepared(prep, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    it = out[0]["iters"].float().mean().item()
    print("B %d: %.2f ms per batch, %.1f iterations per instance on average" % (B, dt * 1e3, it))

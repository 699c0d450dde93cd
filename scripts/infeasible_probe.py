"""Coupled batch with part of the instances made infeasible or nearly so (walking speeds scaled up,
so that consecutive support polygons stop overlapping): which stage ends up solving them, what
they cost, and whether the end status agrees with the CPU restatement."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mpc_amd import batches, planner as P  # noqa: E402
from oracle import oracle as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, steps = 200, 20
for scale in (5.0, 10.0, 20.0, 40.0):
    bt = batches.coupled_batch(B)
    hard = np.arange(B) % 8 == 0
    bt["speed"] = np.where(hard[:, None], bt["speed"] * scale, bt["speed"])
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps, active_stride=192)
    prep = pl.prepare_coupled(**bt)
    out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
    for _ in range(3):
        pl.solve_coupled_prepared(prep, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        pl.solve_coupled_prepared(prep, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 5 * 1e3
    o = {k: v.cpu().numpy() for k, v in out[0].items()}
    ref = O.coupled_batch(O.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=8000), O.COUPLED_VEL, steps, 1000, bt["x0"], bt["y0"],
                          bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"], active_stride=192, nthreads=16)
    print("speed x %4.0f on every 8th: %.3f ms; wide %d block %d dense %d; status %s (restatement %s); rows up to %d; "
          "iters up to %d; status equal %d of %d, iterations equal %d" %
          (scale, ms, pl.last_wide_count(), pl.last_block_count(), pl.last_handover_count(), np.bincount(o["status"], minlength=3).tolist(),
           np.bincount(ref["status"], minlength=3).tolist(), o["n_active"].max(), o["iters"].max(),
           int((o["status"] == ref["status"]).sum()), B, int((o["iters"] == ref["iters"]).sum())), flush=True)

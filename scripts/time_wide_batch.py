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

# Beyond 60 rows: weight_R = 1e-8 at three times the walking speed pushes a few working sets past
# the wide edition's capacity.  The
# batch is timed whole and again without the instances that went to the dense stage: the difference
# is what those few cost.
B = 2048
bt = batches.coupled_batch(B)
bt["speed"] = bt["speed"] * 3.0
pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL, weight_R=1e-8, weight_Q=1.0, number_of_steps=20)


def timed(sub):
    prep = pl.prepare_coupled(**sub)
    n = prep[0]
    out = pl._alloc_outputs(n, P.ALL_OUTPUTS)
    for _ in range(3):
        pl.solve_coupled_prepared(prep, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        pl.solve_coupled_prepared(prep, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 10 * 1e3, out[0]


ms_all, o = timed(bt)
dense = pl.last_handover_count()
block = pl.last_block_count()
big = (o["n_active"] > 60).cpu().numpy() | (o["status"] != 0).cpu().numpy()
keep = ~big
ms_rest, _ = timed({k: v[keep] for k, v in bt.items()})
print("weight_R 1e-8, speed x 3: %.3f ms per solve of %d instances (block edition %d, dense stage %d, working sets up to %d rows, status %s); "
      "%.3f ms without the %d instances that end with more than 60 rows (dense stage then %d)" %
      (ms_all, B, block, dense, int(o["n_active"].max().item()), torch.bincount(o["status"]).tolist(), ms_rest, int(big.sum()),
       pl.last_handover_count()))

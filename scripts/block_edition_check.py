"""Coupled batch at weight_R = 1e-8 and three times the walking speed (working sets up to ~136
rows): which stage solved how many instances, and how the results compare with the CPU restatement
(status, iteration count, ordered working set, trajectories)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mpc_amd import batches, planner as P  # noqa: E402
from oracle import oracle as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
N, steps = 200, 20
bt = batches.coupled_batch(B)
bt["speed"] = bt["speed"] * scale
pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=R, weight_Q=1.0, number_of_steps=steps, active_stride=192)
o = {k: v.cpu().numpy() for k, v in pl.solve_coupled(**bt).items()}
print("wide %d, block %d, dense %d; working sets up to %d rows; status %s" %
      (pl.last_wide_count(), pl.last_block_count(), pl.last_handover_count(), o["n_active"].max(),
       np.bincount(o["status"]).tolist()), flush=True)
prep = pl.prepare_coupled(**bt)
out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
for _ in range(3):
    pl.solve_coupled_prepared(prep, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    pl.solve_coupled_prepared(prep, out=out)
torch.cuda.synchronize()
print("%.3f ms per solve of %d instances" % ((time.perf_counter() - t) / 10 * 1e3, B), flush=True)
ref = O.coupled_batch(O.params(N, R=R, Q=1.0, height=0.5, iter_cap=8000), O.COUPLED_VEL, steps, 1000, bt["x0"], bt["y0"],
                      bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"], active_stride=192, nthreads=16)
big = np.nonzero(ref["n_active"] > 60)[0]
print("instances beyond 60 rows: %d" % len(big))
bad = 0
worst = 0.0
for i in range(B):
    na = ref["n_active"][i]
    same = (o["status"][i] == ref["status"][i] and o["iters"][i] == ref["iters"][i] and o["n_active"][i] == na and
            list(o["active_idx"][i, :na]) == list(ref["active"][i, :na]))
    if not same:
        bad += 1
        if bad <= 5:
            print("  instance %d: status %d/%d iters %d/%d rows %d/%d" % (i, o["status"][i], ref["status"][i], o["iters"][i],
                                                                      ref["iters"][i], o["n_active"][i], na))
        continue
    for k in ("com_x", "com_y", "zmp_x", "zmp_y", "jerk_x", "jerk_y"):
        e = np.abs(o[k][i] - ref[k][i]).max() / max(1e-300, np.abs(ref[k][i]).max())
        worst = max(worst, e)
    if not np.isclose(o["alg_flops"][i], ref["falg"][i], rtol=1e-12):
        print("  instance %d: alg_flops %r against %r" % (i, o["alg_flops"][i], ref["falg"][i]))
print("paths differing from the CPU restatement: %d of %d; largest relative trajectory difference on the others %.2e; "
      "cost %.2e" % (bad, B, worst, np.abs(o["cost"] - ref["cost"]).max() / np.abs(ref["cost"]).max()))

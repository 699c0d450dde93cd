import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from mpc_amd import batches, planner as P

def rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())

def run(B, N, steps, R, Qs, rw, mode=0):
    bt = batches.coupled_batch(B)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_REF, weight_R=R, weight_Q=1.0, weight_Qs=Qs, replanning_window=rw, number_of_steps=steps, solver_mode=mode)
    out = pl.solve_coupled(**bt); torch.cuda.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    p = O.params(N, R=R, Q=1, Qs=Qs, height=0.5, dense=0, iter_cap=4000)
    ref = O.coupled_batch(p, O.COUPLED_REF, steps, rw, bt['x0'], bt['y0'], bt['speed'], bt['feet_x'], bt['feet_y'], bt['swing0'], nthreads=16)
    same = [(list(o['active_idx'][i, :o['n_active'][i]]) == list(ref['active'][i, :ref['n_active'][i]])) for i in range(B)]
    print("REF N=%d Qs=%g mode=%d: status" % (N, Qs, mode), np.bincount(o['status']), np.bincount(ref['status']), "n_active mean %.1f" % ref['n_active'].mean(),
          "iters eq", (o['iters'] == ref['iters']).mean(), "active identical", np.mean(same))
    print("   ", " ".join("%s %.1e" % (k, max(rel(o[k][i], ref[k][i]) for i in range(B))) for k in ("jerk_x", "jerk_y", "com_x", "com_y", "zmp_x")),
          "cost %.1e" % (np.abs(o['cost'] - ref['cost']) / np.abs(ref['cost'])).max(), "margin %.1e" % np.abs(o['min_margin'] - ref['viol_knot']).max())

run(32, 200, 20, 0.01, 100.0, 1000)
run(32, 200, 20, 0.01, 0.1, 1000)
run(32, 60, 6, 0.01, 1.0, 10)
run(8, 60, 6, 0.01, 1.0, 10, mode=1)

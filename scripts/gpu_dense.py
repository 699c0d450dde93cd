"""Dense-kernel-vs-oracle comparison used during development (not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from mpc_amd import batches, planner as P

def rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())

def box(B=64, mode=1):
    N = 100
    bt = batches.decoupled_batch(B)
    pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0, solver_mode=mode)
    p = O.params(N, R=1.0, height=0.5, iter_cap=3000)
    ref = O.box_batch(p, bt['x0'], bt['y0'], bt['xmin'], bt['xmax'], bt['ymin'], bt['ymax'], nthreads=16, active_stride=64)
    for ax, (s0, lo, hi, jr) in enumerate(((bt['x0'], bt['xmin'], bt['xmax'], ref['jerk_x']), (bt['y0'], bt['ymin'], bt['ymax'], ref['jerk_y']))):
        t = time.time()
        out = pl.solve_box(s0, lo, hi)
        torch.cuda.synchronize()
        print("mode", mode, "axis", ax, "time %.4f" % (time.time() - t))
        o = {k: v.cpu().numpy() for k, v in out.items()}
        print(" status gpu", np.bincount(o['status'], minlength=4), "ref", np.bincount(ref['status'][:, ax], minlength=4), "agree", (o['status'] == ref['status'][:, ax]).mean())
        ok = (ref['status'][:, ax] == 0) & (o['status'] == 0)
        print(" iters equal", (o['iters'] == ref['iters'][:, ax]).mean(), "n_active equal", (o['n_active'] == ref['n_active'][:, ax]).mean())
        same = [(list(o['active_idx'][i, :min(64, o['n_active'][i])]) == list(ref['active'][i, ax, :min(64, ref['n_active'][i, ax])])) for i in range(B)]
        print(" active identical", np.mean(same), " jerk max rel err %.2e" % max(rel(o['jerk_x'][i], jr[i]) for i in np.nonzero(ok)[0]),
              "cost rel err %.2e" % (np.abs(o['cost'][ok] - ref['cost'][ok, ax]) / np.maximum(1e-300, np.abs(ref['cost'][ok, ax]))).max(),
              "falg agree", np.allclose(o['alg_flops'], ref['falg'][:, ax], rtol=1e-12))

def coupled(B=16, mode=1, form=P.COUPLED_VEL, oform=O.COUPLED_VEL, N=200, steps=20):
    bt = batches.coupled_batch(B)
    pl = P.BatchedPlanner(N, formulation=form, weight_R=0.01, weight_Q=1.0, number_of_steps=steps, solver_mode=mode)
    t = time.time()
    out = pl.solve_coupled(**bt)
    torch.cuda.synchronize()
    print("coupled mode", mode, "time %.4f" % (time.time() - t))
    o = {k: v.cpu().numpy() for k, v in out.items()}
    p = O.params(N, R=0.01, Q=1, Qs=100, height=0.5, dense=0, iter_cap=4000)
    ref = O.coupled_batch(p, oform, steps, 1000, bt['x0'], bt['y0'], bt['speed'], bt['feet_x'], bt['feet_y'], bt['swing0'], nthreads=16)
    print(" status gpu", np.bincount(o['status']), "ref", np.bincount(ref['status']))
    print(" iters equal", (o['iters'] == ref['iters']).mean(), "n_active equal", (o['n_active'] == ref['n_active']).mean())
    same = [(list(o['active_idx'][i, :o['n_active'][i]]) == list(ref['active'][i, :ref['n_active'][i]])) for i in range(B)]
    print(" active sets identical:", np.mean(same))
    for k in ("jerk_x", "jerk_y", "com_x", "com_y", "zmp_x", "zmp_y"):
        print(" ", k, "max rel err %.2e" % max(rel(o[k][i], ref[k][i]) for i in range(B)))
    print(" min_margin max abs err %.2e" % np.abs(o['min_margin'] - ref['viol_knot']).max(), "cost max rel err %.2e" % (np.abs(o['cost'] - ref['cost']) / np.abs(ref['cost'])).max(), "falg", np.allclose(o['alg_flops'], ref['falg'], rtol=1e-12))

if __name__ == "__main__":
    box(64, 1)
    box(256, 0)
    coupled(16, 1, N=60, steps=6)
    coupled(16, 1)

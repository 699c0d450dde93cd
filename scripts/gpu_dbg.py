import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from mpc_amd import batches, planner as P
B, N = 128, 100
bt = batches.decoupled_batch(B)
pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0)
p = O.params(N, R=1.0, height=0.5, iter_cap=3000)
for ax, (s0, lo, hi) in enumerate(((bt["x0"], bt["xmin"], bt["xmax"]), (bt["y0"], bt["ymin"], bt["ymax"]))):
    o = {k: v.cpu().numpy() for k, v in pl.solve_box(s0, lo, hi).items()}
    ok = o["status"] == 0
    viol = np.minimum(o["zmp_x"] - lo, hi - o["zmp_x"])
    worst = viol[ok].min(axis=1)
    i = np.nonzero(ok)[0][worst.argmin()]
    print("axis", ax, "worst violation", worst.min(), "inst", i, "margin out", o["min_margin"][i].min())
    zi = O.trajectory(p, 3, s0[i], o["jerk_x"][i])
    print("   zmp(out) vs zmp(integrated jerk) max diff", np.abs(zi - o["zmp_x"][i]).max(), "viol of integrated", np.minimum(zi - lo[i], hi[i] - zi).min())

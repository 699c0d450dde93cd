"""Full-size decoupled batch (4,096 X/Y pairs) on the GPU against the CPU restatement:
status agreement (the numerically infeasible draws end chaotic ~200-iteration paths, so a
handful may legitimately differ), and for the feasible instances identical iteration counts,
ordered working sets and jerks to 1e-6."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from mpc_amd import batches, planner as P  # noqa: E402
from oracle import oracle as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = 100
bt = batches.decoupled_batch(B)
t = time.time()
ref = O.box_batch(O.params(N, R=1.0, height=0.5, iter_cap=3000), bt["x0"], bt["y0"], bt["xmin"], bt["xmax"],
                  bt["ymin"], bt["ymax"], nthreads=16)
print("cpu restatement: %.1f s" % (time.time() - t))
pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0)
for ax, (s0, lo, hi, jr) in enumerate(((bt["x0"], bt["xmin"], bt["xmax"], ref["jerk_x"]),
                                       (bt["y0"], bt["ymin"], bt["ymax"], ref["jerk_y"]))):
    o = {k: v.cpu().numpy() for k, v in pl.solve_box(s0, lo, hi).items()}
    handed = pl.last_handover_count()
    st = ref["status"][:, ax]
    flips = np.nonzero(o["status"] != st)[0]
    ok = (st == 0) & (o["status"] == 0)
    bad_it = np.nonzero(ok & (o["iters"] != ref["iters"][:, ax]))[0]
    bad_ws, worst = [], 0.0
    for i in np.nonzero(ok)[0]:
        na = ref["n_active"][i, ax]
        if na <= 64 and list(o["active_idx"][i, :min(na, 64)]) != list(ref["active"][i, ax, :na]):
            bad_ws.append(int(i))
        worst = max(worst, np.abs(o["jerk_x"][i] - jr[i]).max() / max(1e-300, np.abs(jr[i]).max()))
    print("axis %d: handed over %d, infeasible (cpu) %d, status differs on %s, feasible: iteration count differs "
          "on %s, working set differs on %s, worst jerk error %.2e" %
          (ax, handed, int((st != 0).sum()), flips.tolist(), bad_it.tolist(), bad_ws, worst))

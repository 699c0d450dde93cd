"""Index-order solves of the same instances arranged in different orders in the batch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_amd import batches, planner as P
def timed(call, n=40):
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 8192
bt = batches.coupled_batch(B)
for wl in ("vel", "jerk"):
    pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL if wl == "vel" else P.COUPLED_JERK,
                          weight_R=0.01, weight_Q=1.0, number_of_steps=20)
    def run(perm):
        b = {k: v[perm] for k, v in bt.items()}
        prep = pl.prepare_coupled(**b)
        t = timed(lambda: pl.solve_coupled_prepared(prep))
        out = pl.solve_coupled_prepared(prep); torch.cuda.synchronize()
        return t, out["iters"].cpu().numpy()
    ident = np.arange(B)
    t0, it = run(ident)
    sw = bt["swing0"]
    res = {"identity": t0}
    res["by swing0 (1,0,3,2)"] = run(np.argsort(np.array([1, 0, 3, 2])[sw], kind="stable"))[0]
    res["by swing0 (2,3,0,1)"] = run(np.argsort(np.array([2, 3, 0, 1])[sw], kind="stable"))[0]
    res["by swing0 interleaved within class random"] = run(np.argsort(sw + 0.5 * np.random.default_rng(1).random(B)))[0]
    res["iters desc"] = run(np.argsort(-it, kind="stable"))[0]
    res["swing0 then iters desc"] = run(np.lexsort((-it, np.array([1, 0, 3, 2])[sw])))[0]
    # same iteration counts grouped, but groups in random order
    rng = np.random.default_rng(0)
    key = rng.permutation(it.max() + 1)[it]
    res["iters grouped, groups shuffled"] = run(np.argsort(key, kind="stable"))[0]
    res["random"] = run(rng.permutation(B))[0]
    print(wl, "  ".join("%s %.4f" % kv for kv in res.items()))

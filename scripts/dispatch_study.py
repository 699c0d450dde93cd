"""Dumps per-instance cycle counts and iteration counts of the wave-per-QP kernel (needs the
-DMPCB_STAMPS build) for an offline study of dispatch orders."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MPC_AMD_LIB"] = os.path.join(ROOT, "mpc_amd", "libmpc_amd_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_amd import batches, planner as P
res = {}
for wl in ("vel", "jerk"):
    B = 8192
    bt = batches.coupled_batch(B)
    pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL if wl == "vel" else P.COUPLED_JERK,
                          weight_R=0.01, weight_Q=1.0, number_of_steps=20)
    prep = pl.prepare_coupled(**bt)
    dbg = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
    lib = P.load_library()
    lib.mpcb_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
    lib.mpcb_debug_set_stamp_buffer(pl._h, dbg.data_ptr())
    for _ in range(3):
        out = pl.solve_coupled_prepared(prep)
    torch.cuda.synchronize()
    res[wl + "_cycles"] = dbg.cpu().numpy().sum(axis=1)
    res[wl + "_iters"] = out["iters"].cpu().numpy()
    res[wl + "_nact"] = out["n_active"].cpu().numpy()
import heapq
def makespan(cost, slots=2048):
    """Greedy list scheduling in the given order: what the hardware dispatcher does."""
    h = [0.0] * slots
    for c in cost:
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)
inputs = batches.coupled_batch(8192)
for wl in ("vel", "jerk"):
    c, it = res[wl + "_cycles"].astype(float), res[wl + "_iters"]
    ideal = c.sum() / 2048
    print("%s: makespan / (total cycles / 2048 wave slots), list scheduling of the measured per-instance cycles" % wl)
    print("   index order %.3f   longest first by cycles %.3f   by iteration count %.3f   shortest first %.3f" % (
        makespan(c) / ideal, makespan(c[np.argsort(-c)]) / ideal, makespan(c[np.argsort(-it, kind="stable")]) / ideal,
        makespan(c[np.argsort(it, kind="stable")]) / ideal))
    print("   correlation of cycles with iterations %.3f, with the final working-set size %.3f" % (
        np.corrcoef(c, it)[0, 1], np.corrcoef(c, res[wl + "_nact"])[0, 1]))
    sw = inputs["swing0"]
    print("   by first swing leg: " + "  ".join("leg %d: %.1f iterations, %.0f cycles" % (
        s, it[sw == s].mean(), c[sw == s].mean()) for s in range(4)))
    for k in ("speed", "x0", "y0"):
        for j in range(2):
            print("   correlation of iterations with %s[%d]: %.3f" % (k, j, np.corrcoef(inputs[k][:, j], it)[0, 1]))
np.savez(os.path.join(ROOT, "gpurun_out", "dispatch_study.npz"), **res)

"""Per-section cycle breakdown of the wave-per-QP kernel (needs the -DMPCB_STAMPS build)."""
import sys, os, ctypes as C
os.environ["MPC_AMD_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mpc_amd", "libmpc_amd_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_amd import batches, planner as P
names = ["prologue", "l1 rows", "s eval+psi", "select", "g gather", "d,r", "zn+t1", "add writes (+drop)", "zmp/margin out", "t2,t", "f,u,rho", "prologue: inputs+polygons", "scalars/active out", "jerk out", "com,vel out", "prologue: factor init"]
def makespan(cost, slots):
    """Greedy list scheduling in the given order: what the hardware dispatcher does."""
    import heapq
    h = [0.0] * slots
    for c in cost:
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)
def run(wl, B):
    if wl in ("coupled", "jerk"):
        bt = batches.coupled_batch(B)
        pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL if wl == "coupled" else P.COUPLED_JERK,
                              weight_R=0.01, weight_Q=1.0, number_of_steps=20)
        prep = pl.prepare_coupled(**bt); call = lambda: pl.solve_coupled_prepared(prep)
    else:
        bt = batches.decoupled_batch(B)
        pl = P.BatchedPlanner(100, formulation=P.BOX_1D, weight_R=1.0)
        prep = pl.prepare_box(bt["y0"], bt["ymin"], bt["ymax"]); call = lambda: pl.solve_box_prepared(prep)
    dbg = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
    lib = P.load_library()
    lib.mpcb_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
    lib.mpcb_debug_set_stamp_buffer(pl._h, dbg.data_ptr())
    for _ in range(3):
        out = call()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    it = out["iters"].cpu().numpy()
    tot = d.sum(axis=1)
    print(wl, "B", B, "mean cycles/solve %.0f  mean iters %.1f" % (tot.mean(), it.mean()))
    q = np.percentile(tot, [50, 90, 99, 100])
    print("  cycles/solve percentiles 50/90/99/100: %.0f %.0f %.0f %.0f; iters max %d; simulated makespan on 3072 slots"
          " in dispatch order: %.0f cycles vs ideal %.0f" % (q[0], q[1], q[2], q[3], it.max(), makespan(tot, 3072), tot.sum() / 3072))
    for i, n in enumerate(names[:16]):
        print("  %-14s %9.0f cycles  %5.1f%%" % (n, d[:, i].mean(), 100 * d[:, i].sum() / tot.sum()))
run("coupled", 8192)
run("jerk", 8192)
run("box", 4096)

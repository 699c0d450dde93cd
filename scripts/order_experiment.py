"""Times solves with the first stage dispatched in index order and longest first."""
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
def report(name, pl, call):
    out = call(); torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in out.items()}
    t0 = timed(call)
    pl.set_dispatch_hint(P.DISPATCH_LONGEST_FIRST)
    t1 = timed(call)
    out = call(); torch.cuda.synchronize()
    same = all(torch.equal(out[k], ref[k]) for k in ref)
    pl.set_dispatch_hint(P.DISPATCH_IN_ORDER)
    t2 = timed(call)
    print("%-10s index order %.4f ms  longest first %.4f ms  index order again %.4f  outputs identical %s" % (name, t0, t1, t2, same))
for wl in ("vel", "jerk"):
    B = 8192
    bt = batches.coupled_batch(B)
    pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL if wl == "vel" else P.COUPLED_JERK,
                          weight_R=0.01, weight_Q=1.0, number_of_steps=20)
    prep = pl.prepare_coupled(**bt)
    report(wl, pl, lambda: pl.solve_coupled_prepared(prep))
bt = batches.decoupled_batch(4096)
pb = P.BatchedPlanner(100, formulation=P.BOX_1D, weight_R=1.0)
px = pb.prepare_box(bt["x0"], bt["xmin"], bt["xmax"]); py = pb.prepare_box(bt["y0"], bt["ymin"], bt["ymax"])
report("box x", pb, lambda: pb.solve_box_prepared(px))
report("box y", pb, lambda: pb.solve_box_prepared(py))

"""Stage-to-stage correlation of iteration counts in the replanning sweep."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, heapq
from mpc_amd import batches, planner as P
B, N = 16384, 200
inputs = batches.coupled_batch(B)
dist = batches.replanning_disturbance(B)
pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, replanning_window=10,
                      number_of_steps=20, hip_offsets=batches.NOMINAL_HIP_OFFSETS)
prepared = pl.prepare_replanning(disturbance=dist, **inputs)
out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
its = []
for s in range(8):
    pl.replan_sweep(prepared, 1, out=out, record=False)
    torch.cuda.synchronize()
    its.append(out[0]["iters"].cpu().numpy().copy())
its = np.array(its).astype(float)
def makespan(cost, slots=2048):
    h = [0.0] * slots
    for c in cost: heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)
for s in range(8):
    c = its[s]
    line = "stage %d iters mean %.1f max %d  in-order makespan/ideal %.3f  LPT(own) %.3f" % (
        s, c.mean(), c.max(), makespan(c) / (c.sum() / 2048), makespan(c[np.argsort(-c)]) / (c.sum() / 2048))
    if s:
        line += "  corr with previous stage %.3f  LPT(previous) %.3f" % (
            np.corrcoef(its[s - 1], c)[0, 1], makespan(c[np.argsort(-its[s - 1])]) / (c.sum() / 2048))
    for lag in (2, 3, 4):
        if s >= lag:
            line += "  corr lag %d: %.3f" % (lag, np.corrcoef(its[s - lag], c)[0, 1])
    print(line)

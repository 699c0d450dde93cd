"""Receding-horizon sweep of the benchmark (16,384 rollouts x 8 stages) as `parts` sub-batches with a
handle and a stream each, no cross-stream waits: ms per sweep."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mpc_amd import batches, planner as P
B, N, steps, STAGES = 16384, 200, 20, 8
inp = batches.coupled_batch(B)
dist = batches.replanning_disturbance(B)
hip = batches.NOMINAL_HIP_OFFSETS
for parts in (1, 2, 3):
    per = B // parts
    pls, prep, prist, outs, sts = [], [], [], [], []
    for p in range(parts):
        sl = slice(p * per, (p + 1) * per)
        pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, replanning_window=N // steps,
                              number_of_steps=steps, hip_offsets=hip)
        pr = pl.prepare_replanning(disturbance=dist[sl], **{k: v[sl] for k, v in inp.items()})
        pls.append(pl); prep.append(pr)
        prist.append({k: pr[2][k].clone() for k in ("x0", "y0", "feet_x", "feet_y", "swing0")})
        outs.append(pl._alloc_outputs(per, P.ALL_OUTPUTS))
        sts.append(torch.cuda.Stream())
    def sweep():
        for p in range(parts):
            with torch.cuda.stream(sts[p]):
                for k, t in prist[p].items():
                    prep[p][2][k].copy_(t)
                pls[p].replan_sweep(prep[p], STAGES, out=outs[p])
    for _ in range(2):
        sweep()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(6):
        sweep()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 6 * 1e3
    print("parts %d: %.2f ms per sweep, %.3g solves/s" % (parts, ms, parts * per * STAGES / ms * 1e3), flush=True)

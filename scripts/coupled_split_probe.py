"""Headline batch (8,192 coupled instances, velocity form) as `parts` sub-batches with a handle and a
stream each, free-running (no waits between the streams), `steps` solves each: solves per second.
Optional argument: the dispatch hint of every handle (1 = longest first from the previous solve)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mpc_amd import batches, planner as P
B, N, steps = 8192, 200, 20
inp = batches.coupled_batch(B)
K = 100
HINT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for parts in (1, 2, 4, 1, 2):
    per = B // parts
    pls, prep, outs, sts = [], [], [], []
    for p in range(parts):
        sl = slice(p * per, (p + 1) * per)
        pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
        pl.set_dispatch_hint(HINT)
        pls.append(pl)
        prep.append(pl.prepare_coupled(**{k: v[sl] for k, v in inp.items()}))
        outs.append(pl._alloc_outputs(per, P.ALL_OUTPUTS))
        sts.append(torch.cuda.Stream())
    def run(k):
        for _ in range(k):
            for p in range(parts):
                pls[p].solve_coupled_prepared(prep[p], out=outs[p], stream=sts[p])
    run(300)  # settle
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("parts %d: %.4f ms per step, %.4g solves/s" % (parts, dt / K * 1e3, B * K / dt), flush=True)

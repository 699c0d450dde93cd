"""Where the host-side time of a short timed region goes (fresh process): per-call launch cost,
GPU start lag and the wake-up of the final synchronisation."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mpc_amd import batches, planner as P
B = 8192
bt = batches.coupled_batch(B)
pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=20)
prep = pl.prepare_coupled(**bt)
out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
for _ in range(5):
    pl.solve_coupled_prepared(prep, out=out)
torch.cuda.synchronize()
for trial in range(4):
    K = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    calls = []
    for i in range(K):
        a = time.perf_counter()
        pl.solve_coupled_prepared(prep, out=out)
        ev[i + 1].record()
        calls.append(time.perf_counter() - a)
    t_launched = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dev = ev[0].elapsed_time(ev[K])
    steps = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
    print("trial %d: wall %.3f ms, device span %.3f ms, difference %.0f us; all launches issued after %.0f us; "
          "call cost first %.0f us, median %.0f us; device time of step 1 %.0f us, 2 %.0f us, median %.0f us"
          % (trial, (t1 - t0) * 1e3, dev, ((t1 - t0) * 1e3 - dev) * 1e3, (t_launched - t0) * 1e6, calls[0] * 1e6,
             sorted(calls)[K // 2] * 1e6, steps[0] * 1e3, steps[1] * 1e3, sorted(steps)[K // 2] * 1e3))

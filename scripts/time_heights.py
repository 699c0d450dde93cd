"""Headline batch with a CoM height per instance (three table parts combined per entry as they are
read) against the same batch at the handle's height: ms per solve."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mpc_amd import batches, planner as P  # noqa: E402

B = 8192
bt = batches.coupled_batch(B)
height = 0.4 + 0.2 * batches.SplitMix(np.arange(B), seed=7).uniform()
for form in ("COUPLED_VEL", "COUPLED_JERK"):
    pl = P.BatchedPlanner(200, formulation=getattr(P, form), weight_R=0.01, weight_Q=1.0, number_of_steps=20)
    for name, kw in (("handle's height", {}), ("height per instance", dict(height=height))):
        prep = pl.prepare_coupled(**bt, **kw)
        out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
        for _ in range(20):
            pl.solve_coupled_prepared(prep, out=out)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            pl.solve_coupled_prepared(prep, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 50 * 1e3
        print("%s, %s: %.4f ms per solve of %d instances (%.3g solves/s), iterations mean %.2f" %
              (form, name, ms, B, B / ms * 1e3, out[0]["iters"].double().mean().item()), flush=True)

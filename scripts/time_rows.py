"""Headline batch with caller-built polygon rows (rows_pqr, 700 x 3 doubles per instance read from
memory by the kernel) and with the caller's polygon of every footstep phase (phase_pqr, 41 x 12
doubles per instance copied into the kernel's per-phase table) against the same batch with the rows
built in the kernel: ms per solve."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mpc_amd import batches, planner as P  # noqa: E402
from oracle import oracle as O  # noqa: E402

B, N, steps = 8192, 200, 20
bt = batches.coupled_batch(B)
pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
rows = np.zeros((B, pl.num_constraints, 3))
kn = None
for i in range(B):
    st = O.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i], bt["swing0"][i])
    A, b = st["A"], st["b"]
    if kn is None:
        kn = np.array([np.nonzero(A[r])[0][0] % N for r in range(A.shape[0])])
    idx = np.arange(A.shape[0])
    rows[i, :, 0], rows[i, :, 1], rows[i, :, 2] = A[idx, kn], A[idx, kn + N], b
pd = (N // steps) // 2
phases = np.zeros((B, 2 * steps + 1, 4, 3))
phases[:, :, :, 2] = np.inf
for ph in range(2 * steps):
    k0 = (ph // 2) * 2 * pd + (ph % 2) * pd
    r0 = int(np.nonzero(kn == k0)[0][0])
    E = 3 if ph % 2 else 4
    phases[:, ph, :E] = rows[:, r0:r0 + E]
for name, prep in (("rows built in the kernel", pl.prepare_coupled(**bt)),
                   ("caller-built rows", pl.prepare_coupled(bt["x0"], bt["y0"], bt["speed"], rows_pqr=rows)),
                   ("caller-built polygon per phase", pl.prepare_coupled(bt["x0"], bt["y0"], bt["speed"],
                                                                          phase_pqr=phases))):
    out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
    for _ in range(20):
        pl.solve_coupled_prepared(prep, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        pl.solve_coupled_prepared(prep, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 50 * 1e3
    print("%s: %.4f ms per solve of %d instances (%.3g solves/s), iterations mean %.2f" %
          (name, ms, B, B / ms * 1e3, out[0]["iters"].double().mean().item()), flush=True)

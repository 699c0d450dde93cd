"""X axis of the decoupled benchmark batch alone (no second stream): ms per solve and a checksum
of the iteration counts, for comparing second-stage variants (MPCB_DEBUG_* environment)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mpc_amd import batches, planner as P  # noqa: E402

B, N = 4096, 100
i = batches.decoupled_batch(B, N=N, steps=10)
pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0)
px = pl.prepare_box(i["x0"], i["xmin"], i["xmax"])
out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
for _ in range(5):
    pl.solve_box_prepared(px, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    pl.solve_box_prepared(px, out=out)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / 20 * 1e3
o = out[0]
print("%.3f ms per solve; handed over %d; iters sum %d; status counts %s; cost sum %.12e" %
      (ms, pl.last_handover_count(), int(o["iters"].sum().item()), torch.bincount(o["status"]).tolist(),
       float(o["cost"][o["status"] == 0].sum().item())))

"""A/B harness for changes to the box second stage that must not change a single bit: X axis of
the decoupled benchmark batch, milliseconds per solve (median of several timed runs), the
longest outer-iteration counts and a digest of every output.  Two builds whose arithmetic is the
same print the same digest and iteration counts."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
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
ms = []
for rep in range(7):
    t = time.perf_counter()
    for _ in range(10):
        pl.solve_box_prepared(px, out=out)
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t) / 10 * 1e3)
h = hashlib.sha256()
for k in sorted(out[0]):
    h.update(out[0][k].cpu().numpy().tobytes())
it = out[0]["iters"].cpu().numpy()
top = np.sort(it)[::-1][:4]
print("ms per solve: median %.3f min %.3f; handed over %d; longest %s; digest %s" %
      (float(np.median(ms)), min(ms), pl.last_handover_count(), top.tolist(), h.hexdigest()[:16]))

"""X axis of the decoupled benchmark batch with the tables built on the device and on the host:
ms per solve, the longest outer-iteration counts and which draws they belong to -- the step of the
decoupled workload lasts as long as its first stage plus its longest second-stage instance, and
the path of a numerically infeasible draw depends on the last bits of the tables."""
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
for mode, name in ((P.TABLES_DEVICE, "device"), (P.TABLES_HOST, "host")):
    pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0, table_build=mode)
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
    it = out[0]["iters"].cpu().numpy()
    st = out[0]["status"].cpu().numpy()
    top = np.argsort(-it)[:8]
    print("%s tables: %.3f ms per solve; handed over %d; status counts %s" %
          (name, ms, pl.last_handover_count(), np.bincount(st).tolist()))
    print("   longest: " + ", ".join("draw %d: %d iters (status %d)" % (d, it[d], st[d]) for d in top))
    if name == "device":
        S_dev = pl.matrix(0)
    else:
        S_host = pl.matrix(0)
d = np.abs(S_dev - S_host)
print("S tables differ in %d of %d entries, at most %.1f ulp" % ((d > 0).sum(), d.size, (d / np.spacing(np.abs(S_host))).max()))

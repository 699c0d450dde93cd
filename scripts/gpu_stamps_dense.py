"""Per-section cycle breakdown of the dense second-stage kernel (needs the -DMPCB_STAMPS build)."""
import sys, os, ctypes as C
# (another stamped build: MPC_AMD_LIB; argument: batch size, default 512)
os.environ.setdefault("MPC_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mpc_amd", "libmpc_amd_stamps.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_amd import batches, planner as P
names = ["setup, exit", "zmp (E)", "select", "d partial (A)", "A barrier", "z,r partial (B)", "B barrier", "step lengths (C)", "add coefs", "add J pass", "drop: row, coefs", "row ip again", "drop: barrier", "(passes)", "(iters)", "drop: J sweep"]
SECT = [i for i in range(16) if i not in (13, 14)]  # (times are wave 0's)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bt = batches.decoupled_batch(B)
pl = P.BatchedPlanner(100, formulation=P.BOX_1D, weight_R=1.0)
prep = pl.prepare_box(bt["x0"], bt["xmin"], bt["xmax"])
dbg = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
lib = P.load_library()
lib.mpcb_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
lib.mpcb_debug_set_stamp_buffer(pl._h, dbg.data_ptr())
out = pl.solve_box_prepared(prep); torch.cuda.synchronize()
st = out["status"].cpu().numpy(); d = dbg.cpu().numpy().astype(np.float64)
bad = st == 1
print("infeasible", bad.sum(), "iters mean", out["iters"].cpu().numpy()[bad].mean())
tot = d[bad][:, SECT].sum(axis=1)
print("mean cycles per infeasible instance (dense kernel, wave-kernel stamps overwritten): %.0f" % tot.mean())
for i, n in enumerate(names):
    print("  %-14s %10.0f  %5.1f%%" % (n, d[bad][:, i].mean(), 100 * d[bad][:, i].sum() / tot.sum()))
print("passes mean %.0f max %.0f, outer iterations mean %.0f; cycles per pass %.0f; max cycles %.3g" % (
    d[bad][:, 13].mean(), d[bad][:, 13].max(), d[bad][:, 14].mean(), tot.sum() / d[bad][:, 13].sum(), tot.max()))
worst = int(np.argmax(np.where(bad, d[:, SECT].sum(axis=1), 0)))
print("longest instance: draw %d, %d outer iterations, %d passes, %.4g cycles" % (worst, d[worst, 14], d[worst, 13], d[worst, SECT].sum()))
for i, n in enumerate(names):
    if i < 13 or i == 15:
        print("  %-14s %10.0f  %5.1f%%" % (n, d[worst, i], 100 * d[worst, i] / d[worst, SECT].sum()))

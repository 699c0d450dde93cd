"""Per-section cycle breakdown of the dense second-stage kernel (needs the -DMPCB_STAMPS build)."""
import sys, os, ctypes as C
os.environ["MPC_AMD_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mpc_amd", "libmpc_amd_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_amd import batches, planner as P
names = ["setup", "zmp (E)", "select", "d partial (A)", "A barrier", "z,r partial (B)", "B barrier", "step lengths (C)", "add coefs", "add J pass", "drop", "row ip again", "(exit)", "(passes)", "(iters)", "-"]
B = 512
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
tot = d[bad][:, :13].sum(axis=1)
print("mean cycles per infeasible instance (dense kernel, wave-kernel stamps overwritten): %.0f" % tot.mean())
for i, n in enumerate(names):
    print("  %-14s %10.0f  %5.1f%%" % (n, d[bad][:, i].mean(), 100 * d[bad][:, i].sum() / tot.sum()))
print("passes mean %.0f max %.0f, outer iterations mean %.0f; cycles per pass %.0f; max cycles %.3g" % (
    d[bad][:, 13].mean(), d[bad][:, 13].max(), d[bad][:, 14].mean(), tot.sum() / d[bad][:, 13].sum(), tot.max()))

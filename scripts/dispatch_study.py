"""Dumps per-instance cycle counts and iteration counts of the wave-per-QP kernel (needs the
-DMPCB_STAMPS build) for an offline study of dispatch orders."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MPC_AMD_LIB"] = os.path.join(ROOT, "mpc_amd", "libmpc_amd_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_amd import batches, planner as P
res = {}
for wl in ("vel", "jerk"):
    B = 8192
    bt = batches.coupled_batch(B)
    pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL if wl == "vel" else P.COUPLED_JERK,
                          weight_R=0.01, weight_Q=1.0, number_of_steps=20)
    prep = pl.prepare_coupled(**bt)
    dbg = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
    lib = P.load_library()
    lib.mpcb_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
    lib.mpcb_debug_set_stamp_buffer(pl._h, dbg.data_ptr())
    for _ in range(3):
        out = pl.solve_coupled_prepared(prep)
    torch.cuda.synchronize()
    res[wl + "_cycles"] = dbg.cpu().numpy().sum(axis=1)
    res[wl + "_iters"] = out["iters"].cpu().numpy()
    res[wl + "_nact"] = out["n_active"].cpu().numpy()
np.savez(os.path.join(ROOT, "gpurun_out", "dispatch_study.npz"), **res)
print("saved")

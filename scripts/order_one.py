"""One workload, one dispatch mode (for a kernel trace): order_one.py vel|jerk 0|1"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mpc_amd import batches, planner as P
wl, mode = sys.argv[1], int(sys.argv[2])
bt = batches.coupled_batch(8192)
pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL if wl == "vel" else P.COUPLED_JERK,
                      weight_R=0.01, weight_Q=1.0, number_of_steps=20)
prep = pl.prepare_coupled(**bt)
pl.set_dispatch_hint(mode)
for _ in range(30):
    pl.solve_coupled_prepared(prep)
torch.cuda.synchronize()

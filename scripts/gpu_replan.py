import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from mpc_amd import batches, planner as P
B, N, steps, stages = 64, 200, 20, 8
bt = batches.coupled_batch(B); dist = batches.replanning_disturbance(B)
hip = batches.NOMINAL_HIP_OFFSETS
pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, replanning_window=N // steps, number_of_steps=steps, hip_offsets=hip)
prep = pl.prepare_replanning(disturbance=dist, **bt)
t = time.time()
res, stage, st = pl.replan_sweep(prep, stages)
torch.cuda.synchronize(); print("gpu sweep %.4f s" % (time.time() - t))
p = O.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000)
ref = O.replanning_batch(p, steps, stages, bt['x0'], bt['y0'], bt['speed'], bt['feet_x'], bt['feet_y'], bt['swing0'], dist, hip=hip, nthreads=16)
sc = stage['cost'].cpu().numpy(); ss = stage['status'].cpu().numpy(); sn = stage['n_active'].cpu().numpy()
print("status equal", (ss == ref['stage_status']).all(), "n_active equal", (sn == ref['stage_nactive']).mean(), "cost rel err %.2e" % (np.abs(sc - ref['stage_cost']) / np.abs(ref['stage_cost'])).max())
for k in ("x0", "y0", "feet_x", "feet_y"):
    print(k, "max abs err %.2e" % np.abs(st[k].cpu().numpy() - ref[k]).max())
print("swing equal", (st['swing0'].cpu().numpy() == ref['swing0']).all(), "last jerk rel err %.2e" % (np.abs(res['jerk_x'].cpu().numpy() - ref['jerk_x']).max() / np.abs(ref['jerk_x']).max()))
print("n_active per stage (mean)", sn.mean(axis=1))

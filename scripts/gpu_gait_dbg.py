import os, sys, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_amd import planner as P
from oracle import oracle as O
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
gait, duty = sys.argv[1], sys.argv[2]
r = subprocess.run([os.path.join(root, "tests/cpp/dropin_drivers"), "gait", gait, duty], env=dict(os.environ, HOME=tmp), capture_output=True, text=True)
print(r.stdout[-600:], r.stderr[-300:])
N = 200
rows = np.loadtxt(os.path.join(tmp, "gait_rows.txt")).reshape(-1, 5)
m = len(rows)
A = np.zeros((m, 2 * N)); b = np.zeros(m)
slots = np.zeros((1, 4 * N, 3)); slots[:, :, 2] = np.inf
used = np.zeros(N, dtype=int); slot_row = {}
for rr, k, p, q, c in rows:
    rr, k = int(rr), int(k)
    A[rr, k], A[rr, k + N], b[rr] = p, q, c
    s = 4 * k + used[k]; used[k] += 1
    slots[0, s] = (p, q, c); slot_row[s] = rr
ref = O.solve_coupled(O.params(N, Ts=0.04, R=0.01, Q=1.0, height=0.5, iter_cap=8000), O.COUPLED_VEL, np.zeros(3), np.zeros(3), A, b, np.array([0.05, 0.0]), 1000)
print("oracle status", ref["info"].status, "n_active", ref["info"].n_active, "iters", ref["info"].iters, "f", ref["info"].f)
for mode in (0, 1):
    pl = P.BatchedPlanner(N, Ts=0.04, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=0, active_stride=4 * N, solver_mode=mode)
    o = {k: v.cpu().numpy() for k, v in pl.solve_coupled(np.zeros((1, 3)), np.zeros((1, 3)), np.array([[0.05, 0.0]]), rows_pqr=slots).items()}
    na = o["n_active"][0]
    act = [slot_row[s] for s in o["active_idx"][0, :na]]
    print("mode", mode, "status", o["status"][0], "n_active", na, "iters", o["iters"][0], "cost", o["cost"][0],
          "jerk rel err", np.abs(o["jerk_x"][0] - ref["jerk_x"]).max() / np.abs(ref["jerk_x"]).max(),
          "same active", act == list(ref["active"]), "set equal", set(act) == set(ref["active"]))

"""bench.py's timed region repeated in one fresh process: wall clock against device span."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from mpc_amd import planner as P
wl = bench.CoupledWorkload(P.COUPLED_VEL, 8192, 0, 0)
for _ in range(5):
    wl.step()
for trial in range(5):
    K = 50
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    ta = time.perf_counter()
    for _ in range(K):
        wl.step()
    tb = time.perf_counter()
    ev1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dev = ev0.elapsed_time(ev1)
    print("trial %d: wall %.3f ms (%.4f/step), device span %.3f ms (%.4f/step), difference %.0f us; ev0.record took %.0f us, "
          "launch loop %.0f us" % (trial, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / K, dev, dev / K, ((t1 - t0) * 1e3 - dev) * 1e3,
                                   (ta - t0) * 1e6, (tb - ta) * 1e6))

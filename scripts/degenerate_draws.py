"""Where the GPU path and the CPU restatement part ways on the numerically degenerate draws of the
decoupled batch, and how thin the decision was.

For every X-axis draw of the 4,096-pair batch whose end status differs between the two (and for
the degenerate draws that both solve or both give up on with different iteration counts, up to
--extra of them), the number of step-direction passes after which the two working sets first
differ is found by bisection on the pass cap (both sides stop at the same cap and report the
working set they hold: `iter_cap` of mpcb_params / orc_params), and the CPU restatement's record
of that pass is printed: the quantities each branch of the pass compares, and the smallest
relative margin among them.  A margin at the level of the rounding error of the quantities
compared (they are sums of ~100 products of size 1e2-1e4, after ~200 Givens updates of J:
1e-12 - 1e-9 relative) means the decision is made by rounding, not by the method.

    python scripts/degenerate_draws.py > profiles/r03/degenerate_draws.txt      (needs the GPU)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from mpc_amd import batches, planner as P  # noqa: E402
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--extra", type=int, default=4)
args = ap.parse_args()

B, N = 4096, 100
bt = batches.decoupled_batch(B)
O.build()
ref = O.box_batch(O.params(N, R=1.0, height=0.5, iter_cap=3000), bt["x0"], bt["y0"], bt["xmin"], bt["xmax"],
                  bt["ymin"], bt["ymax"], nthreads=16)
pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0, active_stride=128)
gpu = {k: v.cpu().numpy() for k, v in pl.solve_box(bt["x0"], bt["xmin"], bt["xmax"]).items()}
st_ref, it_ref = ref["status"][:, 0], ref["iters"][:, 0]
deg = it_ref > N
flips = np.nonzero(gpu["status"] != st_ref)[0]
others = [d for d in np.nonzero(deg & (gpu["status"] == st_ref) & (gpu["iters"] != it_ref))[0]][:args.extra]
print("decoupled batch, X axis: %d draws whose reference path runs past N = %d outer iterations; end status differs "
      "on %s; %d more degenerate draws end alike after different numbers of iterations"
      % (int(deg.sum()), N, flips.tolist(), int((deg & (gpu["status"] == st_ref) & (gpu["iters"] != it_ref)).sum())))
print("(of the %d regular draws every one has the reference's status, iteration count and working set)\n"
      % int((~deg).sum()))


def gpu_at_cap(d, cap):
    h = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0, active_stride=128, iter_cap=cap)
    o = {k: v.cpu().numpy() for k, v in h.solve_box(bt["x0"][d:d + 1], bt["xmin"][d:d + 1], bt["xmax"][d:d + 1]).items()}
    h.close()
    na = int(o["n_active"][0])
    return int(o["status"][0]), int(o["iters"][0]), [int(a) for a in o["active_idx"][0, :na]]


def cpu_at_cap(d, cap):
    p = O.params(N, R=1.0, height=0.5, iter_cap=cap)
    _, info, act = O.solve_box(p, bt["x0"][d], bt["xmin"][d], bt["xmax"][d])
    return int(info.status), int(info.iters), [int(a) for a in act]


def margins(row):
    r = dict(zip(O.TRACE_COLS, row))
    out = {}
    if np.isfinite(r["s_runner_up"]):
        out["most violated row vs runner-up"] = abs(r["s_runner_up"] - r["s_ip"]) / max(abs(r["s_ip"]), 1e-300)
    if np.isfinite(r["t1"]) and np.isfinite(r["t2"]):
        out["t1 vs t2 (full or partial step)"] = abs(r["t1"] - r["t2"]) / max(abs(r["t1"]), abs(r["t2"]))
    if np.isfinite(r["t1"]) and np.isfinite(r["t1_runner_up"]):
        out["blocking row vs runner-up (u/r)"] = abs(r["t1_runner_up"] - r["t1"]) / max(abs(r["t1"]), 1e-300)
    if np.isfinite(r["r_min_over_max"]):
        out["smallest |r_k| / largest (sign of r_k decides who may block)"] = r["r_min_over_max"]
    out["z.z vs eps (is there a primal step)"] = abs(np.log10(max(r["zz"], 1e-300) / np.finfo(float).eps))
    out["|d(iq)| vs eps R_norm (degenerate addition)"] = abs(np.log10(max(r["d_iq_over_eps_Rnorm"], 1e-300)))
    return r, out


for d in list(flips) + others:
    full = gpu_at_cap(d, 0), cpu_at_cap(d, 3000)
    lo, hi = 0, 3000  # the working sets agree after `lo` passes and differ after `hi`
    p = O.params(N, R=1.0, height=0.5, iter_cap=3000)
    _, info, _, trace = O.solve_box_trace(p, bt["x0"][d], bt["xmin"][d], bt["xmax"][d])
    hi = int(info.n_l2a) + 5
    if gpu_at_cap(d, hi)[1:] == cpu_at_cap(d, hi)[1:] and full[0][0] == full[1][0]:
        print("draw %d: no difference up to %d passes\n" % (d, hi))
        continue
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if gpu_at_cap(d, mid)[1:] == cpu_at_cap(d, mid)[1:]:
            lo = mid
        else:
            hi = mid
    g, c = gpu_at_cap(d, hi), cpu_at_cap(d, hi)
    print("draw %d: GPU ends with status %d after %d outer iterations, the CPU restatement with status %d after %d "
          "(%d passes)" % (d, gpu["status"][d], gpu["iters"][d], st_ref[d], it_ref[d], int(info.n_l2a)))
    print("   working sets identical through pass %d, different after pass %d (GPU holds %d rows, CPU %d; "
          "first differing position %s)" %
          (lo, hi, len(g[2]), len(c[2]),
           next((i for i, (a, b2) in enumerate(zip(g[2], c[2])) if a != b2), min(len(g[2]), len(c[2])))))
    rows = [r for r in trace if int(r[1]) in (hi - 1, hi, hi + 1)]
    for row in rows:
        r, m = margins(row)
        thin = sorted(m.items(), key=lambda kv: kv[1])[:2]
        print("   CPU pass %d (outer iteration %d, %d rows active, candidate row %d): s_ip %.6e, t1 %.6e, t2 %.6e, z.z %.3e, "
              "z.n %.3e" % (int(r["pass"]), int(r["iter"]), int(r["iq"]), int(r["ip"]), r["s_ip"], r["t1"], r["t2"],
                            r["zz"], r["zn"]))
        print("        thinnest decisions: " + "; ".join("%s: %.1e" % kv for kv in thin))
    print()

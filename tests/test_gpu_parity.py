"""Parity of the HIP path (through the C ABI) with the CPU oracle on identical inputs:
CoM/ZMP/jerk trajectories and objective within 1e-6 relative, identical working sets,
identical status; plus oracle-free properties at full batch size."""
import numpy as np
import pytest

from mpc_amd import batches

pytestmark = pytest.mark.gpu

RTOL = 1e-6  # tolerance stated by the project's north star for trajectories and objective


@pytest.fixture(scope="module")
def P():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from mpc_amd import planner
    planner.load_library()
    return planner


def _np(out):
    return {k: v.cpu().numpy() for k, v in out.items()}


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


def _check_coupled(o, ref, B):
    assert (o["status"] == ref["status"]).all()
    assert (o["iters"] == ref["iters"]).all()
    for i in range(B):
        na = ref["n_active"][i]
        assert o["n_active"][i] == na
        assert list(o["active_idx"][i, :na]) == list(ref["active"][i, :na])
        assert (o["active_idx"][i, na:] == -1).all()
        for k in ("jerk_x", "jerk_y", "com_x", "com_y", "zmp_x", "zmp_y"):
            assert _rel(o[k][i], ref[k][i]) < RTOL, (k, i)
    assert (np.abs(o["cost"] - ref["cost"]) <= RTOL * np.abs(ref["cost"])).all()
    assert np.abs(o["min_margin"] - ref["viol_knot"]).max() < 1e-8
    assert np.allclose(o["alg_flops"], ref["falg"], rtol=1e-12)


@pytest.mark.parametrize("form_name,R", [("COUPLED_VEL", 0.01), ("COUPLED_JERK", 0.01)])
def test_coupled_full_size_vs_oracle(P, oracle, form_name, R):
    B, N, steps = 96, 200, 20
    bt = batches.coupled_batch(B)
    form = getattr(P, form_name)
    pl = P.BatchedPlanner(N, formulation=form, weight_R=R, weight_Q=1.0, number_of_steps=steps)
    o = _np(pl.solve_coupled(**bt))
    ref = oracle.coupled_batch(oracle.params(N, R=R, Q=1.0, height=0.5, iter_cap=4000),
                               getattr(oracle, form_name), steps, 1000, bt["x0"], bt["y0"], bt["speed"],
                               bt["feet_x"], bt["feet_y"], bt["swing0"], nthreads=16)
    _check_coupled(o, ref, B)
    if form_name == "COUPLED_VEL" and R == 0.01:
        assert ref["n_active"].min() >= 1  # this batch is the one where constraints bind


@pytest.mark.parametrize("N,steps,rw", [(60, 6, 1000), (100, 10, 10), (130, 7, 1000), (64, 8, 5)])
def test_coupled_other_shapes_vs_oracle(P, oracle, N, steps, rw):
    """Ragged horizons (trailing 4-stance knots), Gaussian velocity weights, odd step lengths."""
    B = 24
    bt = batches.coupled_batch(B, first=1000)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0,
                          replanning_window=rw, number_of_steps=steps)
    o = _np(pl.solve_coupled(**bt))
    ref = oracle.coupled_batch(oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000), oracle.COUPLED_VEL,
                               steps, rw, bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"],
                               bt["swing0"], nthreads=16)
    _check_coupled(o, ref, B)


@pytest.mark.parametrize("N,steps", [(300, 15), (512, 16), (8, 2), (3, 1)])
def test_coupled_extreme_horizons_vs_oracle(P, oracle, N, steps):
    """Horizons beyond 256 knots (eight knots per lane, up to the maximum of 512) and tiny ones
    (fewer knots than lanes; N=3 has empty swing phases)."""
    B = 6 if N > 256 else 12
    bt = batches.coupled_batch(B, first=3000)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    o = _np(pl.solve_coupled(**bt))
    ref = oracle.coupled_batch(oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=8000), oracle.COUPLED_VEL,
                               steps, 1000, bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"],
                               bt["swing0"], nthreads=16)
    _check_coupled(o, ref, B)


def test_box_long_horizon_vs_oracle(P, oracle):
    """Box problem with 300 knots: eight knots per lane in the first kernel, HBM-resident
    factors in the dense one."""
    B, N = 8, 300
    bt = batches.decoupled_batch(B, N=N, steps=10, first=77)
    pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0)
    ref = oracle.box_batch(oracle.params(N, R=1.0, height=0.5, iter_cap=6000), bt["x0"], bt["y0"], bt["xmin"],
                           bt["xmax"], bt["ymin"], bt["ymax"], nthreads=16)
    o = _np(pl.solve_box(bt["y0"], bt["ymin"], bt["ymax"]))
    assert (o["status"] == ref["status"][:, 1]).all()
    ok = ref["status"][:, 1] == 0
    assert ok.any()
    for i in np.nonzero(ok)[0]:
        na = ref["n_active"][i, 1]
        assert list(o["active_idx"][i, :o["n_active"][i]]) == list(ref["active"][i, 1, :na])
        assert _rel(o["jerk_x"][i], ref["jerk_y"][i]) < RTOL


def test_coupled_com_step_correction(P, oracle):
    """computeSteps with an initial CoM (MPCPlanner.cpp:1231-1238, 1266-1283)."""
    B, N, steps = 16, 200, 20
    bt = batches.coupled_batch(B, first=5000)
    bt["com0"] = np.stack([bt["x0"][:, 0] + 0.01, bt["y0"][:, 0] - 0.01], axis=1)
    hip = batches.NOMINAL_HIP_OFFSETS
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, replanning_window=10,
                          number_of_steps=steps, hip_offsets=hip)
    o = _np(pl.solve_coupled(**bt))
    ref = oracle.coupled_batch(oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000), oracle.COUPLED_VEL,
                               steps, 10, bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"],
                               bt["swing0"], com0=bt["com0"], hip=hip, nthreads=16)
    _check_coupled(o, ref, B)


def _check_box_axis(o, st, iters, nact, active, jerk, cost, N, lo=None, hi=None, flips_allowed=0, zmp_of=None):
    """One axis of a batch of box problems against the CPU restatement (arrays of that axis).

    A draw whose reference path runs past N outer iterations is numerically degenerate: the box is
    reachable in exact arithmetic (Zu is triangular with a nonzero diagonal), the dual method keeps
    adding nearly dependent rows for ~2N iterations and what it reports at the end of that path is
    decided by rounding -- two CPU restatements that differ only in summation order disagree on about
    1 in 300 of them.  Those may differ in status (at most `flips_allowed` of them) and in the length
    and order of their path; a degenerate draw that BOTH sides solve is still the same strictly convex
    QP solved to the same stopping rule, so its objective (RTOL), its ZMP trajectory (`zmp_of(i)`, 1e-6 of
    its largest sample), its jerk (1e-5: the direction the box barely constrains) and the box itself
    are compared.  Every other instance must agree exactly: status, iteration count, ordered working
    set, jerk and objective to RTOL.  Returns the indices whose status differs."""
    degenerate = iters > N
    flips = np.nonzero(o["status"] != st)[0]
    assert not (~degenerate)[flips].any(), ("status differs on a regular instance", flips.tolist())
    assert len(flips) <= flips_allowed, flips.tolist()
    ok = (st == 0) & ~degenerate
    assert (o["iters"][ok] == iters[ok]).all()
    for i in np.nonzero(ok)[0]:
        na = nact[i]
        assert list(o["active_idx"][i, :o["n_active"][i]]) == list(active[i, :na])
        assert _rel(o["jerk_x"][i], jerk[i]) < RTOL
        if lo is not None:
            # the ZMP honours the box up to the solver's stopping rule (sum of violations
            # <= m * eps * trace(G) * trace(L^-T) * 100 = 4.4e-8 here, eiquadprog.hpp:267)
            assert (o["zmp_x"][i] >= lo[i] - 1e-7).all() and (o["zmp_x"][i] <= hi[i] + 1e-7).all()
    assert (np.abs(o["cost"][ok] - cost[ok]) <= RTOL * np.maximum(1e-12, np.abs(cost[ok]))).all()
    both = np.nonzero(degenerate & (st == 0) & (o["status"] == 0))[0]
    sane, blown = [], []
    for i in both:
        # Status 0 at the end of such a path does not always mean a plan: the multipliers of the
        # nearly dependent rows can run away (objectives of 1e30 on both sides) until the stopping
        # rule's sum of violations happens to fall under its threshold.  Only a reference solution
        # that is a plan -- finite-sized and inside its box -- is something to agree with.
        zref = zmp_of(i) if zmp_of is not None else None
        is_plan = abs(cost[i]) < 1e12 and (zref is None or lo is None or
                                          ((zref >= lo[i] - 1e-6).all() and (zref <= hi[i] + 1e-6).all()))
        if not is_plan:
            blown.append(int(i))
            continue
        sane.append(int(i))
        assert abs(o["cost"][i] - cost[i]) <= RTOL * max(1e-12, abs(cost[i])), ("degenerate draw", i)
        assert _rel(o["jerk_x"][i], jerk[i]) < 1e-5, ("degenerate draw", i, _rel(o["jerk_x"][i], jerk[i]))
        if zref is not None:
            assert _rel(o["zmp_x"][i], zref) < RTOL, ("degenerate draw", i)
        if lo is not None:
            assert (o["zmp_x"][i] >= lo[i] - 1e-7).all() and (o["zmp_x"][i] <= hi[i] + 1e-7).all()
    if len(both):
        print("degenerate draws with status 0 on both sides: compared %s (jerk within %.1e, objective within %.1e); "
              "not plans on the reference side (objective >= 1e12 or outside the box), not compared: %s" %
              (sane, max([_rel(o["jerk_x"][i], jerk[i]) for i in sane] or [0.0]),
               max([abs(o["cost"][i] - cost[i]) / max(1e-12, abs(cost[i])) for i in sane] or [0.0]), blown))
    # infeasible draws: zero-filled outputs, infinite cost
    bad = o["status"] == 1
    if bad.any():
        assert np.isinf(o["cost"][bad]).all() and (o["jerk_x"][bad] == 0).all()
    return flips


def test_box_vs_oracle(P, oracle):
    B, N = 128, 100
    bt = batches.decoupled_batch(B)
    pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0)
    ref = oracle.box_batch(oracle.params(N, R=1.0, height=0.5, iter_cap=3000), bt["x0"], bt["y0"], bt["xmin"],
                           bt["xmax"], bt["ymin"], bt["ymax"], nthreads=16)
    for ax, (s0, lo, hi, jr) in enumerate(((bt["x0"], bt["xmin"], bt["xmax"], ref["jerk_x"]),
                                           (bt["y0"], bt["ymin"], bt["ymax"], ref["jerk_y"]))):
        o = _np(pl.solve_box(s0, lo, hi))
        assert (ref["status"][:, ax] == 0).sum() > B // 2
        _check_box_axis(o, ref["status"][:, ax], ref["iters"][:, ax], ref["n_active"][:, ax], ref["active"][:, ax],
                        jr, ref["cost"][:, ax], N, lo, hi, flips_allowed=1)


def test_decoupled_full_batch_vs_oracle(P, oracle):
    """The whole 4,096-pair batch of the decoupled workload: every regular instance identical to the
    CPU restatement; of the ~310 numerically degenerate X-axis draws (see _check_box_axis), which take
    the hand-over from the first kernel to the dense one, at least 97 % end with the same status.
    The dense kernel alone (no hand-over, nothing taken over from the first stage) must agree with the
    two-stage path on every regular instance as well."""
    B, N = 4096, 100
    bt = batches.decoupled_batch(B)
    ref = oracle.box_batch(oracle.params(N, R=1.0, height=0.5, iter_cap=3000), bt["x0"], bt["y0"], bt["xmin"],
                           bt["xmax"], bt["ymin"], bt["ymax"], nthreads=16)
    pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0)
    for ax, (s0, lo, hi, jr) in enumerate(((bt["x0"], bt["xmin"], bt["xmax"], ref["jerk_x"]),
                                           (bt["y0"], bt["ymin"], bt["ymax"], ref["jerk_y"]))):
        o = _np(pl.solve_box(s0, lo, hi))
        n_deg = int((ref["iters"][:, ax] > N).sum())
        op = oracle.params(N, R=1.0, height=0.5)
        flips = _check_box_axis(o, ref["status"][:, ax], ref["iters"][:, ax], ref["n_active"][:, ax],
                                ref["active"][:, ax], jr, ref["cost"][:, ax], N, lo, hi,
                                flips_allowed=max(1, (3 * n_deg) // 100),
                                zmp_of=lambda i: oracle.trajectory(op, 3, s0[i], jr[i]))
        if ax == 0:
            assert n_deg > 250 and pl.last_handover_count() >= n_deg  # this batch exercises the hand-over
        print("axis %d: %d degenerate draws, status differs on %s" % (ax, n_deg, flips.tolist()))
    # second stage alone on a slice that contains degenerate draws
    sl = slice(0, 512)
    pd = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0, solver_mode=1)
    od = _np(pd.solve_box(bt["x0"][sl], bt["xmin"][sl], bt["xmax"][sl]))
    _check_box_axis(od, ref["status"][sl, 0], ref["iters"][sl, 0], ref["n_active"][sl, 0], ref["active"][sl, 0],
                    ref["jerk_x"][sl], ref["cost"][sl, 0], N, flips_allowed=2)


def test_box_hand_over_resumes_regular_instances(P, oracle):
    """Tight corridors around a sinusoidal ZMP path: feasible, well-conditioned box problems whose
    working sets outgrow the first kernel's capacity, so they are handed to the dense kernel, which
    rebuilds its factors for the working set it is given and carries on.  Iteration counts, ordered
    working sets, jerks and objectives must come out as if one solver had run from the start."""
    B, N = 160, 100
    k = np.arange(N)
    rng = batches.SplitMix(np.arange(B), seed=77)
    phase, period = rng.uniform(0.0, 6.28), rng.uniform(25.0, 60.0)
    amp, width = rng.uniform(0.3, 0.6), rng.uniform(0.02, 0.04)
    centre = amp[:, None] * (np.sin(2 * np.pi * k[None, :] / period[:, None] + phase[:, None]) - np.sin(phase)[:, None])
    lo, hi = centre - width[:, None], centre + width[:, None]
    x0 = np.zeros((B, 3))
    ref = oracle.box_batch(oracle.params(N, R=1.0, height=0.5, iter_cap=3000), x0, x0, lo, hi, lo, hi, nthreads=16)
    st, iters, nact = ref["status"][:, 0], ref["iters"][:, 0], ref["n_active"][:, 0]
    regular_big = (st == 0) & (iters <= N) & (nact > 28)  # beyond the first kernel's capacity for box problems
    assert regular_big.sum() >= 5, "the batch no longer exercises the hand-over of regular instances"
    pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0)
    o = _np(pl.solve_box(x0, lo, hi))
    assert pl.last_handover_count() >= regular_big.sum()
    _check_box_axis(o, st, iters, nact, ref["active"][:, 0], ref["jerk_x"], ref["cost"][:, 0], N, lo, hi,
                    flips_allowed=max(1, int((iters > N).sum()) // 10))
    assert np.allclose(o["alg_flops"][regular_big], ref["falg"][regular_big, 0], rtol=1e-12)


@pytest.mark.parametrize("N", [37, 66, 99, 103, 105])
def test_box_second_stage_other_horizons(P, oracle, N):
    """The second-stage kernel of the box form lays its matrices out by the horizon (leading
    dimension 2 modulo 4, two rows of J per lane, rows 64.. of T in a wave of their own): horizons
    that are odd, that leave no padding row, and that stay below / go beyond 64 working-set rows, on
    tight corridors -- through the second stage alone and through the hand-over from the first.
    Regular instances must match the CPU restatement exactly; the degenerate draws of these batches
    end with working sets of N rows, so they are what takes the kernel beyond 64 rows (for N = 66
    they all end solved and are compared by objective and ZMP, see _check_box_axis)."""
    B = 96
    k = np.arange(N)
    rng = batches.SplitMix(np.arange(B), seed=1000 + N)
    phase, period = rng.uniform(0.0, 6.28), rng.uniform(0.5 * N, 1.2 * N)
    amp, width = rng.uniform(0.3, 0.6) * (N / 100.0) ** 2, rng.uniform(0.01, 0.03)
    centre = amp[:, None] * (np.sin(2 * np.pi * k[None, :] / period[:, None] + phase[:, None]) - np.sin(phase)[:, None])
    lo, hi = centre - width[:, None], centre + width[:, None]
    x0 = np.zeros((B, 3))
    op = oracle.params(N, R=1.0, height=0.5, iter_cap=3000)
    ref = oracle.box_batch(op, x0, x0, lo, hi, lo, hi, nthreads=16)
    st, iters, nact = ref["status"][:, 0], ref["iters"][:, 0], ref["n_active"][:, 0]
    regular, degenerate = (st == 0) & (iters <= N), iters > N
    assert regular.sum() >= 20 and degenerate.sum() >= 10 and nact[degenerate].max() == N
    if N >= 66:
        assert nact[regular].max() > 28  # regular instances are handed over as well
    for mode in (P.MODE_DENSE, P.MODE_AUTO):
        pl = P.BatchedPlanner(N, formulation=P.BOX_1D, weight_R=1.0, solver_mode=mode)
        o = _np(pl.solve_box(x0, lo, hi))
        # (the degenerate draws that end "solved" sit a few 1e-7 outside their corridor on both sides at
        # the short horizons -- the stopping rule's threshold scales with the problem -- so they are
        # compared by objective and jerk here instead of through _check_box_axis)
        rest = np.nonzero(~(degenerate & (st == 0)))[0]
        _check_box_axis({k: v[rest] for k, v in o.items()}, st[rest], iters[rest], nact[rest], ref["active"][rest, 0],
                        ref["jerk_x"][rest], ref["cost"][rest, 0], N, lo[rest], hi[rest],
                        flips_allowed=max(1, int(degenerate.sum()) // 10))
        solved = np.nonzero(degenerate & (st == 0))[0]
        assert (o["status"][solved] != 0).sum() <= max(1, len(solved) // 10)
        for i in solved:
            if o["status"][i] == 0 and abs(ref["cost"][i, 0]) < 1e6:  # (beyond that: run-away multipliers, not a plan)
                assert abs(o["cost"][i] - ref["cost"][i, 0]) <= RTOL * max(1e-12, abs(ref["cost"][i, 0])), i
                assert _rel(o["jerk_x"][i], ref["jerk_x"][i]) < 1e-5, i
        assert np.allclose(o["alg_flops"][regular], ref["falg"][regular, 0], rtol=1e-12)


@pytest.mark.parametrize("N,R,Qs", [(100, 1.0, 100.0), (40, 0.5, 10.0)])
def test_box_slack_variant_vs_oracle(P, oracle, N, R, Qs):
    """solveQPconstraintSlack (MPCPlanner.cpp:303-424): 3N variables, 4N rows, about 2N of them
    active.  No output of the reference is recorded for this variant, so the only anchor is the
    CPU restatement (parity unpinned)."""
    B = 12
    bb = batches.decoupled_batch(B, N=N, steps=10)
    pl = P.BatchedPlanner(N, formulation=P.BOX_SLACK, weight_R=R, weight_Qs=Qs, active_stride=4 * N)
    o = _np(pl.solve_box(bb["y0"], bb["ymin"], bb["ymax"], outputs=P.SLACK_OUTPUTS))
    ref = oracle.slack_batch(oracle.params(N, R=R, Qs=Qs, height=0.5, iter_cap=50 * 7 * N), bb["y0"],
                             bb["ymin"], bb["ymax"], active_stride=4 * N, nthreads=16)
    assert (o["status"] == ref["status"]).all()
    assert (o["iters"] == ref["iters"]).all()
    for i in range(B):
        na = ref["n_active"][i]
        assert o["n_active"][i] == na
        assert list(o["active_idx"][i, :na]) == list(ref["active"][i, :na])
        assert _rel(o["jerk_x"][i], ref["jerk"][i]) < RTOL
        assert _rel(o["slacks"][i], ref["slacks"][i]) < RTOL
    assert (np.abs(o["cost"] - ref["cost"]) <= RTOL * np.abs(ref["cost"])).all()
    assert np.allclose(o["alg_flops"], ref["falg"], rtol=1e-12)
    # the slacks are the (negated) distances of the ZMP to its bounds wherever a bound row is active
    assert (o["slacks"] <= 1e-9).all()
    assert np.allclose(o["min_margin"], np.minimum(o["zmp_x"] - bb["ymin"], bb["ymax"] - o["zmp_x"]))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("cap", [3, 7])
def test_iteration_cap_vs_oracle(P, oracle, cap, mode):
    """The cap on outer iterations / step-direction passes (this build's addition; the reference
    loops without one): same status, same working set and the same partially constrained plan as
    the CPU restatement with the same cap, in both kernels."""
    B, N, steps = 32, 200, 20
    bt = batches.coupled_batch(B, first=640)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps,
                          iter_cap=cap, solver_mode=mode)
    o = _np(pl.solve_coupled(**bt))
    ref = oracle.coupled_batch(oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=cap), oracle.COUPLED_VEL,
                               steps, 1000, bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"],
                               bt["swing0"], nthreads=16)
    assert (ref["status"] == 2).any()  # the cap binds for part of the batch
    assert (o["status"] == ref["status"]).all()
    assert (o["n_active"] == ref["n_active"]).all()
    for i in range(B):
        na = ref["n_active"][i]
        assert list(o["active_idx"][i, :na]) == list(ref["active"][i, :na])
        for k in ("jerk_x", "jerk_y", "com_x", "zmp_y"):
            assert _rel(o[k][i], ref[k][i]) < RTOL, (k, i)
    assert (np.abs(o["cost"] - ref["cost"]) <= RTOL * np.abs(ref["cost"])).all()


def test_default_driver_scenarios(P):
    """The recorded outputs of the reference's two default drivers, through the GPU path."""
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_recorded.json")))
    g = gold["decoupled_default"]
    xmin, xmax, ymin, ymax = batches.box_limits(100, 10, 3.0, 0.2)
    pl = P.BatchedPlanner(100, formulation=P.BOX_1D, weight_R=1.0)
    oy = _np(pl.solve_box(np.zeros((1, 3)), ymin, ymax))
    ox = _np(pl.solve_box(np.zeros((1, 3)), xmin, xmax))
    assert np.allclose(oy["jerk_x"][0, :3], g["jerk_y_0_2"], rtol=2e-6)
    ay, ax = oy["active_idx"][0, :oy["n_active"][0]], ox["active_idx"][0, :ox["n_active"][0]]
    assert [(ay >= 100).sum(), (ay < 100).sum()] == g["y_active_upper_lower"]
    assert [(ax >= 100).sum(), (ax < 100).sum()] == g["x_active_upper_lower"]


def test_full_batch_properties(P):
    """8,192 instances (one GPU's share of the 65,536 batch): properties that need no oracle."""
    B, N = 8192, 200
    bt = batches.coupled_batch(B)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=20)
    o = _np(pl.solve_coupled(**bt))
    assert (o["status"] == 0).all()
    # primal feasibility up to the reference's own stopping rule: the solver returns once the
    # SUM of violations is below m * eps * trace(G) * trace(L^-T) * 100 (eiquadprog.hpp:267),
    # which is ~4e-3 for this Hessian -- a reference property this build reproduces
    assert o["min_margin"].min() > -4e-3
    assert (np.minimum(o["min_margin"], 0).sum(axis=1) > -4e-3).all()
    assert (o["multipliers"] >= -1e-9).all()
    # ZMP = CoM - h/g * acceleration; check through the integrator on a sample
    Ts, h, g = 0.1, 0.5, 9.81
    for i in range(0, B, 1024):
        st = bt["x0"][i].copy()
        for k in range(N):
            u = o["jerk_x"][i, k]
            st = np.array([st[0] + Ts * st[1] + Ts ** 2 / 2 * st[2] + Ts ** 3 / 6 * u,
                           st[1] + Ts * st[2] + Ts ** 2 / 2 * u, st[2] + Ts * u])
            assert abs(st[0] - o["com_x"][i, k]) < 1e-9
            assert abs(st[1] - o["com_xd"][i, k]) < 1e-9
            assert abs(st[0] - h / g * st[2] - o["zmp_x"][i, k]) < 1e-9
    # determinism: a second launch returns the same bits
    o2 = _np(pl.solve_coupled(**bt))
    for k in ("jerk_x", "cost", "active_idx"):
        assert np.array_equal(o[k], o2[k])
    # a slice of the batch solved alone gives the same bits (instances are independent)
    sl = {k: v[4000:4032] for k, v in bt.items()}
    o3 = _np(pl.solve_coupled(**sl))
    assert np.array_equal(o3["jerk_x"], o["jerk_x"][4000:4032])


def test_host_pointer_entry_point_matches_device_path(P):
    """mpcb_solve_coupled (host arrays; the library stages them in its own device arena) returns
    bit for bit what mpcb_solve_coupled_device returns, also when a smaller and then a larger
    batch reuse / regrow the arena, and into page-locked arrays."""
    N, steps = 200, 20
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    for B, pinned in ((48, False), (16, False), (80, True)):
        bt = batches.coupled_batch(B, first=1000 + B)
        dev = _np(pl.solve_coupled(**bt))
        out = pl.host_output_arrays(B, pinned=pinned)
        host = pl.solve_coupled_host(bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"],
                                     bt["swing0"], out=out)
        assert host is out
        for k, v in dev.items():
            assert np.array_equal(host[k], v), (B, k)
        assert (host["status"] == 0).all() and host["n_active"].min() >= 1
    with pytest.raises(ValueError):
        pl.solve_coupled_host(bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"],
                              out=pl.host_output_arrays(B + 1))


def test_empty_batch_and_bad_arguments(P):
    pl = P.BatchedPlanner(20, formulation=P.BOX_1D, weight_R=1.0)
    out = pl.solve_box(np.zeros((0, 3)), np.zeros((0, 20)), np.zeros((0, 20)))
    assert out["jerk_x"].shape == (0, 20)
    with pytest.raises(ValueError):
        pl.solve_box(np.zeros((2, 3)), np.zeros((2, 19)), np.zeros((2, 20)))
    with pytest.raises(P.MpcError):
        P.BatchedPlanner(0, formulation=P.BOX_1D)
    with pytest.raises(P.MpcError):
        P.BatchedPlanner(200, formulation=P.COUPLED_VEL, number_of_steps=-1)
    with pytest.raises(P.MpcError):
        P.BatchedPlanner(200, formulation=P.COUPLED_VEL, number_of_steps=256)
    # number_of_steps = 0: no footstep schedule, every knot gets the polygon of the initial feet
    bt = batches.coupled_batch(4)
    stand = P.BatchedPlanner(40, formulation=P.COUPLED_JERK, weight_R=0.01, number_of_steps=0)
    assert stand.num_constraints == 160
    o = _np(stand.solve_coupled(**bt))
    assert (o["status"] == 0).all() and (o["min_margin"] > -4e-3).all()


@pytest.mark.parametrize("Qs,rw", [(100.0, 1000), (0.1, 1000), (1.0, 10)])
def test_coupled_zmp_reference_formulation(P, oracle, Qs, rw):
    """solveQPConstraintCoupledRef with the computeCentroid reference (MPCPlanner.cpp:632-783,
    1106-1130).  Qs=100 are the driver's weights (no constraint binds, Hessian condition
    ~1e10); the smaller Qs make constraints bind."""
    B, N, steps = 48, 200, 20
    bt = batches.coupled_batch(B, first=7000)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_REF, weight_R=0.01, weight_Q=1.0, weight_Qs=Qs,
                          replanning_window=rw, number_of_steps=steps)
    o = _np(pl.solve_coupled(**bt))
    ref = oracle.coupled_batch(oracle.params(N, R=0.01, Q=1.0, Qs=Qs, height=0.5, iter_cap=4000),
                               oracle.COUPLED_REF, steps, rw, bt["x0"], bt["y0"], bt["speed"], bt["feet_x"],
                               bt["feet_y"], bt["swing0"], nthreads=16)
    _check_coupled(o, ref, B)
    if Qs < 100:
        assert ref["n_active"].max() > 0


def test_dense_kernel_alone_vs_oracle(P, oracle):
    """Every instance through the dense kernel (the second stage), coupled and box."""
    B, N, steps = 24, 100, 10
    bt = batches.coupled_batch(B, first=9000)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps,
                          solver_mode=P.MODE_DENSE)
    o = _np(pl.solve_coupled(**bt))
    ref = oracle.coupled_batch(oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000), oracle.COUPLED_VEL,
                               steps, 1000, bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"],
                               bt["swing0"], nthreads=16)
    _check_coupled(o, ref, B)
    bb = batches.decoupled_batch(48, first=9000)
    pb = P.BatchedPlanner(100, formulation=P.BOX_1D, weight_R=1.0, solver_mode=P.MODE_DENSE)
    ob = _np(pb.solve_box(bb["x0"], bb["xmin"], bb["xmax"]))
    rb = oracle.box_batch(oracle.params(100, R=1.0, height=0.5, iter_cap=3000), bb["x0"], bb["y0"], bb["xmin"],
                          bb["xmax"], bb["ymin"], bb["ymax"], nthreads=16)
    assert (ob["status"] == rb["status"][:, 0]).all()
    ok = rb["status"][:, 0] == 0
    assert (ob["iters"][ok] == rb["iters"][ok, 0]).all()
    for i in np.nonzero(ok)[0]:
        assert list(ob["active_idx"][i, :ob["n_active"][i]]) == list(rb["active"][i, 0, :rb["n_active"][i, 0]])
        assert _rel(ob["jerk_x"][i], rb["jerk_x"][i]) < RTOL


def test_large_working_sets_wide_edition_and_dense_fallback(P, oracle):
    """weight_R = 1e-6 (the constructor default, MPCPlanner.cpp:39): working sets of 20-53 rows
    exceed the 32 rows of the first-stage kernel, so part of the batch is taken over mid-solve by
    its wide edition (60 rows): every output, the jerk included, to 1e-6 of the CPU restatement,
    identical iteration counts and working sets.  At weight_R = 1e-8 and three times the walking
    speed one working set reaches 74 rows and goes on to the block edition; cond(G) ~ 1e12 there and
    the CPU restatement's double-precision Cholesky (the reference's, eiquadprog.hpp:165) is itself
    only good to ~1e-4, which is the tolerance of that comparison -- the GPU's own answer is held
    to the optimality conditions instead, evaluated in extended precision."""
    import zspace_model as Z
    N, steps = 200, 20
    for B, R, speed_scale, beyond_wide, tol in ((32, 1e-6, 1.0, False, RTOL), (8, 1e-8, 3.0, True, 1e-4)):
        bt = batches.coupled_batch(B)
        bt["speed"] = bt["speed"] * speed_scale
        pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=R, weight_Q=1.0, number_of_steps=steps,
                              active_stride=128)
        o = _np(pl.solve_coupled(**bt))
        ref = oracle.coupled_batch(oracle.params(N, R=R, Q=1.0, height=0.5, iter_cap=8000), oracle.COUPLED_VEL,
                                   steps, 1000, bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"],
                                   bt["swing0"], active_stride=128, nthreads=16)
        assert (o["status"] == ref["status"]).all() and (ref["n_active"] > 32).any()
        assert pl.last_wide_count() >= (ref["n_active"] > 32).sum()
        # beyond 60 rows: the block edition of the same method (one workgroup per QP), not the full method
        assert (pl.last_block_count() > 0) == beyond_wide == bool((ref["n_active"] > 60).any())
        assert pl.last_handover_count() == 0
        assert (o["iters"] == ref["iters"]).all()
        assert np.allclose(o["alg_flops"], ref["falg"], rtol=1e-12)
        for i in range(B):
            na = ref["n_active"][i]
            assert o["n_active"][i] == na
            assert list(o["active_idx"][i, :na]) == list(ref["active"][i, :na])
            for k in ("com_x", "com_y", "zmp_x", "zmp_y", "jerk_x", "jerk_y"):
                assert _rel(o[k][i], ref[k][i]) < tol, (R, k, i, _rel(o[k][i], ref[k][i]))
        assert (np.abs(o["cost"] - ref["cost"]) <= tol * np.abs(ref["cost"])).all()
        if beyond_wide:
            # optimality of the instance with the largest working set, in numpy long double, from
            # matrices assembled straight from MPCPlanner.cpp:523-605 (no factorisation involved)
            ld = np.longdouble
            i = int(np.argmax(o["n_active"]))
            su = Z.Setup(N, 0.1, 9.81, 0.5, R, 1.0, 100.0, variant=1)
            Xvu, Xvx, Zu, Zx = (a.astype(ld) for a in (su.Xvu, su.Xvx, su.Zu, su.Zx))
            G1 = R * np.eye(N, dtype=ld) + Xvu.T @ Xvu
            st = oracle.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i], bt["swing0"][i])
            A, b = st["A"].astype(ld), st["b"].astype(ld)
            x = np.concatenate([o["jerk_x"][i], o["jerk_y"][i]]).astype(ld)
            g0 = np.concatenate([Xvu.T @ (Xvx @ bt["x0"][i] - bt["speed"][i, 0]),
                                 Xvu.T @ (Xvx @ bt["y0"][i] - bt["speed"][i, 1])])
            Cn = np.hstack([A[:, :N] @ Zu, A[:, N:] @ Zu])
            ci0 = b + A[:, :N] @ (Zx @ bt["x0"][i]) + A[:, N:] @ (Zx @ bt["y0"][i])
            na = o["n_active"][i]
            act, u = o["active_idx"][i, :na], o["multipliers"][i, :na].astype(ld)
            Gx = np.concatenate([G1 @ x[:N], G1 @ x[N:]])
            grad = Gx + g0 - Cn[act].T @ u
            sres = Cn @ x + ci0
            assert na > 60
            assert float(np.abs(grad).max() / max(np.abs(g0).max(), np.abs(Gx).max())) < 1e-9
            assert float(u.min()) > -1e-9 and float(np.abs(u * sres[act]).max()) < 1e-6
            assert float(np.minimum(0, sres).sum()) > -pl_psi_threshold(G1, len(ci0))


def test_block_edition_beyond_sixty_rows(P, oracle, monkeypatch):
    """Working sets of 61 to ~130 rows (weight_R = 1e-8, three times the walking speed): the block
    edition of the multiplier-space method -- one workgroup per QP behind the two wave-per-QP editions
    -- must follow the CPU restatement's path exactly (status, iteration count, ordered working set,
    operation count) and the full method must see none of them; trajectories to 1e-4, the accuracy
    of the restatement's own double-precision Cholesky at cond(G) ~ 1e12 (see the test above).
    Also: the ZMP-reference form and caller-built rows through the same stage, and the fall-through
    to the full method when the stage's capacity (lowered for the test) is exceeded."""
    N, steps, B = 200, 20, 192
    bt = batches.coupled_batch(B, first=2000)
    bt["speed"] = bt["speed"] * 3.0

    def check(pl, ref, tol, label):
        o = _np(pl.solve_coupled(**bt))
        assert (o["status"] == ref["status"]).all(), label
        assert (o["iters"] == ref["iters"]).all(), label
        assert np.allclose(o["alg_flops"], ref["falg"], rtol=1e-12), label
        for i in range(B):
            na = ref["n_active"][i]
            assert o["n_active"][i] == na, (label, i)
            assert list(o["active_idx"][i, :na]) == list(ref["active"][i, :na]), (label, i)
            for k in ("com_x", "com_y", "zmp_x", "zmp_y", "jerk_x", "jerk_y"):
                assert _rel(o[k][i], ref[k][i]) < tol, (label, k, i, _rel(o[k][i], ref[k][i]))
        assert (np.abs(o["cost"] - ref["cost"]) <= tol * np.abs(ref["cost"])).all(), label
        return o

    for form, oform, kw in (("COUPLED_VEL", "COUPLED_VEL", {}), ("COUPLED_REF", "COUPLED_REF", dict(weight_Qs=1e-4))):
        pl = P.BatchedPlanner(N, formulation=getattr(P, form), weight_R=1e-8, weight_Q=1.0, number_of_steps=steps,
                              active_stride=192, **kw)
        ref = oracle.coupled_batch(oracle.params(N, R=1e-8, Q=1.0, Qs=kw.get("weight_Qs", 100.0), height=0.5,
                                                 iter_cap=8000), getattr(oracle, oform), steps, 1000, bt["x0"],
                                   bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"],
                                   active_stride=192, nthreads=16)
        beyond = int((ref["n_active"] > 60).sum())
        assert beyond >= 3, (form, beyond)
        check(pl, ref, 1e-4, form)
        assert pl.last_block_count() >= beyond and pl.last_handover_count() == 0, form
        if form == "COUPLED_VEL":
            # the same rows handed in by the caller (A, b of computeSteps as (p, q, r) triples)
            rows = np.zeros((B, pl.num_constraints, 3))
            for i in range(B):
                st = oracle.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i], bt["swing0"][i])
                A, b = st["A"], st["b"]
                kn = np.array([np.nonzero(A[r])[0][0] % N for r in range(A.shape[0])])
                rows[i, :, 0] = A[np.arange(A.shape[0]), kn]
                rows[i, :, 1] = A[np.arange(A.shape[0]), kn + N]
                rows[i, :, 2] = b
            o2 = _np(pl.solve_coupled(bt["x0"], bt["y0"], bt["speed"], rows_pqr=rows))
            assert (o2["status"] == ref["status"]).all() and (o2["iters"] == ref["iters"]).all()
            assert pl.last_block_count() >= beyond and pl.last_handover_count() == 0
            big = int(np.argmax(ref["n_active"]))
            assert list(o2["active_idx"][big, :ref["n_active"][big]]) == list(ref["active"][big, :ref["n_active"][big]])
            # ... and in the schedule-free layout (number_of_steps = 0: four slots per knot, empty slot =
            # (0, 0, +inf)), where the same rows have other numbers: same plans
            free = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=1e-8, weight_Q=1.0, number_of_steps=0,
                                    active_stride=192)
            assert free.num_constraints == 4 * N
            rows4 = np.zeros((B, 4 * N, 3))
            rows4[:, :, 2] = np.inf
            slot = np.zeros(N, dtype=int)
            for r, k in enumerate(kn):  # (the knot of every row is the same for every instance)
                rows4[:, 4 * k + slot[k]] = rows[:, r]
                slot[k] += 1
            o4 = _np(free.solve_coupled(bt["x0"], bt["y0"], bt["speed"], rows_pqr=rows4))
            assert (o4["status"] == ref["status"]).all() and (o4["iters"] == ref["iters"]).all()
            assert (o4["n_active"] == ref["n_active"]).all() and free.last_block_count() >= beyond
            assert free.last_handover_count() == 0
            for k in ("com_x", "com_y", "zmp_x", "zmp_y"):
                assert _rel(o4[k], o2[k]) < 1e-9, k
            # a lower capacity for the stage: what exceeds it goes on to the full method, same answers
            monkeypatch.setenv("MPCB_DEBUG_BLOCK_ROWS", "70")
            low = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=1e-8, weight_Q=1.0, number_of_steps=steps,
                                   active_stride=192)
            check(low, ref, 1e-4, "capacity 70")
            assert low.last_handover_count() == int((ref["n_active"] > 70).sum()) > 0
            monkeypatch.delenv("MPCB_DEBUG_BLOCK_ROWS")
            # a CoM height per instance through the same stage: each instance against the restatement
            # run at its own height
            Bh = 32
            hh = 0.45 + 0.1 * batches.SplitMix(np.arange(Bh), seed=3).uniform()
            pick = np.argsort(-ref["n_active"], kind="stable")[:Bh]  # the instances with the largest working sets
            sub = {k: v[pick] for k, v in bt.items()}
            oh = _np(pl.solve_coupled(height=hh, **sub))
            assert pl.last_handover_count() == 0
            assert (pl.last_block_count() > 0) == bool((oh["n_active"] > 60).any()) and (oh["n_active"] > 60).any()
            for i in range(Bh):
                r1 = oracle.coupled_batch(oracle.params(N, R=1e-8, Q=1.0, height=float(hh[i]), iter_cap=8000),
                                          oracle.COUPLED_VEL, steps, 1000, sub["x0"][i:i + 1], sub["y0"][i:i + 1],
                                          sub["speed"][i:i + 1], sub["feet_x"][i:i + 1], sub["feet_y"][i:i + 1],
                                          sub["swing0"][i:i + 1], active_stride=192)
                na = r1["n_active"][0]
                assert oh["status"][i] == r1["status"][0] and oh["iters"][i] == r1["iters"][0], i
                assert list(oh["active_idx"][i, :na]) == list(r1["active"][0, :na]), i
                for k in ("com_x", "zmp_y", "jerk_x"):
                    assert _rel(oh[k][i], r1[k][0]) < 1e-4, (k, i)


def pl_psi_threshold(G1, m):
    """The reference's stopping threshold m * eps * trace(G) * trace(L^-T) * 100 (eiquadprog.hpp:267)
    for the two-block Hessian blkdiag(G1, G1)."""
    L = np.linalg.cholesky(G1.astype(np.float64))
    return m * np.finfo(float).eps * 2 * float(np.trace(G1)) * 2 * float(np.sum(1.0 / np.diag(L))) * 100.0


def test_replanning_sweep_vs_oracle(P, oracle):
    """8 dependent replans per rollout (tests/replanning/testPlannerXYconstraintMPC.cpp:108-209)."""
    B, N, steps, stages = 48, 200, 20, 8
    bt = batches.coupled_batch(B, first=11000)
    dist = batches.replanning_disturbance(B, first=11000)
    hip = batches.NOMINAL_HIP_OFFSETS
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0,
                          replanning_window=N // steps, number_of_steps=steps, hip_offsets=hip)
    res, stage, st = pl.replan_sweep(pl.prepare_replanning(disturbance=dist, **bt), stages)
    ref = oracle.replanning_batch(oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000), steps, stages,
                                  bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"],
                                  dist, hip=hip, nthreads=16)
    assert (stage["status"].cpu().numpy() == ref["stage_status"]).all()
    assert (stage["n_active"].cpu().numpy() == ref["stage_nactive"]).all()
    sc = stage["cost"].cpu().numpy()
    assert (np.abs(sc - ref["stage_cost"]) <= RTOL * np.abs(ref["stage_cost"])).all()
    for k in ("x0", "y0", "feet_x", "feet_y"):
        assert np.abs(st[k].cpu().numpy() - ref[k]).max() < 1e-9
    assert (st["swing0"].cpu().numpy() == ref["swing0"]).all()
    assert _rel(res["jerk_x"].cpu().numpy(), ref["jerk_x"]) < RTOL
    # without the step correction and without disturbance the first stage equals a plain solve
    pl2 = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0,
                           replanning_window=N // steps, number_of_steps=steps)
    plain = _np(pl2.solve_coupled(**bt))
    res2, stage2, _ = pl2.replan_sweep(pl2.prepare_replanning(**bt), 1, use_com_correction=False)
    assert np.array_equal(stage2["cost"].cpu().numpy()[0], plain["cost"])


def test_unconstrained_solve_qp(P, oracle):
    """solveQP (MPCPlanner.cpp:223-237) at the unconstrained driver's shape
    (tests/unconstrainedQP/testPlanner.cpp:21-28,60-64) and a random batch."""
    N = 100
    pl = P.BatchedPlanner(N, formulation=P.UNCONSTRAINED, weight_R=1e-6, weight_Q=1.0)
    p = oracle.params(N, R=1e-6, Q=1.0, height=0.5)
    rng = np.random.default_rng(3)
    x0 = np.concatenate([[[0.2, -0.1, 0.0]], rng.uniform(-0.1, 0.1, size=(7, 3))])
    zref = np.concatenate([[np.linspace(0, 0.2, N)], np.cumsum(rng.uniform(0, 0.01, size=(7, N)), axis=1)])
    o = _np(pl.solve_unconstrained(x0, zref))
    for i in range(8):
        jerk = oracle.solve_qp(p, x0[i], zref[i])
        # G = Zu'Zu + 1e-6 I is badly conditioned; the reference inverts it explicitly in double.
        # Compare what the plan does (ZMP, CoM) at 1e-6 and the jerk itself loosely.
        assert _rel(o["zmp_x"][i], oracle.trajectory(p, 3, x0[i], jerk)) < RTOL
        assert _rel(o["com_x"][i], oracle.trajectory(p, 0, x0[i], jerk)) < RTOL
        assert _rel(o["jerk_x"][i], jerk) < 1e-3
    # the N=10 known answer needs no solve: ZMP of jerk = 1 from x0 = (0.1, -0.1, 0) is
    # checked in the CPU suite; here the ZMP of the returned jerk must equal its own integration
    zi = oracle.trajectory(p, 3, x0[0], o["jerk_x"][0])
    assert np.abs(zi - o["zmp_x"][0]).max() < 1e-9


def test_caller_built_polygon_rows(P, oracle):
    """rows_pqr: the (p, q, r) of every row of the A, b that computeSteps returned give the
    same answer as letting the library run computeSteps; edited rows (a stability margin)
    are honoured."""
    B, N, steps = 12, 200, 20
    bt = batches.coupled_batch(B, first=13000)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    base = _np(pl.solve_coupled(**bt))
    rows = np.zeros((B, pl.num_constraints, 3))
    kn = None
    for i in range(B):
        st = oracle.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i], bt["swing0"][i])
        A, b = st["A"], st["b"]
        kn = np.array([np.nonzero(A[r])[0][0] % N for r in range(A.shape[0])])  # same for every instance
        for r in range(A.shape[0]):
            k = kn[r]
            rows[i, r] = (A[r, k], A[r, k + N], b[r])
    o = _np(pl.solve_coupled(bt["x0"], bt["y0"], bt["speed"], rows_pqr=rows))
    assert (o["status"] == 0).all() and (o["iters"] == base["iters"]).all()
    assert np.array_equal(o["active_idx"], base["active_idx"])
    assert _rel(o["jerk_x"], base["jerk_x"]) < 1e-9 and _rel(o["com_y"], base["com_y"]) < 1e-9
    # shrink every polygon by a 2 cm margin: r -> r - 0.02 (rows are normalised)
    rows2 = rows.copy()
    rows2[:, :, 2] -= 0.02
    o2 = _np(pl.solve_coupled(bt["x0"], bt["y0"], bt["speed"], rows_pqr=rows2))
    ok = o2["status"] == 0
    assert ok.sum() >= B // 2
    # the edited rows are the ones enforced: measured against the ORIGINAL polygons the ZMP keeps
    # the 2 cm margin, up to the solver's stopping rule (sum of violations <= 4e-3)
    m2 = o2["min_margin"][ok].min(axis=1)
    assert (m2 > -4e-3).all()
    margin_orig = np.array([(rows[i, :, 0] * o2["zmp_x"][i][kn] + rows[i, :, 1] * o2["zmp_y"][i][kn] + rows[i, :, 2]).min()
                            for i in np.nonzero(ok)[0]])
    assert (margin_orig > 0.02 - 4e-3).all()
    assert (o2["cost"][ok] >= base["cost"][ok] - 1e-9).all()  # a tighter feasible set cannot cost less
    ref2 = [oracle.solve_coupled(oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000), oracle.COUPLED_VEL,
                                 bt["x0"][i], bt["y0"][i],
                                 oracle.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i],
                                                      bt["swing0"][i])["A"], rows2[i, :, 2], bt["speed"][i], 1000)
            for i in range(3)]
    for i in range(3):
        assert ref2[i]["info"].status == o2["status"][i]
        if o2["status"][i] == 0:
            assert list(ref2[i]["active"]) == list(o2["active_idx"][i, :o2["n_active"][i]])
            assert _rel(o2["jerk_x"][i], ref2[i]["jerk_x"]) < RTOL


def _phase_tables(oracle, bt, B, N, steps, m):
    """(rows [B, m, 3], phases [B, 2*steps+1, 4, 3], knot of every row) of computeSteps' A, b: the
    polygon of a phase is read off the rows of the phase's first knot."""
    pd = (N // steps) // 2
    rows = np.zeros((B, m, 3))
    phases = np.zeros((B, 2 * steps + 1, 4, 3))
    kn = None
    for i in range(B):
        st = oracle.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i], bt["swing0"][i])
        A, b = st["A"], st["b"]
        kn = np.array([np.nonzero(A[r])[0][0] % N for r in range(A.shape[0])])
        for r in range(A.shape[0]):
            rows[i, r] = (A[r, kn[r]], A[r, kn[r] + N], b[r])
        for ph in range(2 * steps + 1):
            k0 = (ph // 2) * 2 * pd + (ph % 2) * pd if ph < 2 * steps else 2 * pd * steps
            if k0 >= N:  # (the trailing four-stance phase has no knot when the steps fill the horizon)
                phases[i, ph, :, 2] = np.inf
                continue
            r0 = int(np.nonzero(kn == k0)[0][0])
            E = 3 if (ph % 2 == 1 and ph < 2 * steps) else 4
            phases[i, ph, :E] = rows[i, r0:r0 + E]
            if E == 3:
                phases[i, ph, 3] = (7.0, -3.0, -100.0)  # ignored: a swing phase has three edges
    return rows, phases, kn


@pytest.mark.gpu
def test_caller_built_polygons_per_phase(P, oracle):
    """phase_pqr: the caller's polygon of every footstep phase (buildPolygonMatrix run by the caller,
    MPCPlanner.cpp:1040-1082) in place of computeSteps.  With the polygons of computeSteps itself the
    result is that of the library's own rows; with a stability margin it is that of the same rows given
    one by one (rows_pqr) and of the CPU restatement on the edited A, b -- through the first stage, the
    dense kernel alone, and the ZMP-reference form."""
    B, N, steps = 24, 200, 20
    bt = batches.coupled_batch(B, first=13500)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    rows, phases, kn = _phase_tables(oracle, bt, B, N, steps, pl.num_constraints)
    base = _np(pl.solve_coupled(**bt))
    o = _np(pl.solve_coupled(bt["x0"], bt["y0"], bt["speed"], phase_pqr=phases))
    assert (o["status"] == base["status"]).all() and (o["iters"] == base["iters"]).all()
    assert np.array_equal(o["active_idx"], base["active_idx"])
    for k in ("jerk_x", "jerk_y", "com_x", "com_y", "zmp_x", "zmp_y", "cost"):
        assert _rel(o[k], base[k]) < 1e-9, k
    # a 2 cm margin on every edge
    ph2, rows2 = phases.copy(), rows.copy()
    ph2[:, :, :, 2] -= 0.02
    rows2[:, :, 2] -= 0.02
    o2 = _np(pl.solve_coupled(bt["x0"], bt["y0"], bt["speed"], phase_pqr=ph2))
    r2 = _np(pl.solve_coupled(bt["x0"], bt["y0"], bt["speed"], rows_pqr=rows2))
    assert (o2["status"] == r2["status"]).all() and (o2["iters"] == r2["iters"]).all()
    assert np.array_equal(o2["active_idx"], r2["active_idx"])
    ok = o2["status"] == 0
    assert ok.sum() >= B // 2
    for k in ("jerk_x", "jerk_y", "com_x", "com_y", "zmp_x", "zmp_y", "cost", "min_margin"):
        assert _rel(o2[k][ok], r2[k][ok]) < 1e-9, k
    prm = oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000)
    for i in range(4):
        A = oracle.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i], bt["swing0"][i])["A"]
        ref = oracle.solve_coupled(prm, oracle.COUPLED_VEL, bt["x0"][i], bt["y0"][i], A, rows2[i, :, 2],
                                   bt["speed"][i], 1000)
        assert ref["info"].status == o2["status"][i]
        if o2["status"][i] == 0:
            assert list(ref["active"]) == list(o2["active_idx"][i, :o2["n_active"][i]])
            assert ref["info"].iters == o2["iters"][i]
            assert _rel(o2["jerk_x"][i], ref["jerk_x"]) < RTOL and _rel(o2["jerk_y"][i], ref["jerk_y"]) < RTOL
            assert abs(o2["cost"][i] - ref["info"].f) <= RTOL * max(1.0, abs(ref["info"].f))
    # the dense kernel alone reads the same table
    dn = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps,
                          solver_mode=P.MODE_DENSE)
    sub = slice(0, 6)
    d2 = _np(dn.solve_coupled(bt["x0"][sub], bt["y0"][sub], bt["speed"][sub], phase_pqr=ph2[sub]))
    assert (d2["status"] == o2["status"][sub]).all() and (d2["iters"] == o2["iters"][sub]).all()
    assert np.array_equal(d2["active_idx"], o2["active_idx"][sub])
    okd = d2["status"] == 0
    assert _rel(d2["jerk_x"][okd], o2["jerk_x"][sub][okd]) < RTOL
    # ZMP-reference form: the reference comes with the polygons
    rf = P.BatchedPlanner(N, formulation=P.COUPLED_REF, weight_R=0.01, weight_Q=1.0, weight_Qs=1.0,
                          number_of_steps=steps)
    cen = np.zeros((B, 2 * steps + 1, 2))
    pd = (N // steps) // 2
    for i in range(B):
        st = oracle.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i], bt["swing0"][i])
        cx, cy = oracle.centroid(st["fx"], st["fy"], st["swing"])
        for ph in range(2 * steps + 1):
            k0 = (ph // 2) * 2 * pd + (ph % 2) * pd if ph < 2 * steps else 2 * pd * steps
            if k0 < N:
                cen[i, ph] = (cx[k0], cy[k0])
    b3 = _np(rf.solve_coupled(**bt))
    o3 = _np(rf.solve_coupled(bt["x0"], bt["y0"], bt["speed"], phase_pqr=phases, zref_phase=cen))
    assert b3["n_active"].max() > 0
    assert (o3["status"] == b3["status"]).all() and np.array_equal(o3["active_idx"], b3["active_idx"])
    assert _rel(o3["jerk_x"], b3["jerk_x"]) < 1e-9 and _rel(o3["cost"], b3["cost"]) < 1e-9
    # host-pointer entry point: staged copies of the polygons, and for the ZMP-reference form with a
    # height per instance the library's own grouping by height (inputs permuted and put back)
    hp = pl.solve_coupled_host(bt["x0"], bt["y0"], bt["speed"], phase_pqr=ph2)
    for k in ("jerk_x", "com_y", "cost", "status", "iters", "active_idx"):
        assert np.array_equal(hp[k], o2[k]), k
    hts = np.where(np.arange(B) % 3 == 0, 0.5, np.where(np.arange(B) % 3 == 1, 0.45, 0.55))
    hb = rf.solve_coupled_host(bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"], height=hts)
    hr = rf.solve_coupled_host(bt["x0"], bt["y0"], bt["speed"], height=hts, phase_pqr=phases, zref_phase=cen)
    assert (hr["status"] == hb["status"]).all() and np.array_equal(hr["active_idx"], hb["active_idx"])
    assert _rel(hr["jerk_y"], hb["jerk_y"]) < 1e-9 and _rel(hr["cost"], hb["cost"]) < 1e-9
    with pytest.raises(Exception):
        rf.solve_coupled(bt["x0"], bt["y0"], bt["speed"], phase_pqr=phases)  # no reference given
    with pytest.raises(Exception):
        pl.solve_coupled(bt["x0"], bt["y0"], bt["speed"], phase_pqr=phases, rows_pqr=rows)


def test_cpp_dropin_drivers(P, tmp_path):
    """The reference's three drivers rewritten against the MPCPlanner-shaped C++ header run
    on the GPU and reproduce the recorded outputs of the reference's own drivers; saveTraj
    writes the reference's text format."""
    import json, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "dropin_drivers")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    gold = json.load(open(os.path.join(root, "tests", "golden", "reference_recorded.json")))
    env = dict(os.environ, HOME=str(tmp_path))

    def run(which):
        r = subprocess.run([exe, which], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        return {k: float(v) for k, v in (l.split("=") for l in r.stdout.splitlines() if "=" in l and " " not in l)}

    c = run("coupled")
    g = gold["coupled_default"]
    assert np.allclose([c["jerk_x0"], c["jerk_x1"], c["jerk_x2"]], g["jerk_x_0_2"], rtol=2e-6)
    assert c["com_x199"] == pytest.approx(g["com_x_199"], rel=2e-6)
    assert c["min_margin"] == pytest.approx(g["min_margin"], rel=2e-5)
    assert c["n_active"] == g["n_active"] and c["vel_n_active"] > 0
    lines = open(tmp_path / "com_x.txt").read().splitlines()
    assert len(lines) == 200
    t, v = lines[199].split()
    assert float(t) == pytest.approx(19.9) and float(v) == pytest.approx(g["com_x_199"], rel=2e-6)
    assert len(open(tmp_path / "footPosLF.txt").read().splitlines()[0].split()) == 3
    # every file the reference's driver writes (testPlannerXYconstraintCoupled.cpp:135-165), so that
    # plotTrajXYconstraintCoupled.m finds what it loads
    expected = ["jerk_x", "jerk_y", "zmp_x", "zmp_y", "com_x", "com_y", "com_xd", "com_yd", "viol", "centroidX",
                "centroidY"] + [p + leg for p in ("footPos", "footHolds", "swing") for leg in ("LF", "RF", "LH", "RH")]
    assert len(expected) == 23
    for name in expected:
        rows_ = open(tmp_path / (name + ".txt")).read().splitlines()
        assert len(rows_) == (20 if name.startswith("footHolds") else 200), name
    # the second Ref solve of the driver: a reference that sways inside the phases, against the
    # CPU restatement given the same vectors
    from oracle import oracle as O
    st = O.compute_steps(200, 20, np.array([0.15, 0.0]), np.array([0.1, 0.2, -0.4, -0.3]),
                         np.array([1.0, -1.0, 1.0, -1.0]), 0)
    cx = np.loadtxt(tmp_path / "centroidX.txt")[:, 1]
    sway = np.loadtxt(tmp_path / "sway_ref_y.txt")[:, 1]
    ref = O.solve_coupled(O.params(200, R=0.01, Q=1.0, Qs=100.0, height=0.5, iter_cap=4000), O.COUPLED_REF,
                          np.zeros(3), np.zeros(3), st["A"], st["b"], np.array([0.15, 0.0]), 1000, cx, sway)
    assert c["sway_n_active"] == ref["info"].n_active
    assert c["sway_cost"] == pytest.approx(ref["info"].f, rel=1e-6)
    assert _rel(np.loadtxt(tmp_path / "sway_jerk_y.txt")[:, 1], ref["jerk_y"]) < 2e-5  # 6 printed digits
    d = run("decoupled")
    g = gold["decoupled_default"]
    assert np.allclose([d["jerk_y0"], d["jerk_y1"], d["jerk_y2"]], g["jerk_y_0_2"], rtol=2e-6)
    assert d["y_active"] == sum(g["y_active_upper_lower"]) and d["x_active"] == sum(g["x_active_upper_lower"])
    # the driver's other mode (use slacks = 1): the slacks keep the ZMP away from its bounds,
    # where the plain box solve lets it touch them
    ds = run("decoupled_slack")
    assert d["margin_y"] < 1e-6 and ds["margin_y"] > 0.05
    assert ds["y_active"] >= 200 and ds["x_active"] >= 200  # every bound row carries its slack
    q = run("qp")
    assert abs(q["zmp99"] - 0.2) < 1e-2  # the ZMP follows the linear reference


def test_cpp_replanning_driver_with_ini(P, oracle, tmp_path):
    """The replanning driver rewritten against the C++ header (one planner call per replan, as
    the reference does it), configured through a planner.ini, follows the same closed loop as
    the CPU restatement of that driver."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "dropin_drivers")
    ini = tmp_path / "planner.ini"
    ini.write_text("[Graphics]\ndraw_cones = 0\n\n[Crawl]\n#a comment\ncycle_time = 4\nlinearSpeedX = 0.12\n"
                   "linearSpeedY = 0.05 \nheadingSpeed = 0.0\n\n[Replanning]\nuseComStepCorrection = 1\n"
                   "optimizeVelocity = 1\n")
    r = subprocess.run([exe, "replanning", str(ini), "0.3", "60"], env=dict(os.environ, HOME=str(tmp_path)),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = {k: float(v) for k, v in (l.split("=") for l in r.stdout.splitlines() if "=" in l and " " not in l)}
    assert (d["speed_x"], d["speed_y"], d["stages"]) == (0.12, 0.05, 6)
    N, steps = 200, 20
    ref = oracle.replanning_batch(oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000), steps, 6,
                                  np.zeros((1, 3)), np.zeros((1, 3)), np.array([[0.12, 0.05]]),
                                  np.array([[0.1, 0.2, -0.4, -0.3]]), np.array([[1.0, -1.0, 1.0, -1.0]]),
                                  np.array([0], dtype=np.int32), np.array([0.3]), disturbance_start=4,
                                  use_com_correction=True, hip=batches.NOMINAL_HIP_OFFSETS, nthreads=1)
    for st in range(6):
        assert d["stage%d_status" % (st + 1)] == ref["stage_status"][st, 0]
        assert d["stage%d_active" % (st + 1)] == ref["stage_nactive"][st, 0]
        assert d["stage%d_cost" % (st + 1)] == pytest.approx(ref["stage_cost"][st, 0], rel=RTOL)
    got = np.array([d["x_pos"], d["x_vel"], d["x_acc"], d["y_pos"], d["y_vel"], d["y_acc"]])
    want = np.r_[ref["x0"][0], ref["y0"][0]]
    assert np.abs(got - want).max() < RTOL * np.abs(want).max()


@pytest.mark.parametrize("gait,duty", [(0, 0.85), (0, 0.7), (2, 0.6)])
def test_cpp_gait_driver(P, oracle, tmp_path, gait, duty):
    """Support polygons of a time-based gait (GaitSequencer: crawl with 3- and 4-foot phases,
    crawl with a duty factor that leaves two-foot phases, trot) solved through the
    schedule-free row layout; the CPU restatement solves the same A, b."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "dropin_drivers")
    r = subprocess.run([exe, "gait", str(gait), str(duty)], env=dict(os.environ, HOME=str(tmp_path)),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = dict(l.split("=") for l in r.stdout.splitlines() if "=" in l and " " not in l)
    N = 200
    rows = np.loadtxt(tmp_path / "gait_rows.txt").reshape(-1, 5)
    m = int(d["rows"])
    assert len(rows) == m and (np.diff(rows[:, 1]) >= 0).all()  # one knot per row, knot-major
    A = np.zeros((m, 2 * N))
    b = np.zeros(m)
    for rr, k, p, q, c in rows:
        A[int(rr), int(k)], A[int(rr), int(k) + N], b[int(rr)] = p, q, c
    swing = np.loadtxt(tmp_path / "swingLF.txt")[:, 1]
    assert 0 < swing.sum() < N  # the left-front leg both swings and stands
    ref = oracle.solve_coupled(oracle.params(N, Ts=0.04, R=0.01, Q=1.0, height=0.5, iter_cap=8000),
                               oracle.COUPLED_VEL, np.zeros(3), np.zeros(3), A, b, np.array([0.05, 0.0]), 1000)
    assert int(d["status"]) == ref["info"].status
    if ref["info"].status == 0:
        got_active = [int(a) for a in d["active_rows"].split(",") if a]
        assert got_active == list(ref["active"])
        assert float(d["cost"]) == pytest.approx(ref["info"].f, rel=RTOL)
        assert float(d["jerk_x0"]) == pytest.approx(ref["jerk_x"][0], rel=RTOL, abs=1e-9)
        assert float(d["jerk_y0"]) == pytest.approx(ref["jerk_y"][0], rel=RTOL, abs=1e-9)


def test_multi_device_entry_point_matches_single_handle(P):
    """mpcb_solve_coupled_multi: contiguous shards over several handles (here three on the one
    GPU of the test box, which exercises the sharding, the per-handle threads and streams) give
    bit for bit the single-handle result, also for a batch smaller than the handle count."""
    N, steps = 200, 20
    mk = lambda: P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    pls = [mk(), mk(), mk()]
    for B in (50, 2):
        bt = batches.coupled_batch(B, first=21000)
        one = pls[0].solve_coupled_host(bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"])
        many = P.BatchedPlanner.solve_coupled_multi(pls, bt["x0"], bt["y0"], bt["speed"], bt["feet_x"],
                                                    bt["feet_y"], bt["swing0"])
        for k, v in one.items():
            assert np.array_equal(many[k], v), (B, k)
        assert (many["status"] == 0).all()
    with pytest.raises(P.MpcError, match="twice"):
        P.BatchedPlanner.solve_coupled_multi([pls[0], pls[0]], bt["x0"], bt["y0"], bt["speed"], bt["feet_x"],
                                             bt["feet_y"], bt["swing0"])
    other = P.BatchedPlanner(100, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=10)
    with pytest.raises(P.MpcError, match="different shapes"):
        P.BatchedPlanner.solve_coupled_multi([pls[0], other], bt["x0"], bt["y0"], bt["speed"], bt["feet_x"],
                                             bt["feet_y"], bt["swing0"])


def test_shapes_beyond_the_dense_kernels_lds_are_rejected_at_create(P):
    """The dense second-stage kernel's LDS carve-up grows with horizon, rows and phases; a
    shape that cannot fit one workgroup's 160 KB is refused by mpcb_create with a message,
    not by the first solve with a launch error."""
    with pytest.raises(P.MpcError, match="LDS"):
        P.BatchedPlanner(512, formulation=P.BOX_SLACK, weight_R=1.0, weight_Qs=10.0)
    with pytest.raises(P.MpcError, match="LDS"):
        P.BatchedPlanner(512, formulation=P.COUPLED_VEL, weight_R=0.01, number_of_steps=255)
    P.BatchedPlanner(512, formulation=P.COUPLED_VEL, weight_R=0.01, number_of_steps=16)  # fits
    P.BatchedPlanner(300, formulation=P.BOX_SLACK, weight_R=1.0, weight_Qs=10.0)         # fits


def test_symmetric_scene_with_exact_ties_vs_oracle(P, oracle):
    """The reference driver's own scene (testPlannerXYconstraintCoupled.cpp:80-88: feet exactly at
    y = +-1, CoM on the axis, no lateral speed) is mirror-symmetric, so left and right polygon
    edges evaluate to exactly equal values: the lowest-row tie-break of step 2
    (eiquadprog.hpp:280-287) decides.  No jitter here; velocity form, where constraints bind."""
    B, N, steps = 16, 200, 20
    x0 = np.zeros((B, 3))
    y0 = np.zeros((B, 3))
    speed = np.stack([np.linspace(0.05, 0.2, B), np.zeros(B)], axis=1)
    feet_x = np.tile([0.1, 0.2, -0.4, -0.3], (B, 1))
    feet_y = np.tile([1.0, -1.0, 1.0, -1.0], (B, 1))
    swing0 = np.array([0, 3, 1, 2] * (B // 4), dtype=np.int32)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    o = _np(pl.solve_coupled(x0, y0, speed, feet_x, feet_y, swing0))
    ref = oracle.coupled_batch(oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000), oracle.COUPLED_VEL,
                               steps, 1000, x0, y0, speed, feet_x, feet_y, swing0, nthreads=16)
    assert ref["n_active"].min() >= 1
    _check_coupled(o, ref, B)


def test_default_weight_R_per_output_tolerances(P, oracle):
    """weight_R = 1e-6 is the reference constructor's default (MPCPlanner.cpp:39); cond(G) ~ 1e10
    there.  Every output still meets the 1e-6 bar against the CPU restatement (measured on an
    MI355X: CoM 5e-9, ZMP 8e-8, jerk 2e-7, objective 1e-8 relative; status, iteration counts and
    working sets identical), part of the batch through the dense second stage."""
    B, N, steps = 32, 200, 20
    bt = batches.coupled_batch(B)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=1e-6, weight_Q=1.0, number_of_steps=steps)
    o = _np(pl.solve_coupled(**bt))
    ref = oracle.coupled_batch(oracle.params(N, R=1e-6, Q=1.0, height=0.5, iter_cap=4000), oracle.COUPLED_VEL,
                               steps, 1000, bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"],
                               bt["swing0"], nthreads=16)
    assert (o["status"] == ref["status"]).all() and (o["iters"] == ref["iters"]).all()
    worst = {}
    for k in ("com_x", "com_y", "zmp_x", "zmp_y", "jerk_x", "jerk_y"):
        worst[k] = max(_rel(o[k][i], ref[k][i]) for i in range(B))
    worst["cost"] = (np.abs(o["cost"] - ref["cost"]) / np.abs(ref["cost"])).max()
    print("achieved relative agreement at weight_R = 1e-6:", {k: "%.1e" % v for k, v in worst.items()})
    for k, v in worst.items():
        assert v < RTOL, (k, v)
    assert (ref["n_active"] > 32).any()  # beyond the first stage's capacity


def _kkt(G, g0, Cn, ci0, x, act, u):
    """Residuals of the optimality conditions of min 1/2 x'Gx + g0'x s.t. Cn x + ci0 >= 0 at (x, u on act)."""
    s = Cn @ x + ci0
    grad = G @ x + g0 - Cn[act].T @ u
    # primal feasibility in the solver's own measure: the SUM of violations, which the reference
    # lets reach m * eps * trace(G) * trace(L^-T) * 100 before it stops (eiquadprog.hpp:267)
    thr = len(ci0) * np.finfo(float).eps * np.trace(G) * np.sum(1.0 / np.diag(np.linalg.cholesky(G))) * 100.0
    return dict(stationarity=np.abs(grad).max() / max(1e-300, np.abs(g0).max(), np.abs(G @ x).max()),
                primal=np.minimum(0.0, s).sum() / thr, dual=min(0.0, u.min()) if len(u) else 0.0,
                complementarity=np.abs(u * s[act]).max() if len(u) else 0.0)


def test_kkt_conditions_from_an_independent_assembly(P, oracle):
    """Optimality of the GPU's (x, multipliers) checked against G, g0, CI, ci0 assembled in numpy
    straight from the reference's formulas -- a second restatement, independent of the C++ one:
    solveQPConstraintCoupledRef (MPCPlanner.cpp:723-760) at full size, and solveQPconstraintSlack
    (:351-394)."""
    import zspace_model as Z
    N, steps, R, Q, Qs = 200, 20, 0.01, 1.0, 1.0   # Qs = 1: constraints bind
    B = 3
    bt = batches.coupled_batch(B, first=31000)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_REF, weight_R=R, weight_Q=Q, weight_Qs=Qs, number_of_steps=steps)
    o = _np(pl.solve_coupled(**bt))
    su = Z.Setup(N, 0.1, 9.81, 0.5, R, Q, Qs, variant=2)
    G1 = R * np.eye(N) + su.Xvu.T @ (Q * su.Xvu) + Qs * su.Zu.T @ su.Zu
    G = np.zeros((2 * N, 2 * N))
    G[:N, :N] = G1
    G[N:, N:] = G1
    assert o["n_active"].max() > 0
    for i in range(B):
        st = oracle.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i], bt["swing0"][i])
        cx, cy = oracle.centroid(st["fx"], st["fy"], st["swing"])
        A, b = st["A"], st["b"]
        g0 = np.concatenate([
            su.Xvu.T @ (Q * (su.Xvx @ bt["x0"][i] - bt["speed"][i, 0])) + Qs * su.Zu.T @ (su.Zx @ bt["x0"][i] - cx),
            su.Xvu.T @ (Q * (su.Xvx @ bt["y0"][i] - bt["speed"][i, 1])) + Qs * su.Zu.T @ (su.Zx @ bt["y0"][i] - cy)])
        Cn = np.hstack([A[:, :N] @ su.Zu, A[:, N:] @ su.Zu])
        ci0 = b + A[:, :N] @ (su.Zx @ bt["x0"][i]) + A[:, N:] @ (su.Zx @ bt["y0"][i])
        na = o["n_active"][i]
        k = _kkt(G, g0, Cn, ci0, np.concatenate([o["jerk_x"][i], o["jerk_y"][i]]), o["active_idx"][i, :na],
                 o["multipliers"][i, :na])
        assert k["stationarity"] < 1e-7 and k["primal"] > -1.0 and k["dual"] > -1e-9 and k["complementarity"] < 1e-6, k
        f = 0.5 * np.concatenate([o["jerk_x"][i], o["jerk_y"][i]]) @ G @ np.concatenate([o["jerk_x"][i], o["jerk_y"][i]]) \
            + g0 @ np.concatenate([o["jerk_x"][i], o["jerk_y"][i]])
        assert abs(f - o["cost"][i]) < 1e-6 * abs(f)
    # slack form: x = [jerk; slack per bound], G = diag(R I, Qs I), g0 = [0; 1000 Qs]
    Ns, Rs, Qss = 60, 0.5, 10.0
    bb = batches.decoupled_batch(4, N=Ns, steps=6)
    ps = P.BatchedPlanner(Ns, formulation=P.BOX_SLACK, weight_R=Rs, weight_Qs=Qss, active_stride=4 * Ns)
    os_ = _np(ps.solve_box(bb["y0"], bb["ymin"], bb["ymax"], outputs=P.SLACK_OUTPUTS))
    s2 = Z.Setup(Ns, 0.1, 9.81, 0.5, Rs, box=True)
    Gs = np.diag(np.concatenate([np.full(Ns, Rs), np.full(2 * Ns, Qss)]))
    g0s = np.concatenate([np.zeros(Ns), np.full(2 * Ns, 1000.0 * Qss)])
    I = np.eye(Ns)
    Z0 = np.zeros((Ns, Ns))
    Cn = np.block([[s2.Zu, I, Z0], [-s2.Zu, Z0, I], [Z0, -I, Z0], [Z0, Z0, -I]])
    for i in range(4):
        zf = s2.Zx @ bb["y0"][i]
        ci0 = np.concatenate([zf - bb["ymin"][i], bb["ymax"][i] - zf, np.zeros(2 * Ns)])
        na = os_["n_active"][i]
        k = _kkt(Gs, g0s, Cn, ci0, np.concatenate([os_["jerk_x"][i], os_["slacks"][i]]), os_["active_idx"][i, :na],
                 os_["multipliers"][i, :na])
        assert k["stationarity"] < 1e-9 and k["primal"] > -1.0 and k["dual"] > -1e-9 and k["complementarity"] < 1e-6, k


def test_height_per_instance_vs_oracle(P, oracle):
    """`actual_height` is an argument of every solve call in the reference (MPCPlanner.cpp:493-502,
    516); one handle serves a batch whose instances all have different heights (velocity, jerk-only
    and box forms, both kernels), each compared with the CPU restatement run at that height."""
    N, steps = 200, 20
    for form_name, B, mode in (("COUPLED_VEL", 24, 0), ("COUPLED_JERK", 8, 0), ("COUPLED_VEL", 6, 1)):
        bt = batches.coupled_batch(B, first=41000)
        height = 0.4 + 0.2 * batches.SplitMix(np.arange(B), seed=7).uniform()
        pl = P.BatchedPlanner(N, formulation=getattr(P, form_name), weight_R=0.01, weight_Q=1.0,
                              number_of_steps=steps, height=0.5, solver_mode=mode)
        o = _np(pl.solve_coupled(height=height, **bt))
        refs = [oracle.coupled_batch(oracle.params(N, R=0.01, Q=1.0, height=float(height[i]), iter_cap=4000),
                                     getattr(oracle, form_name), steps, 1000, bt["x0"][i:i + 1], bt["y0"][i:i + 1],
                                     bt["speed"][i:i + 1], bt["feet_x"][i:i + 1], bt["feet_y"][i:i + 1],
                                     bt["swing0"][i:i + 1]) for i in range(B)]
        ref = {k: np.concatenate([r[k] for r in refs]) for k in refs[0] if refs[0][k] is not None}
        _check_coupled(o, ref, B)
        # and the heights matter: the same batch at the handle's height gives another plan
        o5 = _np(pl.solve_coupled(**bt))
        assert _rel(o5["jerk_x"], o["jerk_x"]) > 1e-4
        same = _np(pl.solve_coupled(height=np.full(B, 0.5), **bt))
        assert _rel(same["jerk_x"], o5["jerk_x"]) < 1e-9 and np.array_equal(same["active_idx"], o5["active_idx"])
    # box form, through both of its kernels (the infeasible draws go to the dense one)
    Bb, Nb = 96, 100
    bb = batches.decoupled_batch(Bb, first=500)
    hb = 0.4 + 0.2 * batches.SplitMix(np.arange(Bb), seed=9).uniform()
    pb = P.BatchedPlanner(Nb, formulation=P.BOX_1D, weight_R=1.0)
    ob = _np(pb.solve_box(bb["y0"], bb["ymin"], bb["ymax"], height=hb))
    ox = _np(pb.solve_box(bb["x0"], bb["xmin"], bb["xmax"], height=hb))
    for i in range(Bb):
        rb = oracle.box_batch(oracle.params(Nb, R=1.0, height=float(hb[i]), iter_cap=3000), bb["x0"][i:i + 1],
                              bb["y0"][i:i + 1], bb["xmin"][i:i + 1], bb["xmax"][i:i + 1], bb["ymin"][i:i + 1],
                              bb["ymax"][i:i + 1])
        for ax, og, key in ((0, ox, "jerk_x"), (1, ob, "jerk_y")):
            if rb["iters"][0, ax] > Nb:
                continue  # numerically degenerate draw: see _check_box_axis
            assert og["status"][i] == rb["status"][0, ax]
            if rb["status"][0, ax] == 0:
                na = rb["n_active"][0, ax]
                assert og["iters"][i] == rb["iters"][0, ax]
                assert list(og["active_idx"][i, :og["n_active"][i]]) == list(rb["active"][0, ax, :na])
                assert _rel(og["jerk_x"][i], rb[key][0]) < RTOL
    # slack form (its Hessian does not depend on the height either; dense kernel only): heights that
    # differ from the handle's own, against solveQPconstraintSlack run at each height
    Bs, Ns = 10, 60
    bs = batches.decoupled_batch(Bs, N=Ns, steps=6, first=900)
    hs = 0.4 + 0.2 * batches.SplitMix(np.arange(Bs), seed=11).uniform()
    ps = P.BatchedPlanner(Ns, formulation=P.BOX_SLACK, weight_R=1.0, weight_Qs=50.0, active_stride=4 * Ns, height=0.5)
    osl = _np(ps.solve_box(bs["y0"], bs["ymin"], bs["ymax"], height=hs, outputs=P.SLACK_OUTPUTS))
    for i in range(Bs):
        rs = oracle.slack_batch(oracle.params(Ns, R=1.0, Qs=50.0, height=float(hs[i]), iter_cap=50 * 7 * Ns),
                                bs["y0"][i:i + 1], bs["ymin"][i:i + 1], bs["ymax"][i:i + 1], active_stride=4 * Ns)
        na = rs["n_active"][0]
        assert osl["status"][i] == rs["status"][0] and osl["iters"][i] == rs["iters"][0]
        assert list(osl["active_idx"][i, :na]) == list(rs["active"][0, :na])
        assert _rel(osl["jerk_x"][i], rs["jerk"][0]) < RTOL and _rel(osl["slacks"][i], rs["slacks"][0]) < RTOL
        assert abs(osl["cost"][i] - rs["cost"][0]) <= RTOL * abs(rs["cost"][0])
    at_half = _np(ps.solve_box(bs["y0"], bs["ymin"], bs["ymax"], outputs=P.SLACK_OUTPUTS))
    assert _rel(at_half["jerk_x"], osl["jerk_x"]) > 1e-4  # the heights matter
    ref_handle = P.BatchedPlanner(N, formulation=P.COUPLED_REF, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    with pytest.raises(P.MpcError, match="COUPLED_REF"):
        ref_handle.solve_coupled(height=np.full(4, 0.5), **batches.coupled_batch(4))


def test_zmp_reference_per_knot_vs_oracle(P, oracle):
    """solveQPConstraintCoupledRef takes any zmp_ref vectors (MPCPlanner.cpp:632-640, 728-735); the
    per-knot reference input against the CPU restatement, with weights that make constraints bind,
    and against the per-phase input where the reference happens to be constant over the phases."""
    B, N, steps, Qs = 10, 200, 20, 1.0
    bt = batches.coupled_batch(B, first=51000)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_REF, weight_R=0.01, weight_Q=1.0, weight_Qs=Qs,
                          number_of_steps=steps)
    k = np.arange(N)
    zk = np.zeros((B, N, 2))
    p = oracle.params(N, R=0.01, Q=1.0, Qs=Qs, height=0.5, iter_cap=4000)
    refs = []
    for i in range(B):
        st = oracle.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i], bt["swing0"][i])
        cx, cy = oracle.centroid(st["fx"], st["fy"], st["swing"])
        zk[i, :, 0] = cx + 0.03 * np.sin(0.05 * k + i)
        zk[i, :, 1] = cy + 0.2 * np.cos(0.11 * k) * (1 + 0.1 * i)
        refs.append(oracle.solve_coupled(p, oracle.COUPLED_REF, bt["x0"][i], bt["y0"][i], st["A"], st["b"],
                                         bt["speed"][i], 1000, zk[i, :, 0], zk[i, :, 1]))
    for mode in (0, 1):
        plm = pl if mode == 0 else P.BatchedPlanner(N, formulation=P.COUPLED_REF, weight_R=0.01, weight_Q=1.0,
                                                    weight_Qs=Qs, number_of_steps=steps, solver_mode=1)
        o = _np(plm.solve_coupled(zref_knot=zk, **bt))
        assert max(r["info"].n_active for r in refs) > 0
        for i, r in enumerate(refs):
            assert o["status"][i] == r["info"].status and o["iters"][i] == r["info"].iters
            assert list(o["active_idx"][i, :o["n_active"][i]]) == list(r["active"])
            assert _rel(o["jerk_x"][i], r["jerk_x"]) < RTOL and _rel(o["jerk_y"][i], r["jerk_y"]) < RTOL
            assert abs(o["cost"][i] - r["info"].f) <= RTOL * abs(r["info"].f)
    # a per-knot reference that is constant over each phase = the per-phase input
    plain = _np(pl.solve_coupled(**bt))  # centroid reference, built in the kernel
    cz = np.zeros((B, N, 2))
    for i in range(B):
        st = oracle.compute_steps(N, steps, bt["speed"][i], bt["feet_x"][i], bt["feet_y"][i], bt["swing0"][i])
        cz[i, :, 0], cz[i, :, 1] = oracle.centroid(st["fx"], st["fy"], st["swing"])
    byknot = _np(pl.solve_coupled(zref_knot=cz, **bt))
    assert np.array_equal(byknot["active_idx"], plain["active_idx"])
    assert _rel(byknot["jerk_x"], plain["jerk_x"]) < 1e-8 and _rel(byknot["cost"], plain["cost"]) < 1e-8
    with pytest.raises(P.MpcError, match="COUPLED_REF"):
        P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, number_of_steps=steps).solve_coupled(
            zref_knot=zk, **bt)


def test_dispatch_hint_reorders_the_launch_not_the_results(P):
    """mpcb_set_dispatch_hint(LONGEST_FIRST): the first stage starts the instances in the order of
    the record the previous solve left.  Whatever that record is -- none, one of another batch,
    one of another batch size, one with hand-overs -- every instance is solved exactly once and
    every output is bit-identical to the index-order solve."""
    import torch

    def run(pl, prepare, solve, cases):
        plain = []
        for c in cases:
            plain.append({k: v.cpu().numpy() for k, v in solve(prepare(c)).items()})
        pl.set_dispatch_hint(P.DISPATCH_LONGEST_FIRST)
        for rounds in range(2):  # second round: each case ordered by the record of the one before
            for c, ref in zip(cases, plain):
                prepared = prepare(c)
                B = next(iter(ref.values())).shape[0]
                out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
                for t in out[0].values():  # an instance the permutation skipped would show
                    t.fill_(-7)
                torch.cuda.synchronize()
                got = solve(prepared, out=out)
                torch.cuda.synchronize()
                for k, v in ref.items():
                    assert np.array_equal(got[k].cpu().numpy(), v, equal_nan=True), (c, k, rounds)
        pl.set_dispatch_hint(P.DISPATCH_IN_ORDER)
        c, ref = cases[0], plain[0]
        got = solve(prepare(c))
        for k, v in ref.items():
            assert np.array_equal(got[k].cpu().numpy(), v, equal_nan=True), (c, k, "in order again")

    N, steps = 200, 20
    # (batch size, first instance): same size twice with different instances, a batch of 2,500
    # (more than one workgroup of the ordering kernel, not a multiple of its 256 threads), a
    # smaller one after it, a single instance
    cases = [(700, 0), (700, 5000), (2500, 100), (333, 40), (1, 9)]
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    run(pl, lambda c: pl.prepare_coupled(**batches.coupled_batch(c[0], first=c[1])),
        lambda prepared, out=None: pl.solve_coupled_prepared(prepared, out=out), cases)
    # working sets of 33 to 53 rows: part of every batch goes on to the wide edition (the longest bin)
    hard = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=1e-6, weight_Q=1.0, number_of_steps=steps)
    run(hard, lambda c: hard.prepare_coupled(**batches.coupled_batch(c[0], first=c[1])),
        lambda prepared, out=None: hard.solve_coupled_prepared(prepared, out=out), [(48, 0), (48, 300), (20, 7)])
    assert hard.last_wide_count() > 0
    # box form: the dense second stage takes over mid-solve from the records of the first
    box = P.BatchedPlanner(100, formulation=P.BOX_1D, weight_R=1.0)

    def prepare_box(c):
        bt = batches.decoupled_batch(c[0], first=c[1])
        return box.prepare_box(bt["x0"], bt["xmin"], bt["xmax"])

    run(box, prepare_box, lambda prepared, out=None: box.solve_box_prepared(prepared, out=out),
        [(600, 0), (600, 2000), (90, 11)])
    assert box.last_handover_count() > 0


def test_nearly_dependent_rows_at_the_solvers_thresholds(P, oracle):
    """Caller-built rows that put the solver on its z != 0 test (eiquadprog.hpp:336) and on the
    thresholds of the thin-factor kernel: an active row is moved 5 cm inwards and a second row of
    the same knot is replaced by its mirror image turned by theta, closing a slab of the given
    width behind it (negative: the two half-planes only meet far away, or -- theta = 0 -- never).

    theta = 0 and 1e-9: the second normal is exactly / numerically dependent on the first, the
    reference takes its dual-step-only branch and reports infeasibility after a few iterations.
    theta = 1e-4: the pair can be added (z . n / n . G^-1 n ~ 3e-9) but leaves a factor with a
    condition number of ~1e4 that the reference's own arithmetic just survives: the first-stage
    kernel passes such an instance to the dense kernel.  theta = 1e-2: regular.  Status and
    iteration count equal the CPU restatement's in all of them, the ordered working set and the
    jerk too where a solution exists.  (With theta between 1e-7 and 1e-5 and an empty slab the
    reference itself wanders for ~430 iterations until 399 of the 400 possible rows are active
    and returns an arbitrary point or 'infeasible' depending on rounding: not compared.)"""
    N, steps = 200, 20
    bt = batches.coupled_batch(1, first=13000)
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    st = oracle.compute_steps(N, steps, bt["speed"][0], bt["feet_x"][0], bt["feet_y"][0], bt["swing0"][0])
    A, b = st["A"], st["b"]
    m = A.shape[0]
    kn = np.array([np.nonzero(A[r])[0][0] % N for r in range(m)])
    rows0 = np.array([(A[r, kn[r]], A[r, kn[r] + N], b[r]) for r in range(m)])
    base = _np(pl.solve_coupled(bt["x0"], bt["y0"], bt["speed"], rows_pqr=rows0[None]))
    act = base["active_idx"][0, :base["n_active"][0]]
    ra = int(act[len(act) // 2])
    k = kn[ra]
    victim = [r for r in range(m) if kn[r] == k and r != ra][0]
    p, q, r = rows0[ra]
    shift = 0.05
    cases = [(w, th) for w in (-0.05, -0.01) for th in (0.0, 1e-9, 1e-4, 1e-2)]
    cases += [(w, th) for w in (0.0, 1e-6, 1e-3) for th in (0.0, 1e-9, 1e-7, 1e-5, 1e-4, 1e-2)]
    allrows = []
    for width, theta in cases:
        c, s = np.cos(theta), np.sin(theta)
        rows = rows0.copy()
        rows[ra] = (p, q, r - shift)
        rows[victim] = (-(c * p - s * q), -(s * p + c * q), -(r - shift) + width)
        allrows.append(rows)
    allrows = np.array(allrows)
    rep = lambda a: np.repeat(a, len(cases), axis=0)
    o = _np(pl.solve_coupled(rep(bt["x0"]), rep(bt["y0"]), rep(bt["speed"]), rows_pqr=allrows))
    assert pl.last_handover_count() >= 2  # the theta = 1e-4 pairs
    par = oracle.params(N, R=0.01, Q=1.0, height=0.5, iter_cap=4000)
    solved = 0
    for i, (width, theta) in enumerate(cases):
        A2 = A.copy()
        A2[victim] = 0.0
        A2[victim, k], A2[victim, k + N] = allrows[i, victim, 0], allrows[i, victim, 1]
        ref = oracle.solve_coupled(par, oracle.COUPLED_VEL, bt["x0"][0], bt["y0"][0], A2, allrows[i, :, 2],
                                   bt["speed"][0], 1000)
        assert ref["info"].status == o["status"][i], (width, theta)
        assert ref["info"].iters == o["iters"][i], (width, theta)
        if ref["info"].status == 0:
            solved += 1
            assert list(ref["active"]) == list(o["active_idx"][i, :o["n_active"][i]]), (width, theta)
            assert _rel(o["jerk_x"][i], ref["jerk_x"]) < RTOL and _rel(o["jerk_y"][i], ref["jerk_y"]) < RTOL
    assert solved >= 18 and (o["status"][:3] == 1).all()


def test_dispatch_hint_in_the_replanning_sweep_and_the_host_entry_point(P):
    """With a dispatch hint k every stage of a sweep is ordered by the k-th stage before it, and the
    host-pointer entry point by the handle's k-th previous solve: per-stage records, final plans
    and evolved states are those of the index-order sweep, bit for bit."""
    import torch
    B, N, steps, stages = 300, 200, 20, 4
    inputs = batches.coupled_batch(B, first=7000)
    dist = batches.replanning_disturbance(B, first=7000)

    def sweep(hint):
        pl = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0,
                              replanning_window=N // steps, number_of_steps=steps,
                              hip_offsets=batches.NOMINAL_HIP_OFFSETS)
        pl.set_dispatch_hint(hint)
        prepared = pl.prepare_replanning(disturbance=dist, **inputs)
        res, stage, state = pl.replan_sweep(prepared, stages)
        torch.cuda.synchronize()
        out = {k: v.cpu().numpy() for k, v in res.items()}
        out.update({"stage_" + k: v.cpu().numpy() for k, v in stage.items()})
        out.update({"state_" + k: state[k].cpu().numpy() for k in ("x0", "y0", "feet_x", "feet_y", "swing0")})
        return out

    plain = sweep(P.DISPATCH_IN_ORDER)
    # lag 1: by the stage before; lag 3: a ring of three records, stage s ordered by stage s - 3
    for hint in (P.DISPATCH_LONGEST_FIRST, 3):
        ordered = sweep(hint)
        assert set(plain) == set(ordered)
        for k, v in plain.items():
            assert np.array_equal(ordered[k], v, equal_nan=True), (hint, k)

    pl = P.BatchedPlanner(N, formulation=P.COUPLED_JERK, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    args = [inputs[k] for k in ("x0", "y0", "speed", "feet_x", "feet_y", "swing0")]
    ref = {k: np.array(v) for k, v in pl.solve_coupled_host(*args).items()}
    for hint in (P.DISPATCH_LONGEST_FIRST, 2, 8, P.DISPATCH_LONGEST_FIRST):  # changing the lag drops the records
        pl.set_dispatch_hint(hint)
        for _ in range(3):
            got = pl.solve_coupled_host(*args)
            for k, v in ref.items():
                assert np.array_equal(np.array(got[k]), v, equal_nan=True), (hint, k)
    with pytest.raises(P.MpcError, match="out of range"):
        pl.set_dispatch_hint(9)

"""Several handles in one call: shards on several devices (device pointers), parameter groups on
one device (mixed horizons, weights, heights), and the tables those handles are built from."""
import numpy as np
import pytest

from mpc_amd import batches

pytestmark = pytest.mark.gpu

RTOL = 1e-6


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


def test_multi_device_pointer_entry_matches_per_handle_calls(P):
    """mpcb_solve_coupled_multi_device with two handles (both on the one GPU of the test box, each
    with a stream of its own) gives bit for bit what two mpcb_solve_coupled_device calls give; the
    shards may differ in size and the handles in shape."""
    import torch
    pa = P.BatchedPlanner(200, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=20)
    pb = P.BatchedPlanner(100, formulation=P.COUPLED_JERK, weight_R=0.01, number_of_steps=10)
    ba, bb = batches.coupled_batch(70, first=100), batches.coupled_batch(33, first=900)
    prep = [pa.prepare_coupled(**ba), pb.prepare_coupled(**bb)]
    one = [_np(pa.solve_coupled_prepared(prep[0])), _np(pb.solve_coupled_prepared(prep[1]))]
    torch.cuda.synchronize()
    outs = [pa._alloc_outputs(70, P.ALL_OUTPUTS), pb._alloc_outputs(33, P.ALL_OUTPUTS)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in (streams, None):
        for o in outs:
            for t in o[0].values():
                t.zero_()
        torch.cuda.synchronize()
        res = P.BatchedPlanner.solve_coupled_multi_device([pa, pb], prep, outs, streams=st)
        torch.cuda.synchronize()
        for g in range(2):
            got = _np(res[g])
            for k, v in one[g].items():
                assert np.array_equal(got[k], v, equal_nan=True), (g, k)
            assert (got["status"] == 0).all()
    with pytest.raises(P.MpcError, match="twice"):
        P.BatchedPlanner.solve_coupled_multi_device([pa, pa], [prep[0], prep[0]], [outs[0], outs[0]])


def test_groups_mix_horizons_and_weights_in_one_call(P, oracle):
    """mpcb_solve_coupled_groups: one call, five groups with horizons 100 / 150 / 200 and different
    weights and formulations, spread over the caller's stream and the library's two; every group
    is compared with the CPU restatement run with that group's parameters."""
    import torch
    specs = [dict(N=100, steps=10, form="COUPLED_VEL", R=0.01, rw=1000, B=40),
             dict(N=150, steps=15, form="COUPLED_VEL", R=0.02, rw=1000, B=25),
             dict(N=200, steps=20, form="COUPLED_VEL", R=0.01, rw=10, B=31),
             dict(N=200, steps=20, form="COUPLED_JERK", R=0.01, rw=1000, B=17),
             dict(N=150, steps=10, form="COUPLED_REF", R=0.01, rw=1000, B=9)]
    pls, prep, outs, bts = [], [], [], []
    for g, s in enumerate(specs):
        pl = P.BatchedPlanner(s["N"], formulation=getattr(P, s["form"]), weight_R=s["R"], weight_Q=1.0,
                              weight_Qs=100.0, replanning_window=s["rw"], number_of_steps=s["steps"])
        bt = batches.coupled_batch(s["B"], first=40000 + 1000 * g)
        pls.append(pl)
        bts.append(bt)
        prep.append(pl.prepare_coupled(**bt))
        outs.append(pl._alloc_outputs(s["B"], P.ALL_OUTPUTS))
    st = torch.cuda.Stream()
    for rep in range(2):  # twice: the second call reuses lanes, events and workspaces
        res = P.BatchedPlanner.solve_coupled_groups(pls, prep, outs, stream=st)
    st.synchronize()
    for g, s in enumerate(specs):
        bt = bts[g]
        ref = oracle.coupled_batch(oracle.params(s["N"], R=s["R"], Q=1.0, Qs=100.0, height=0.5, iter_cap=4000),
                                   getattr(oracle, s["form"]), s["steps"], s["rw"], bt["x0"], bt["y0"],
                                   bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"], nthreads=16)
        o = _np(res[g])
        assert (o["status"] == ref["status"]).all(), g
        assert (o["iters"] == ref["iters"]).all(), g
        for i in range(s["B"]):
            na = ref["n_active"][i]
            assert list(o["active_idx"][i, :na]) == list(ref["active"][i, :na]), (g, i)
            for k in ("jerk_x", "jerk_y", "com_x", "com_y", "zmp_x", "zmp_y"):
                assert _rel(o[k][i], ref[k][i]) < RTOL, (g, k, i)
        assert (np.abs(o["cost"] - ref["cost"]) <= RTOL * np.abs(ref["cost"])).all(), g
    # the opt-in spread over the library's two streams (one synchronised call: see the header on why
    # it is not the default) gives the same results bit for bit
    import os
    plain = [_np(r) for r in res]
    os.environ["MPCB_GROUP_LANES"] = "2"
    try:
        torch.cuda.synchronize()
        res2 = P.BatchedPlanner.solve_coupled_groups(pls, prep, outs, stream=st)
        torch.cuda.synchronize()
    finally:
        del os.environ["MPCB_GROUP_LANES"]
    for g in range(len(specs)):
        for k, v in _np(res2[g]).items():
            assert np.array_equal(v, plain[g][k], equal_nan=True), (g, k)
    other = P.BatchedPlanner(50, formulation=P.BOX_1D, weight_R=1.0)
    with pytest.raises(P.MpcError, match="coupled"):
        P.BatchedPlanner.solve_coupled_groups([pls[0], other], [prep[0], prep[1]], [outs[0], outs[1]])


# ---- the tables themselves: built on the GPU, checked against the host construction and the
# ---- CPU restatement's buildMatrix products

_TABLE_CASES = [
    dict(horizon=200, formulation="COUPLED_VEL", weight_R=0.01, weight_Q=1.0, number_of_steps=20),
    dict(horizon=200, formulation="COUPLED_VEL", weight_R=1e-6, weight_Q=1.0, number_of_steps=20),
    dict(horizon=200, formulation="COUPLED_VEL", weight_R=0.01, weight_Q=1.0, replanning_window=10, number_of_steps=20),
    dict(horizon=200, formulation="COUPLED_REF", weight_R=0.01, weight_Q=1.0, weight_Qs=100.0, number_of_steps=20),
    dict(horizon=150, formulation="COUPLED_REF", weight_R=0.01, weight_Q=1.0, weight_Qs=0.1, number_of_steps=10,
         height=0.43),
    dict(horizon=200, formulation="COUPLED_JERK", weight_R=0.01, number_of_steps=20),
    dict(horizon=100, formulation="BOX_1D", weight_R=1.0),
    dict(horizon=100, formulation="UNCONSTRAINED", weight_R=1e-6, weight_Q=1.0),
    dict(horizon=64, formulation="BOX_SLACK", weight_R=0.5, weight_Qs=10.0),
    dict(horizon=300, formulation="COUPLED_VEL", weight_R=0.01, weight_Q=1.0, number_of_steps=15),
    dict(horizon=3, formulation="COUPLED_VEL", weight_R=0.01, weight_Q=1.0, number_of_steps=1),
]
_TABLE_IDS = {0: "S", 1: "Wt", 2: "Sc", 3: "Sv", 4: "Kz", 5: "Kx", 6: "zu_col", 7: "zx", 8: "xp_col", 9: "xpx",
              10: "xv_col", 11: "xvx", 12: "J0", 13: "F", 14: "Sp", 15: "Wp", 16: "F2", 17: "F3", 18: "Spp",
              19: "Sps", 20: "Saa", 21: "Wtp", 22: "Wta", 23: "Kzp", 24: "Kza"}


def _spec(P, c, **more):
    s = dict(c, **more)
    s["formulation"] = getattr(P, s["formulation"])
    return s


def test_device_tables_against_the_decimal_reference(P):
    """The tables built on the GPU against tests/golden/table_reference.npz (60-digit decimal
    arithmetic on the model's formulas, make_table_reference.py): every entry within 4 units in the
    last place of the table's largest entry -- i.e. correctly rounded up to the final rounding --
    also where cond(G1) ~ 1e10 (weight_R = 1e-6, weight_Qs = 100), where the host long-double
    construction is off by 1e-12 (tests/test_oracle.py::test_host_tables_against_the_decimal_reference)."""
    from table_golden import golden_cases, TABLE_KEYS
    cases = golden_cases(P)
    pls = P.BatchedPlanner.create_groups([kw for _, kw, _ in cases])
    worst = 0.0
    for (name, kw, ref), pl in zip(cases, pls):
        for key, which in TABLE_KEYS.items():
            if key not in ref:
                continue
            a, b = pl.matrix(which), ref[key]
            if key == "J0":
                a = a.T  # column-major on the device
            assert a.shape == b.shape, (name, key)
            scale = np.abs(b).max()
            err = np.abs(a - b).max() / scale if scale > 0 else np.abs(a).max()
            worst = max(worst, err)
            assert err <= 4 * 2.3e-16, (name, key, err)
    print("largest difference of a GPU-built table from the decimal reference: %.2e of its largest entry" % worst)


def test_device_tables_against_the_host_construction(P):
    """Every table of every formulation at full size, built by ONE mpcb_create_groups call on the
    GPU, against the long-double host construction of the same handle.  The host's 64-bit mantissa
    limits the agreement: 1e-9 of the table's largest entry is demanded (measured: 2e-11 at
    weight_R = 0.01, N = 200), and the prediction matrices, which involve no factorisation, must
    agree to the last bit or two."""
    dev = P.BatchedPlanner.create_groups([_spec(P, c) for c in _TABLE_CASES])
    worst = {}
    for c, pd in zip(_TABLE_CASES, dev):
        ph = P.BatchedPlanner(**_spec(P, c, table_build=P.TABLES_HOST))
        if pd.formulation in (P.BOX_1D, P.COUPLED_JERK, P.COUPLED_VEL):
            # the host builds the height-free parts on first use
            B = 2
            if pd.formulation == P.BOX_1D:
                bb = batches.decoupled_batch(B, N=c["horizon"])
                ph.solve_box(bb["y0"], bb["ymin"], bb["ymax"], height=np.full(B, 0.5))
            elif c["horizon"] >= 8:
                ph.solve_coupled(**dict(batches.coupled_batch(B), height=np.full(B, 0.5)))
        for which, name in _TABLE_IDS.items():
            try:
                a = pd.matrix(which)
            except P.MpcError:
                continue
            try:
                b = ph.matrix(which)
            except P.MpcError:
                continue
            assert a.shape == b.shape, (c, name)
            scale = np.abs(b).max()
            err = np.abs(a - b).max() / scale if scale > 0 else np.abs(a).max()
            worst[name] = max(worst.get(name, 0.0), err)
            assert err <= (5e-16 if 6 <= which <= 11 else 1e-9), (c, name, err)
        assert np.array_equal(pd.matrix(0), pd.matrix(0).T)  # S is stored exactly symmetric
    print("largest device/host difference per table (of the table's largest entry): " +
          " ".join("%s=%.1e" % kv for kv in sorted(worst.items())))


def test_device_tables_vs_the_oracles_build_matrix_products(P, oracle):
    """Independent of the library's own host construction: the prediction matrices are those of the
    CPU restatement's buildMatrix (MPCPlanner.cpp:101-156), and the response tables satisfy their
    defining equations with the Hessian assembled from those matrices in numpy long double
    (MPCPlanner.cpp:523-559, 723-739): G1 W = Zu', S = Zu W, G1 Kx = -Gm, J0' G1 J0 = I."""
    ld = np.longdouble
    for c in (dict(N=200, form="COUPLED_VEL", R=0.01, Q=1.0, Qs=100.0, rw=1000, steps=20, h=0.5),
              dict(N=100, form="COUPLED_VEL", R=0.01, Q=1.0, Qs=100.0, rw=10, steps=10, h=0.55),
              dict(N=120, form="COUPLED_REF", R=0.01, Q=1.0, Qs=1.0, rw=1000, steps=12, h=0.47)):
        N = c["N"]
        pl = P.BatchedPlanner(N, formulation=getattr(P, c["form"]), weight_R=c["R"], weight_Q=c["Q"],
                              weight_Qs=c["Qs"], replanning_window=c["rw"], number_of_steps=c["steps"],
                              height=c["h"])
        op = oracle.params(N, R=c["R"], Q=c["Q"], Qs=c["Qs"], height=c["h"])
        Zx, Zu = oracle.build_matrix(op, 3)  # 3 = Cz, 0 = Cx, 1 = Cv
        Xpx, Xpu = oracle.build_matrix(op, 0)
        Xvx, Xvu = oracle.build_matrix(op, 1)
        for which, ref in ((6, Zu[:, 0]), (7, Zx), (8, Xpu[:, 0]), (9, Xpx), (10, Xvu[:, 0]), (11, Xvx)):
            got = pl.matrix(which)
            assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), which
        for k in range(1, N):  # Toeplitz: sub-diagonal k of the reference's matrix is constant
            assert np.ptp(np.diagonal(Zu, -k)) <= 1e-13
        w = np.full(N, c["Q"]) if c["rw"] == 1000 else oracle.make_gaussian(N, c["rw"], c["rw"]) * c["Q"] * N
        G1 = c["R"] * np.eye(N, dtype=ld) + Xvu.astype(ld).T @ (w.astype(ld)[:, None] * Xvu.astype(ld))
        Gm = np.concatenate([Xvu.astype(ld).T @ (w.astype(ld)[:, None] * Xvx.astype(ld)),
                             -(Xvu.astype(ld).T @ w.astype(ld))[:, None]], axis=1)
        if c["form"] == "COUPLED_REF":
            G1 = G1 + c["Qs"] * Zu.astype(ld).T @ Zu.astype(ld)
            Gm[:, :3] += c["Qs"] * Zu.astype(ld).T @ Zx.astype(ld)
        W = pl.matrix(1).T.astype(ld)      # table 1 holds W^T
        S, Kx, Kz = pl.matrix(0).astype(ld), pl.matrix(5).astype(ld), pl.matrix(4).astype(ld)
        J0 = pl.matrix(12).T.astype(ld)    # column-major on the device
        gn = np.abs(G1).max()
        assert np.abs(G1 @ W - Zu.T).max() <= 1e-12 * gn * np.abs(W).max()
        assert np.abs(Zu.astype(ld) @ W - S).max() <= 1e-13 * np.abs(S).max()
        assert np.abs(G1 @ Kx + Gm).max() <= 1e-12 * gn * np.abs(Kx).max()
        assert np.abs(Zu.astype(ld) @ Kx[:, :3] + Zx - Kz[:, :3]).max() <= 1e-13 * np.abs(Kz).max()
        assert np.abs(J0.T @ G1 @ J0 - np.eye(N)).max() <= 1e-10
        assert np.allclose(np.triu(J0), J0)


def test_create_time_device_against_host(P):
    """mpcb_create at N = 200 with the tables built on the GPU and on the host (reported)."""
    import time
    kw = dict(formulation=P.COUPLED_REF, weight_R=0.01, weight_Q=1.0, weight_Qs=100.0, number_of_steps=20)
    P.BatchedPlanner(200, **kw)  # first use of the kernels
    t = time.perf_counter()
    P.BatchedPlanner(200, **kw)
    t_dev = time.perf_counter() - t
    t = time.perf_counter()
    P.BatchedPlanner(200, table_build=P.TABLES_HOST, **kw)
    t_host = time.perf_counter() - t
    t = time.perf_counter()
    P.BatchedPlanner.create_groups([dict(horizon=200, height=0.4 + 0.002 * k, **kw) for k in range(96)])
    t_many = time.perf_counter() - t
    print("mpcb_create N=200 COUPLED_REF: device %.1f ms, host %.1f ms; 96 handles in one mpcb_create_groups "
          "call %.1f ms" % (t_dev * 1e3, t_host * 1e3, t_many * 1e3))
    assert t_dev < t_host


def test_zmp_reference_form_with_a_height_per_instance(P, oracle):
    """COUPLED_REF with CoM heights ~U[0.4, 0.6] per instance through the host-pointer entry point
    (the library groups on distinct heights: here every instance is a group of its own, plus a few
    repeated heights) against the CPU restatement run at each instance's height."""
    B, N, steps = 28, 200, 20
    bt = batches.coupled_batch(B, first=52000)
    rng = batches.SplitMix(np.arange(B), seed=99)
    height = 0.4 + 0.2 * rng.uniform()
    height[5] = height[17] = height[2]     # one group of three
    height[9] = 0.5                        # the handle's own height
    pl = P.BatchedPlanner(N, formulation=P.COUPLED_REF, weight_R=0.01, weight_Q=1.0, weight_Qs=100.0,
                          number_of_steps=steps, height=0.5)
    for rep in range(2):  # the second call finds every height's handle
        o = pl.solve_coupled_host(bt["x0"], bt["y0"], bt["speed"], bt["feet_x"], bt["feet_y"], bt["swing0"],
                                  height=height)
    for i in range(B):
        ref = oracle.coupled_batch(oracle.params(N, R=0.01, Q=1.0, Qs=100.0, height=float(height[i]), iter_cap=4000),
                                   oracle.COUPLED_REF, steps, 1000, bt["x0"][i:i + 1], bt["y0"][i:i + 1],
                                   bt["speed"][i:i + 1], bt["feet_x"][i:i + 1], bt["feet_y"][i:i + 1],
                                   bt["swing0"][i:i + 1], nthreads=1)
        assert o["status"][i] == ref["status"][0] and o["iters"][i] == ref["iters"][0]
        na = ref["n_active"][0]
        assert list(o["active_idx"][i, :na]) == list(ref["active"][0, :na])
        for k in ("jerk_x", "jerk_y", "com_x", "com_y", "zmp_x", "zmp_y"):
            assert _rel(o[k][i], ref[k][0]) < RTOL, (k, i)
        assert abs(o["cost"][i] - ref["cost"][0]) <= RTOL * abs(ref["cost"][0])
    # the device-pointer entry point cannot look at the heights: it says where to go instead
    with pytest.raises(P.MpcError, match="mpcb_create_groups"):
        pl.solve_coupled(**dict(bt, height=height))


def test_unconstrained_form_with_a_height_per_instance(P, oracle):
    """solveQP (MPCPlanner.cpp:223-237) with `actual_height` per instance: grouped on distinct
    heights by the library, against the CPU restatement run at each height.  weight_R / weight_Q =
    1e-6 as in the reference driver makes the jerk ill-conditioned (cond(G) ~ 1e10: jerk to 1e-3,
    as in test_unconstrained_solve_qp), the ZMP that is being tracked agrees to 1e-6."""
    B, N = 20, 100
    rng = batches.SplitMix(np.arange(B), seed=5)
    x0 = np.stack([rng.uniform(0.0, 0.2), rng.uniform(-0.1, 0.1), np.zeros(B)], axis=1)
    zref = np.linspace(0, 0.2, N)[None, :] * (1.0 + rng.uniform(-0.3, 0.3))[:, None]
    height = 0.4 + 0.2 * rng.uniform()
    height[3] = height[11] = 0.5
    pl = P.BatchedPlanner(N, formulation=P.UNCONSTRAINED, weight_R=1e-6, weight_Q=1.0, height=0.5)
    every = ("jerk_x", "zmp_x", "com_x", "com_xd", "status", "iters", "n_active")
    o = pl.solve_unconstrained_host(x0, zref, height=height, outputs=every)
    plain = pl.solve_unconstrained_host(x0, zref, outputs=every)
    assert (o["status"] == 0).all()
    # solveQP iterates over nothing: both paths report no iterations and an empty working set
    for r in (o, plain):
        assert (r["iters"] == 0).all() and (r["n_active"] == 0).all()
    for i in range(B):
        p = oracle.params(N, R=1e-6, Q=1.0, height=float(height[i]))
        jerk = oracle.solve_qp(p, x0[i], zref[i])
        zmp = oracle.trajectory(p, 3, x0[i], jerk)
        assert _rel(o["zmp_x"][i], zmp) < RTOL, i
        assert _rel(o["jerk_x"][i], jerk) < 1e-3, i
        if height[i] == 0.5:
            assert np.array_equal(o["jerk_x"][i], plain["jerk_x"][i])
    assert _rel(o["jerk_x"][0], plain["jerk_x"][0]) > 1e-4  # the heights matter


def test_height_groups_beyond_the_handle_cache(P):
    """A handle keeps the tables of at most 1,024 heights it was asked for.  Three batches of 700
    distinct heights each (the third repeats the first) push it over that limit twice: the tables
    used longest ago are dropped, rebuilt when they are asked for again, and every instance still
    gets the result of a planner created at its own height.  More than 1,024 distinct heights in
    ONE batch are refused."""
    N, B = 16, 700
    rng = batches.SplitMix(np.arange(3 * B), seed=11)
    x0 = np.stack([rng.uniform(0.0, 0.2), rng.uniform(-0.1, 0.1), np.zeros(3 * B)], axis=1)
    zref = np.linspace(0, 0.1, N)[None, :] * (1.0 + rng.uniform(-0.3, 0.3))[:, None]
    heights = 0.40 + 0.0001 * np.arange(2 * B)
    pl = P.BatchedPlanner(N, formulation=P.UNCONSTRAINED, weight_R=1e-4, weight_Q=1.0, height=0.5)
    rounds = [(slice(0, B), heights[:B]), (slice(B, 2 * B), heights[B:]), (slice(2 * B, 3 * B), heights[:B])]
    for sl, hh in rounds:
        o = pl.solve_unconstrained_host(x0[sl], zref[sl], height=hh)
        assert (o["status"] == 0).all()
        for i in (0, 1, B // 2, B - 1):
            single = P.BatchedPlanner(N, formulation=P.UNCONSTRAINED, weight_R=1e-4, weight_Q=1.0, height=float(hh[i]))
            ref = single.solve_unconstrained_host(x0[sl][i:i + 1], zref[sl][i:i + 1])
            assert np.array_equal(o["jerk_x"][i], ref["jerk_x"][0]), i
            assert np.array_equal(o["zmp_x"][i], ref["zmp_x"][0]), i
    with pytest.raises(P.MpcError, match="distinct heights"):
        pl.solve_unconstrained_host(np.tile(x0[:1], (1100, 1)), np.tile(zref[:1], (1100, 1)),
                                    height=0.40 + 0.0001 * np.arange(1100))


def test_groups_edge_cases_empty_groups_and_a_bad_parameter_set(P):
    """An empty group between two others changes nothing; a parameter set that cannot be built
    fails the whole mpcb_create_groups call, names the set and leaves no handle behind."""
    import torch
    pls = P.BatchedPlanner.create_groups([
        dict(horizon=100, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=10),
        dict(horizon=60, formulation=P.COUPLED_JERK, weight_R=0.02, number_of_steps=6),
        dict(horizon=150, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=15)])
    bts = [batches.coupled_batch(12, first=100), batches.coupled_batch(1, first=200), batches.coupled_batch(7, first=300)]
    bts[1] = {k: v[:0] for k, v in bts[1].items()}  # nothing to solve for the middle group
    prep = [pl.prepare_coupled(**bt) for pl, bt in zip(pls, bts)]
    assert prep[1][0] == 0
    outs = [pl._alloc_outputs(p[0], P.ALL_OUTPUTS) for pl, p in zip(pls, prep)]
    res = P.BatchedPlanner.solve_coupled_groups(pls, prep, outs)
    torch.cuda.synchronize()
    for g in (0, 2):
        alone = _np(pls[g].solve_coupled_prepared(prep[g]))
        got = _np(res[g])
        assert (got["status"] == 0).all()
        for k, v in alone.items():
            assert np.array_equal(got[k], v, equal_nan=True), (g, k)
    with pytest.raises(P.MpcError, match="parameter set 1"):
        P.BatchedPlanner.create_groups([
            dict(horizon=100, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=10),
            dict(horizon=100, formulation=P.COUPLED_VEL, weight_R=-1.0, weight_Q=1.0, number_of_steps=10)])

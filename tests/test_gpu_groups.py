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
    other = P.BatchedPlanner(50, formulation=P.BOX_1D, weight_R=1.0)
    with pytest.raises(P.MpcError, match="coupled"):
        P.BatchedPlanner.solve_coupled_groups([pls[0], other], [prep[0], prep[1]], [outs[0], outs[1]])

"""The entry point that allocates nothing (static device scratch instead of the handle's
hipMalloc'd lists and dense workspace) returns, bit for bit, what mpcb_solve_coupled_device
returns -- on two calls in a row, the second with a smaller batch, with instances going through
the wide edition of the first-stage kernel (the scratch's second list), through the block edition
behind it (the third list) and through the dense second stage (the scratch's J / R slots)."""
import numpy as np
import pytest

from mpc_amd import batches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from mpc_amd import planner
    planner.load_library()
    return planner


def _compare(P, pl, B, first, expect_handover, expect_wide=False, speed_scale=1.0, expect_block=False):
    import torch
    bt = batches.coupled_batch(B, first=first)
    bt["speed"] = bt["speed"] * speed_scale
    prepared = pl.prepare_coupled(**bt)
    ref = {k: v.cpu().numpy() for k, v in pl.solve_coupled_prepared(prepared).items()}
    handed, wide = pl.last_handover_count(), pl.last_wide_count()
    assert (pl.last_block_count() > 0) == expect_block
    out = pl._alloc_outputs(B, P.ALL_OUTPUTS)
    for t in out[0].values():  # whatever the call does not write must show up
        t.fill_(-7)
    torch.cuda.synchronize()
    got = pl.solve_coupled_prepared(prepared, out=out, no_alloc=True)
    torch.cuda.synchronize()
    for k, v in ref.items():
        assert np.array_equal(got[k].cpu().numpy(), v, equal_nan=True), (B, k)
    assert (handed > 0) == expect_handover and (wide > 0) == expect_wide
    return handed + wide


def test_no_alloc_entry_point_matches_allocating_one(P):
    N, steps = 200, 20
    # weight_R = 1e-6: working sets of 33 to 53 rows, so part of each batch goes through the wide
    # edition of the first-stage kernel, whose work list is the scratch's
    hard = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=1e-6, weight_Q=1.0, number_of_steps=steps)
    h1 = _compare(P, hard, 40, 0, False, True)
    h2 = _compare(P, hard, 12, 500, False, True)
    assert h1 > 8 and h2 >= 1
    # every instance through the dense stage, whose J / R slots are the scratch's (8 of them: both
    # batches hold more instances than that)
    dense = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps,
                             solver_mode=1)
    assert _compare(P, dense, 24, 0, True) == 24
    assert _compare(P, dense, 10, 300, True) == 10
    # the benchmark's weights: nothing is handed over, counter and list only
    easy = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps)
    _compare(P, easy, 96, 0, False)
    _compare(P, easy, 33, 96, False)
    _compare(P, hard, 20, 900, False, True)  # and the first handle again, after others used the scratch
    # working sets beyond 60 rows (weight_R = 1e-8, three times the walking speed): the wide edition
    # passes them on through the scratch's third list to the block edition
    harder = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=1e-8, weight_Q=1.0, number_of_steps=steps)
    _compare(P, harder, 64, 0, False, True, speed_scale=3.0, expect_block=True)
    _compare(P, harder, 8, 0, False, True, speed_scale=3.0, expect_block=True)
    # a handle set to longest-first dispatch: the allocating path orders its launch (second call on:
    # from the record of the one before), the no-alloc path always runs in index order -- same results
    easy.set_dispatch_hint(P.DISPATCH_LONGEST_FIRST)
    _compare(P, easy, 96, 200, False)
    _compare(P, easy, 96, 300, False)


def test_no_alloc_entry_point_takes_no_device_memory(P):
    """What the entry point exists for: on a handle that has never solved anything -- so that the
    allocating path would now size its lists and 2 x 24 J / R slots of 1.28 MB -- the call leaves
    the device's free memory where it was (the allocating call afterwards does take its slots)."""
    import torch
    N, steps, B = 200, 20, 24
    fresh = P.BatchedPlanner(N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=steps,
                             solver_mode=1)
    prepared = fresh.prepare_coupled(**batches.coupled_batch(B, first=7000))
    out = fresh._alloc_outputs(B, P.ALL_OUTPUTS)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    fresh.solve_coupled_prepared(prepared, out=out, no_alloc=True)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    got = {k: v.cpu().numpy().copy() for k, v in out[0].items()}
    assert (got["status"] == 0).all()
    fresh.solve_coupled_prepared(prepared, out=out)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] < free0 - 2 * 16 * N * N * 8  # at least the 16 initial slots of J and of R
    for k, v in out[0].items():
        assert np.array_equal(v.cpu().numpy(), got[k], equal_nan=True), k


def test_no_alloc_entry_point_rejects_sizes_beyond_its_scratch(P):
    big = P.BatchedPlanner(300, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=15)
    bt = batches.coupled_batch(4)
    prepared = big.prepare_coupled(**bt)
    with pytest.raises(P.MpcError, match="400 variables"):
        big.solve_coupled_prepared(prepared, no_alloc=True)
    pl = P.BatchedPlanner(200, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0, number_of_steps=20)
    B = 65537
    bt = batches.coupled_batch(B)
    prepared = pl.prepare_coupled(**bt)
    out = pl._alloc_outputs(B, ("status",))
    with pytest.raises(P.MpcError, match="65536"):
        pl.solve_coupled_prepared(prepared, out=out, no_alloc=True)
    box = P.BatchedPlanner(100, formulation=P.BOX_1D, weight_R=1.0)
    with pytest.raises(P.MpcError, match="coupled"):
        box.solve_coupled_prepared(pl.prepare_coupled(**batches.coupled_batch(2)), no_alloc=True)

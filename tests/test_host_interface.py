"""The host-side pieces of the MPCPlanner-shaped C++ header (no GPU needed): footstep and
polygon assembly, centroid, integrator, FootScheduler, Timer, GaitSequencer, planner.ini
reader -- against the CPU restatement of the reference and the reference's own check values."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_only")


def _run(*args):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "host_only"])
    r = subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = {}
    for line in r.stdout.splitlines():
        k, v = line.split("=", 1)
        out[k] = np.array([float(x) for x in v.split(",") if x]) if "," in v else v
    return out


@pytest.mark.parametrize("N,steps,swing,with_com", [(60, 6, 3, 0), (60, 6, 0, 1), (50, 7, 1, 1), (200, 20, 2, 0)])
def test_compute_steps_matches_the_restatement(oracle, N, steps, swing, with_com):
    d = _run(N, steps, swing, with_com)
    hip = np.array([[0.1, 1.0], [0.2, -1.0], [-0.4, 1.0], [-0.3, -1.0]])
    ref = oracle.compute_steps(N, steps, [0.13, -0.02], [0.11, 0.23, -0.41, -0.28], [1.02, -0.97, 1.01, -1.03],
                               swing0=swing, com0=(0.02, -0.01) if with_com else (0.0, 0.0), hip=hip)
    assert int(d["m"]) == ref["m"]
    # same formulas; the two builds differ in FMA contraction, hence a few ulps
    assert np.allclose(d["b"], ref["b"], rtol=1e-13, atol=1e-15)
    A = ref["A"]
    knots = np.array([np.nonzero(A[r, :N] != 0)[0][0] if (A[r, :N] != 0).any() else np.nonzero(A[r, N:] != 0)[0][0]
                      for r in range(ref["m"])])
    assert np.array_equal(d["row_knot"], knots)
    want = np.array([A[r, k] + 2.0 * A[r, k + N] for r, k in enumerate(knots)])
    assert np.allclose(d["row_p_plus_2q"], want, rtol=1e-13, atol=1e-15)
    assert np.array_equal(d["lf_x"], ref["fx"][0]) and np.array_equal(d["rh_y"], ref["fy"][3])
    assert np.array_equal(d["rf_swing"], ref["swing"][1])
    assert np.array_equal(d["hold_lh_x"], ref["hold_x"][2])
    cx, cy = oracle.centroid(ref["fx"], ref["fy"], ref["swing"])
    assert np.allclose(d["centroid_x"], cx, rtol=1e-14) and np.allclose(d["centroid_y"], cy, rtol=1e-14)


def test_integrator_known_answer_and_helpers(oracle, tmp_path):
    ini = tmp_path / "planner.ini"
    ini.write_text("[Crawl]\nlinearSpeedX = 0.12 # trailing comment\nlinearSpeedY=0.05\n[Other]\nlinearSpeedX = 9\n"
                   "[Replanning]\nuseComStepCorrection = 0\noptimizeVelocity = 1\n")
    d = _run(60, 6, 3, 0, ini)
    # the ten ZMP samples printed next to tests/unconstrainedQP/testPlanner.cpp:57
    kat = [0.0850698, 0.0711397, 0.0592095, 0.0502793, 0.0453491, 0.045419, 0.0514888, 0.0645586, 0.0856284, 0.115698]
    assert np.abs(d["zmp_kat"] - kat).max() < 2e-6
    assert np.allclose(d["gauss"], oracle.make_gaussian(8, 3, 2), rtol=1e-15)
    # sequence LF, RH, RF, LH positioned on RF (1): next LH (2), previous RH (3), opposite of RF
    # is LH (2); two steps further the schedule has wrapped to LF (0)
    assert d["sched"] == "12320"
    assert d["timer"] == "101" and float(d["missing"]) == pytest.approx(0.15)
    assert np.array_equal(d["ini"], [1, 0.12, 0.05, 0, 1])
    # crawl with a quarter-cycle phase shift: four lift-offs per cycle, every leg in the air for
    # a quarter of the time (plus the sample on which the timer's strict comparison releases it)
    assert int(d["gait_liftoffs"]) == 16 and float(d["swing_duration"]) == 1.0
    assert (np.abs(d["gait_air"] - 100) <= 2).all() and int(d["gait_max_air"]) <= 2

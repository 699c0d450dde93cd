"""ctypes binding of the CPU oracle (``libmpc_oracle.so``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py``.  The shipped package ``mpc_amd`` never
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LF, RF, LH, RH = 0, 1, 2, 3
COUPLED_JERK, COUPLED_VEL, COUPLED_REF = 0, 1, 2
OK, INFEASIBLE, ITER_CAP = 0, 1, 2


class Params(C.Structure):
    _fields_ = [("N", C.c_int), ("Ts", C.c_double), ("g", C.c_double),
                ("R", C.c_double), ("Q", C.c_double), ("Qs", C.c_double),
                ("height", C.c_double), ("dense", C.c_int), ("iter_cap", C.c_int)]


class Info(C.Structure):
    _fields_ = [("f", C.c_double), ("status", C.c_int), ("iters", C.c_int),
                ("n_active", C.c_int), ("n_add", C.c_int), ("n_drop", C.c_int),
                ("n_l2a", C.c_int), ("falg", C.c_double)]


def build(force=False):
    """Compile the oracle with g++ (no-op when the .so is newer than its sources)."""
    so = os.path.join(_HERE, "libmpc_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("mpc_oracle.cpp", "mpc_oracle.h")]
    if not force and os.path.exists(so) and all(
            os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    subprocess.check_call(["make", "-C", _HERE, "libmpc_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmpc_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_point2_is_right_of_line.restype = C.c_double
    return _LIB


def params(N, Ts=0.1, g=9.81, R=1e-6, Q=1.0, Qs=100.0, height=0.5, dense=0, iter_cap=0):
    return Params(N, Ts, g, R, Q, Qs, height, dense, iter_cap)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def build_matrix(p, which, size=None):
    size = p.N if size is None else size
    sx = np.zeros((size, 3))
    su = np.zeros((size, size))
    lib().orc_build_matrix(C.byref(p), which, size, _p(sx), _p(su))
    return sx, su


def trajectory(p, which, x0, jerk):
    x0 = _d(x0)
    jerk = _d(jerk)
    out = np.zeros(jerk.size)
    lib().orc_trajectory(C.byref(p), which, _p(x0), _p(jerk), jerk.size, _p(out))
    return out


def com_update(p, state, u):
    s = _d(state).copy()
    lib().orc_com_update(C.byref(p), _p(s), C.c_double(u))
    return s


def make_gaussian(length, mean, std):
    w = np.zeros(length)
    lib().orc_make_gaussian(length, mean, std, _p(w))
    return w


def solve_qp(p, x0, zref):
    x0 = _d(x0)
    zref = _d(zref)
    jerk = np.zeros(p.N)
    lib().orc_solve_qp(C.byref(p), _p(x0), _p(zref), _p(jerk))
    return jerk


def solve_box(p, x0, zmin, zmax):
    x0, zmin, zmax = _d(x0), _d(zmin), _d(zmax)
    jerk = np.zeros(p.N)
    info = Info()
    act = np.zeros(2 * p.N + 1, dtype=np.int32)
    lib().orc_solve_box(C.byref(p), _p(x0), _p(zmin), _p(zmax), _p(jerk), C.byref(info), _p(act))
    return jerk, info, act[:info.n_active].copy()


TRACE_COLS = ("iter", "pass", "iq", "ip", "s_ip", "s_runner_up", "t1", "t1_runner_up", "t2", "zz", "zn",
              "r_min_over_max", "blocking_row", "d_iq_over_eps_Rnorm")


def solve_box_trace(p, x0, zmin, zmax, max_rows=4000):
    """solve_box plus the record of its decisions: (jerk, info, active rows, trace [passes, 14])."""
    x0, zmin, zmax = _d(x0), _d(zmin), _d(zmax)
    jerk = np.zeros(p.N)
    info = Info()
    act = np.zeros(2 * p.N + 1, dtype=np.int32)
    rec = np.zeros((max_rows, len(TRACE_COLS)))
    fn = lib().orc_solve_box_trace
    n = fn(C.byref(p), _p(x0), _p(zmin), _p(zmax), _p(jerk), C.byref(info), _p(act), max_rows, _p(rec))
    return jerk, info, act[:info.n_active].copy(), rec[:n]


def point2_is_right_of_line(p0, p1, p2):
    p0, p1, p2 = _d(p0), _d(p1), _d(p2)
    return lib().orc_point2_is_right_of_line(_p(p0), _p(p1), _p(p2))


def ccw_sort(pts):
    pts = _d(pts).copy()
    lib().orc_ccw_sort(_p(pts), pts.shape[0])
    return pts


def line_coeff(p0, p1, normalize=True):
    p0, p1 = _d(p0), _d(p1)
    out = np.zeros(3)
    lib().orc_line_coeff(_p(p0), _p(p1), int(normalize), _p(out))
    return out


def compute_steps(N, steps, speed, feet_x, feet_y, swing0=LF, com0=(0.0, 0.0), hip=None):
    """Returns dict(A, b, fx, fy, swing, hold_x, hold_y, m)."""
    speed, feet_x, feet_y, com0 = _d(speed), _d(feet_x), _d(feet_y), _d(com0)
    hip_a = _d(hip) if hip is not None else None
    cap = 4 * (N // steps) * steps
    A = np.zeros((cap, 2 * N))
    b = np.zeros(cap)
    fx = np.zeros((4, N))
    fy = np.zeros((4, N))
    sw = np.zeros((4, N))
    hx = np.zeros((4, steps))
    hy = np.zeros((4, steps))
    m = lib().orc_compute_steps(N, steps, _p(speed), _p(feet_x), _p(feet_y), int(swing0),
                                _p(com0), _p(hip_a), _p(A), _p(b), _p(fx), _p(fy), _p(sw),
                                _p(hx), _p(hy))
    if m < 0:
        raise ValueError("computeSteps failed: %d" % m)
    # the C side wrote an m x 2N row-major matrix at the start of the buffer
    A = A.reshape(-1)[:m * 2 * N].reshape(m, 2 * N).copy()
    return dict(A=A, b=b[:m].copy(), fx=fx, fy=fy, swing=sw, hold_x=hx, hold_y=hy, m=m)


def centroid(fx, fy, swing):
    fx, fy, swing = _d(fx), _d(fy), _d(swing)
    N = fx.shape[1]
    cx = np.zeros(N)
    cy = np.zeros(N)
    lib().orc_centroid(N, _p(fx), _p(fy), _p(swing), _p(cx), _p(cy))
    return cx, cy


def constraint_violation(swing, viol_all):
    swing, viol_all = _d(swing), _d(viol_all)
    N = swing.shape[1]
    out = np.zeros(N)
    lib().orc_constraint_violation(N, _p(swing), _p(viol_all), _p(out))
    return out


def solve_coupled(p, variant, x0, y0, A, b, speed=(0.0, 0.0), rw=1000, zrefx=None, zrefy=None):
    """Returns dict(jerk_x, jerk_y, viol_all, info, active)."""
    x0, y0, A, b, speed = _d(x0), _d(y0), _d(A), _d(b), _d(speed)
    m = b.size
    zx = _d(zrefx) if zrefx is not None else None
    zy = _d(zrefy) if zrefy is not None else None
    jx = np.zeros(p.N)
    jy = np.zeros(p.N)
    viol = np.zeros(m)
    info = Info()
    act = np.zeros(m + 1, dtype=np.int32)
    lib().orc_solve_coupled(C.byref(p), variant, _p(x0), _p(y0), _p(zx), _p(zy), _p(A), _p(b), m,
                            _p(speed), int(rw), _p(jx), _p(jy), _p(viol), C.byref(info), _p(act))
    return dict(jerk_x=jx, jerk_y=jy, viol_all=viol, info=info,
                active=act[:info.n_active].copy())


def solve_quadprog(G, g0, CI, ci0, iter_cap=0):
    G, g0, CI, ci0 = _d(G), _d(g0), _d(CI), _d(ci0)
    n, m = g0.size, ci0.size
    x = np.zeros(n)
    info = Info()
    act = np.zeros(m + 1, dtype=np.int32)
    lib().orc_solve_quadprog(n, m, _p(G), _p(g0), _p(CI), _p(ci0), _p(x), C.byref(info), _p(act),
                             iter_cap)
    return x, info, act[:info.n_active].copy()


def coupled_batch(p, variant, steps, rw, x0, y0, speed, feet_x, feet_y, swing0, com0=None,
                  active_stride=64, nthreads=1, trajectories=True, hip=None):
    """The reference planner looped over a batch (one planner per thread)."""
    x0, y0, speed, feet_x, feet_y = _d(x0), _d(y0), _d(speed), _d(feet_x), _d(feet_y)
    swing0 = np.ascontiguousarray(swing0, dtype=np.int32)
    com0_a = _d(com0) if com0 is not None else None
    hip_a = _d(hip).reshape(8) if hip is not None else None
    B, N = x0.shape[0], p.N
    out = dict(jerk_x=np.zeros((B, N)), jerk_y=np.zeros((B, N)),
               viol_knot=np.zeros((B, N)), cost=np.zeros(B),
               status=np.zeros(B, dtype=np.int32), n_active=np.zeros(B, dtype=np.int32),
               active=np.zeros((B, active_stride), dtype=np.int32),
               iters=np.zeros(B, dtype=np.int32), falg=np.zeros(B))
    for k in ("com_x", "com_y", "zmp_x", "zmp_y"):
        out[k] = np.zeros((B, N)) if trajectories else None
    rc = lib().orc_coupled_batch(
        C.byref(p), variant, steps, int(rw), B, _p(x0), _p(y0), _p(speed), _p(feet_x), _p(feet_y),
        _p(swing0), _p(com0_a), _p(hip_a), _p(out["jerk_x"]), _p(out["jerk_y"]), _p(out["com_x"]),
        _p(out["com_y"]), _p(out["zmp_x"]), _p(out["zmp_y"]), _p(out["viol_knot"]),
        _p(out["cost"]), _p(out["status"]), _p(out["n_active"]), _p(out["active"]),
        active_stride, _p(out["iters"]), _p(out["falg"]), nthreads)
    if rc:
        raise ValueError("orc_coupled_batch failed: %d" % rc)
    return out


def box_batch(p, x0, y0, xmin, xmax, ymin, ymax, active_stride=64, nthreads=1):
    x0, y0, xmin, xmax, ymin, ymax = map(_d, (x0, y0, xmin, xmax, ymin, ymax))
    B, N = x0.shape[0], p.N
    out = dict(jerk_x=np.zeros((B, N)), jerk_y=np.zeros((B, N)), cost=np.zeros((B, 2)),
               status=np.zeros((B, 2), dtype=np.int32), n_active=np.zeros((B, 2), dtype=np.int32),
               active=np.zeros((B, 2, active_stride), dtype=np.int32),
               iters=np.zeros((B, 2), dtype=np.int32), falg=np.zeros((B, 2)))
    lib().orc_box_batch(C.byref(p), B, _p(x0), _p(y0), _p(xmin), _p(xmax), _p(ymin), _p(ymax),
                        _p(out["jerk_x"]), _p(out["jerk_y"]), _p(out["cost"]), _p(out["status"]),
                        _p(out["n_active"]), _p(out["active"]), active_stride, _p(out["iters"]),
                        _p(out["falg"]), nthreads)
    return out


def slack_batch(p, x0, zmin, zmax, active_stride=512, nthreads=1):
    """solveQPconstraintSlack (MPCPlanner.cpp:303-424) for a batch of one-axis instances.
    sol[:, :N] is the jerk, sol[:, N:] the 2N slacks."""
    x0, zmin, zmax = map(_d, (x0, zmin, zmax))
    B, N = x0.shape[0], p.N
    out = dict(sol=np.zeros((B, 3 * N)), cost=np.zeros(B), status=np.zeros(B, dtype=np.int32),
               n_active=np.zeros(B, dtype=np.int32), active=np.zeros((B, active_stride), dtype=np.int32),
               iters=np.zeros(B, dtype=np.int32), falg=np.zeros(B))
    lib().orc_slack_batch(C.byref(p), B, _p(x0), _p(zmin), _p(zmax), _p(out["sol"]), _p(out["cost"]),
                          _p(out["status"]), _p(out["n_active"]), _p(out["active"]), active_stride,
                          _p(out["iters"]), _p(out["falg"]), nthreads)
    out["jerk"] = out["sol"][:, :N]
    out["slacks"] = out["sol"][:, N:]
    return out


def replanning_batch(p, steps, n_stages, x0, y0, speed, feet_x, feet_y, swing0, disturbance,
                     disturbance_start=4, use_com_correction=True, hip=None, nthreads=1):
    """The replanning driver's loop for a batch of rollouts.  Returns the evolved state
    (x0, y0, feet_x, feet_y, swing0), per-stage cost/status/n_active and the last plan."""
    x0, y0, speed, feet_x, feet_y = (_d(a).copy() for a in (x0, y0, speed, feet_x, feet_y))
    swing0 = np.ascontiguousarray(swing0, dtype=np.int32).copy()
    disturbance = _d(disturbance)
    hip_a = _d(hip).reshape(8) if hip is not None else None
    B, N = x0.shape[0], p.N
    out = dict(stage_cost=np.zeros((n_stages, B)), stage_status=np.zeros((n_stages, B), dtype=np.int32),
               stage_nactive=np.zeros((n_stages, B), dtype=np.int32), jerk_x=np.zeros((B, N)),
               jerk_y=np.zeros((B, N)))
    rc = lib().orc_replanning_batch(C.byref(p), steps, n_stages, int(use_com_correction), _p(hip_a), B,
                                    _p(x0), _p(y0), _p(speed), _p(feet_x), _p(feet_y), _p(swing0),
                                    _p(disturbance), int(disturbance_start), _p(out["stage_cost"]),
                                    _p(out["stage_status"]), _p(out["stage_nactive"]),
                                    _p(out["jerk_x"]), _p(out["jerk_y"]), nthreads)
    if rc:
        raise ValueError("orc_replanning_batch failed: %d" % rc)
    out.update(x0=x0, y0=y0, feet_x=feet_x, feet_y=feet_y, swing0=swing0)
    return out

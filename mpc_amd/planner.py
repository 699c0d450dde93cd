"""Python host side of the batched planner: a thin ctypes layer over the C ABI
(``include/mpc_amd.h``) that keeps MPCPlanner's vocabulary (horizon, weights,
footsteps, jerk/CoM/ZMP trajectories).  PyTorch only provides device memory and
streams; every computation happens in ``libmpc_amd.so``.

There is no CPU fallback: if the shared library or a GPU is missing, construction
fails.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPC_AMD_LIB selects another build of the same library (e.g. the cycle-stamp build)
LIB_PATH = os.environ.get("MPC_AMD_LIB") or os.path.join(_HERE, "libmpc_amd.so")

BOX_1D, COUPLED_JERK, COUPLED_VEL, COUPLED_REF, UNCONSTRAINED, BOX_SLACK = 0, 1, 2, 3, 4, 5
OK, INFEASIBLE, ITER_CAP = 0, 1, 2
MODE_AUTO, MODE_DENSE = 0, 1
TABLES_DEVICE, TABLES_HOST = 0, 1
DISPATCH_IN_ORDER, DISPATCH_LONGEST_FIRST = 0, 1

# every symbol include/mpc_amd.h declares
EXPORTS = ("mpcb_create", "mpcb_destroy", "mpcb_last_error", "mpcb_num_constraints", "mpcb_num_phases",
           "mpcb_solve_coupled_device", "mpcb_solve_coupled", "mpcb_solve_box_device",
           "mpcb_solve_box", "mpcb_get_matrix", "mpcb_replan_sweep_device",
           "mpcb_solve_unconstrained_device", "mpcb_solve_unconstrained", "mpcb_last_handover_count", "mpcb_last_wide_count", "mpcb_last_block_count",
           "mpcb_solve_coupled_device_noalloc", "mpcb_solve_coupled_multi", "mpcb_debug_set_stamp_buffer",
           "mpcb_set_dispatch_hint", "mpcb_solve_coupled_multi_device", "mpcb_solve_coupled_groups",
           "mpcb_create_groups", "mpcb_debug_host_table", "mpcb_solve_unconstrained_heights")


class Params(C.Structure):
    _fields_ = [("horizon", C.c_int32), ("formulation", C.c_int32), ("Ts", C.c_double),
                ("gravity", C.c_double), ("height", C.c_double), ("weight_R", C.c_double),
                ("weight_Q", C.c_double), ("weight_Qs", C.c_double),
                ("replanning_window", C.c_int32), ("number_of_steps", C.c_int32),
                ("hip_offsets", C.c_double * 8), ("iter_cap", C.c_int32), ("device", C.c_int32),
                ("solver_mode", C.c_int32), ("table_build", C.c_int32)]


class CoupledInputs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in
                ("x0", "y0", "speed", "feet_x", "feet_y", "swing0", "com0", "rows_pqr", "zref_phase", "height", "zref_knot",
                 "phase_pqr")]


class ReplanState(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in
                ("x0", "y0", "speed", "feet_x", "feet_y", "swing0", "com0", "disturbance")]


_OUT_F64 = ("jerk_x", "jerk_y", "com_x", "com_y", "com_xd", "com_yd", "zmp_x", "zmp_y",
            "min_margin", "cost")
_OUT_I32 = ("status", "iters", "n_active", "active_idx")


class Outputs(C.Structure):
    _fields_ = ([(k, C.c_void_p) for k in _OUT_F64] + [(k, C.c_void_p) for k in _OUT_I32] +
                [("multipliers", C.c_void_p), ("active_stride", C.c_int32),
                 ("alg_flops", C.c_void_p), ("slacks", C.c_void_p)])


_lib = None


def load_library():
    """Loads libmpc_amd.so; raises if it is missing (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found: build it with `make -C mpc_amd/csrc` "
                               "(or __graft_entry__.build())" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.mpcb_last_error.restype = C.c_char_p
        lib.mpcb_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
        lib.mpcb_create_groups.argtypes = [C.POINTER(Params), C.c_int32, C.POINTER(C.c_void_p)]
        lib.mpcb_debug_host_table.argtypes = [C.POINTER(Params), C.c_int32, C.c_void_p]
        lib.mpcb_destroy.argtypes = [C.c_void_p]
        lib.mpcb_destroy.restype = None
        lib.mpcb_num_constraints.argtypes = [C.c_void_p]
        lib.mpcb_solve_coupled_device.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CoupledInputs),
                                                  C.POINTER(Outputs), C.c_void_p]
        lib.mpcb_solve_coupled.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CoupledInputs),
                                           C.POINTER(Outputs)]
        lib.mpcb_solve_coupled_device_noalloc.argtypes = lib.mpcb_solve_coupled_device.argtypes
        lib.mpcb_solve_coupled_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32,
                                                 C.POINTER(CoupledInputs), C.POINTER(Outputs)]
        lib.mpcb_solve_coupled_multi_device.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_int32),
                                                        C.POINTER(CoupledInputs), C.POINTER(Outputs),
                                                        C.POINTER(C.c_void_p)]
        lib.mpcb_solve_coupled_groups.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_int32),
                                                  C.POINTER(CoupledInputs), C.POINTER(Outputs), C.c_void_p]
        lib.mpcb_solve_box_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.POINTER(Outputs), C.c_void_p]
        lib.mpcb_get_matrix.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.mpcb_last_handover_count.argtypes = [C.c_void_p]
        lib.mpcb_last_wide_count.argtypes = [C.c_void_p]
        lib.mpcb_last_block_count.argtypes = [C.c_void_p]
        lib.mpcb_set_dispatch_hint.argtypes = [C.c_void_p, C.c_int32]
        lib.mpcb_solve_unconstrained_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                                        C.POINTER(Outputs), C.c_void_p]
        lib.mpcb_solve_unconstrained_heights.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                         C.POINTER(Outputs)]
        lib.mpcb_replan_sweep_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                 C.POINTER(ReplanState), C.POINTER(Outputs), C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


def host_table(which, horizon, Ts=0.1, gravity=9.81, formulation=COUPLED_VEL, height=0.5, weight_R=1e-6,
               weight_Q=1.0, weight_Qs=100.0, replanning_window=1000, number_of_steps=20):
    """A table of the long-double host construction (mpcb_debug_host_table; needs no GPU), flat."""
    lib = load_library()
    p = Params()
    p.horizon, p.formulation = int(horizon), int(formulation)
    p.Ts, p.gravity, p.height = float(Ts), float(gravity), float(height)
    p.weight_R, p.weight_Q, p.weight_Qs = float(weight_R), float(weight_Q), float(weight_Qs)
    p.replanning_window, p.number_of_steps = int(replanning_window), int(number_of_steps)
    n = _check(lib.mpcb_debug_host_table(C.byref(p), which, None))
    out = np.zeros(n)
    _check(lib.mpcb_debug_host_table(C.byref(p), which, out.ctypes.data_as(C.c_void_p)))
    return out


class MpcError(RuntimeError):
    pass


def _check(rc):
    if rc < 0:
        raise MpcError("libmpc_amd error %d: %s" % (rc, load_library().mpcb_last_error().decode()))
    return rc


ALL_OUTPUTS = _OUT_F64 + _OUT_I32 + ("multipliers", "alg_flops")
SLACK_OUTPUTS = ALL_OUTPUTS + ("slacks",)  # BOX_SLACK handles
DEFAULT_OUTPUTS = ALL_OUTPUTS


class BatchedPlanner:
    """B independent MPCPlanner calls at once on one GPU.

    Constructor arguments are those of ``MPCPlanner::MPCPlanner(horizon, Ts, gravity)``
    (MPCPlanner.cpp:22), ``setWeights`` (:94-98) and the per-call scalars the reference
    passes to its solve methods (height, replanning window, number of steps)."""

    def __init__(self, horizon, Ts=0.1, gravity=9.81, formulation=COUPLED_VEL, height=0.5,
                 weight_R=1e-6, weight_Q=1.0, weight_Qs=100.0, replanning_window=1000,
                 number_of_steps=20, hip_offsets=None, iter_cap=0, device=0, active_stride=64,
                 solver_mode=0, table_build=TABLES_DEVICE, _handle=None):
        import torch
        self._torch = torch
        lib = load_library()
        p = Params()
        p.horizon, p.formulation = int(horizon), int(formulation)
        p.Ts, p.gravity, p.height = float(Ts), float(gravity), float(height)
        p.weight_R, p.weight_Q, p.weight_Qs = float(weight_R), float(weight_Q), float(weight_Qs)
        p.replanning_window, p.number_of_steps = int(replanning_window), int(number_of_steps)
        hip = np.zeros(8) if hip_offsets is None else np.asarray(hip_offsets, dtype=np.float64).reshape(8)
        for i in range(8):
            p.hip_offsets[i] = float(hip[i])
        p.iter_cap, p.device = int(iter_cap), int(device)
        p.solver_mode = int(solver_mode)
        p.table_build = int(table_build)
        self.params = p
        self.horizon = int(horizon)
        self.formulation = int(formulation)
        self.device = torch.device("cuda", int(device))
        self.active_stride = int(active_stride)
        if _handle is None:
            h = C.c_void_p()
            _check(lib.mpcb_create(C.byref(p), C.byref(h)))
        elif _handle == "deferred":
            h = None    # create_groups fills it in
        else:
            h = _handle
        self._h = h
        self.num_constraints = lib.mpcb_num_constraints(h) if h else None

    @classmethod
    def create_groups(cls, specs, active_stride=64):
        """One planner per entry of `specs` (dicts of constructor arguments; horizons, weights,
        heights and formulations may all differ), their tables built together by one
        mpcb_create_groups call -- one chain of launches for the whole list."""
        lib = load_library()
        pls = [cls(active_stride=active_stride, _handle="deferred", **s) for s in specs]
        n = len(pls)
        ps = (Params * n)()
        for g, pl in enumerate(pls):
            C.memmove(C.byref(ps[g]), C.byref(pl.params), C.sizeof(Params))
        hs = (C.c_void_p * n)()
        _check(lib.mpcb_create_groups(ps, n, hs))
        for g, pl in enumerate(pls):
            pl._h = C.c_void_p(hs[g])
            pl.num_constraints = lib.mpcb_num_constraints(pl._h)
        return pls

    def close(self):
        if getattr(self, "_h", None):
            load_library().mpcb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------
    def _dev(self, a, dtype):
        torch = self._torch
        if a is None:
            return None
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=dtype)
        else:
            t = torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=self.device)
        return t.contiguous()

    def _alloc_outputs(self, B, outputs):
        torch = self._torch
        N = self.horizon
        res, o = {}, Outputs()
        o.active_stride = self.active_stride
        for name in SLACK_OUTPUTS:
            if name not in outputs:
                continue
            if name == "slacks":
                t = torch.empty((B, 2 * N), dtype=torch.float64, device=self.device)
            elif name in ("cost", "alg_flops"):
                t = torch.empty(B, dtype=torch.float64, device=self.device)
            elif name in ("status", "iters", "n_active"):
                t = torch.empty(B, dtype=torch.int32, device=self.device)
            elif name == "active_idx":
                t = torch.empty((B, self.active_stride), dtype=torch.int32, device=self.device)
            elif name == "multipliers":
                t = torch.empty((B, self.active_stride), dtype=torch.float64, device=self.device)
            else:
                t = torch.empty((B, N), dtype=torch.float64, device=self.device)
            res[name] = t
            setattr(o, name, t.data_ptr())
        return res, o

    def output_views(self, out, first, count):
        """An output set (as `_alloc_outputs` returns it) over instances [first, first + count) of
        `out`: a sub-batch solved into it lands in the arrays of the whole batch."""
        res_all, _ = out
        res, o = {}, Outputs()
        o.active_stride = self.active_stride
        for name, t in res_all.items():
            v = t[first:first + count]
            res[name] = v
            setattr(o, name, v.data_ptr())
        return res, o

    def _stream(self, stream):
        if stream is None:
            stream = self._torch.cuda.current_stream(self.device)
        return C.c_void_p(stream.cuda_stream)

    def set_dispatch_hint(self, mode):
        """0 = DISPATCH_IN_ORDER (default); k = 1 (DISPATCH_LONGEST_FIRST) .. 8: start the instances
        of a solve in decreasing order of the running times the k-th solve before it recorded
        (scheduling only: results do not change)."""
        _check(load_library().mpcb_set_dispatch_hint(self._h, int(mode)))

    def last_wide_count(self):
        """Instances of the most recent coupled solve that the wide edition of the first-stage kernel took
        over (working sets of 33 to 60 rows)."""
        return _check(load_library().mpcb_last_wide_count(self._h))

    def last_block_count(self):
        """Instances of the most recent coupled solve that outgrew the wide edition as well and went
        to the block edition (working sets of 61 to 160 rows)."""
        return _check(load_library().mpcb_last_block_count(self._h))

    def last_handover_count(self):
        """Instances of the most recent solve that went through the dense second-stage kernel."""
        return _check(load_library().mpcb_last_handover_count(self._h))

    def matrix(self, which):
        """Instance-independent tables copied from the device (ids of mpcb_get_matrix: 0 S, 1 W^T,
        2 CoM, 3 CoM velocity, 4 Kz, 5 Kx, 6-11 prediction matrices, 12 L^-T, 13 F, 14-17 the
        per-phase tables of COUPLED_REF, 18-24 the height-free parts)."""
        lib = load_library()
        n = _check(lib.mpcb_get_matrix(self._h, which, None))
        out = np.zeros(n)
        _check(lib.mpcb_get_matrix(self._h, which, out.ctypes.data_as(C.c_void_p)))
        N = self.horizon
        if n == N * N:
            return out.reshape(N, N)
        if which in (14, 15):
            return out.reshape(-1, N)
        if which == 16:
            return out.reshape(4, -1)
        if which == 17:
            k = int(round(n ** 0.5))
            return out.reshape(k, k)
        if which == 13:
            return out.reshape(4, 4)
        if n == 4 * N:
            return out.reshape(N, 4)
        if n == 3 * N:
            return out.reshape(N, 3)
        return out

    # -- solves ----------------------------------------------------------------
    def prepare_coupled(self, x0, y0, speed, feet_x=None, feet_y=None, swing0=None, com0=None,
                        rows_pqr=None, zref_phase=None, height=None, zref_knot=None, phase_pqr=None):
        """Uploads one batch of coupled inputs; returns an opaque (B, struct, keepalive).
        rows_pqr [B, m, 3] replaces the footstep inputs by caller-built polygon rows;
        zref_phase [B, 2*steps+1, 2] overrides the centroid ZMP reference (COUPLED_REF);
        height [B] gives every instance its own CoM height (COUPLED_JERK / COUPLED_VEL);
        zref_knot [B, N, 2] is a ZMP reference that may change at every knot (COUPLED_REF);
        phase_pqr [B, 2*steps+1, 4, 3] replaces the footstep inputs by the caller's polygon of every
        footstep phase (edge e of phase ph = (p, q, r); a swing phase uses its first three)."""
        f64, i32 = self._torch.float64, self._torch.int32
        ts = dict(x0=self._dev(x0, f64), y0=self._dev(y0, f64), speed=self._dev(speed, f64),
                  feet_x=self._dev(feet_x, f64), feet_y=self._dev(feet_y, f64),
                  swing0=self._dev(swing0, i32), com0=self._dev(com0, f64),
                  rows_pqr=self._dev(rows_pqr, f64), zref_phase=self._dev(zref_phase, f64),
                  height=self._dev(height, f64), zref_knot=self._dev(zref_knot, f64),
                  phase_pqr=self._dev(phase_pqr, f64))
        B = ts["x0"].shape[0]
        nph = 2 * self.params.number_of_steps + 1
        shapes = dict(x0=(B, 3), y0=(B, 3), speed=(B, 2), feet_x=(B, 4), feet_y=(B, 4),
                      swing0=(B,), com0=(B, 2), rows_pqr=(B, self.num_constraints, 3),
                      zref_phase=(B, nph, 2), height=(B,), zref_knot=(B, self.horizon, 2),
                      phase_pqr=(B, nph, 4, 3))
        ci = CoupledInputs()
        for k, t in ts.items():
            if t is None:
                continue
            if tuple(t.shape) != shapes[k]:
                raise ValueError("%s must have shape %s, got %s" % (k, shapes[k], tuple(t.shape)))
            setattr(ci, k, t.data_ptr())
        return B, ci, ts

    def solve_coupled_prepared(self, prepared, outputs=DEFAULT_OUTPUTS, out=None, stream=None,
                               no_alloc=False):
        """no_alloc=True goes through mpcb_solve_coupled_device_noalloc (static scratch instead of
        the handle's allocated workspace; pass `out` so that nothing is allocated here either)."""
        B, ci, _ = prepared
        if out is None:
            out = self._alloc_outputs(B, outputs)
        res, o = out
        lib = load_library()
        fn = lib.mpcb_solve_coupled_device_noalloc if no_alloc else lib.mpcb_solve_coupled_device
        _check(fn(self._h, B, C.byref(ci), C.byref(o), self._stream(stream)))
        return res

    def solve_coupled(self, x0, y0, speed, feet_x=None, feet_y=None, swing0=None, com0=None,
                      rows_pqr=None, zref_phase=None, height=None, zref_knot=None, outputs=DEFAULT_OUTPUTS,
                      stream=None, phase_pqr=None):
        """computeSteps + solveQPConstraintCoupled + trajectory read-out for B instances.
        Returns a dict of device tensors (asynchronous on `stream`)."""
        prepared = self.prepare_coupled(x0, y0, speed, feet_x, feet_y, swing0, com0, rows_pqr,
                                        zref_phase, height, zref_knot, phase_pqr)
        return self.solve_coupled_prepared(prepared, outputs=outputs, stream=stream)

    def host_output_arrays(self, B, outputs=DEFAULT_OUTPUTS, pinned=False):
        """Host arrays for solve_coupled_host(out=...); pinned=True page-locks them (views of
        pinned torch tensors), which is what lets the copies run at PCIe speed."""
        N, res = self.horizon, {}
        for name in outputs:
            if name in ("cost", "alg_flops"):
                shape, dt = (B,), np.float64
            elif name in ("status", "iters", "n_active"):
                shape, dt = (B,), np.int32
            elif name == "active_idx":
                shape, dt = (B, self.active_stride), np.int32
            elif name == "multipliers":
                shape, dt = (B, self.active_stride), np.float64
            else:
                shape, dt = (B, N), np.float64
            if pinned:
                tdt = self._torch.float64 if dt is np.float64 else self._torch.int32
                res[name] = self._torch.zeros(shape, dtype=tdt).pin_memory().numpy()
            else:
                res[name] = np.zeros(shape, dt)
        return res

    def solve_coupled_host(self, x0, y0, speed, feet_x=None, feet_y=None, swing0=None, com0=None,
                           outputs=DEFAULT_OUTPUTS, out=None, height=None, phase_pqr=None, zref_phase=None):
        """The same call on host memory (numpy in, numpy out) through mpcb_solve_coupled: the
        library stages inputs and outputs over PCIe itself and returns when they are back.
        out: arrays from host_output_arrays to fill (reused between calls).  height [B]: CoM height
        per instance (COUPLED_REF: the library groups the instances by height itself); phase_pqr /
        zref_phase as in prepare_coupled (the footstep inputs may then be left out)."""
        f64 = np.float64
        opt = lambda a, dt=f64: None if a is None else np.ascontiguousarray(a, dt)
        arrs = dict(x0=np.ascontiguousarray(x0, f64), y0=np.ascontiguousarray(y0, f64),
                    speed=np.ascontiguousarray(speed, f64), feet_x=opt(feet_x), feet_y=opt(feet_y),
                    swing0=opt(swing0, np.int32), com0=opt(com0), height=opt(height),
                    phase_pqr=opt(phase_pqr), zref_phase=opt(zref_phase))
        B = arrs["x0"].shape[0]
        ci = CoupledInputs()
        for k, a in arrs.items():
            if a is not None:
                setattr(ci, k, a.ctypes.data)
        res = out if out is not None else self.host_output_arrays(B, outputs)
        o = Outputs()
        o.active_stride = self.active_stride
        for name, a in res.items():
            if a.shape[0] != B or not a.flags.c_contiguous:
                raise ValueError("output array %s does not fit the batch" % name)
            setattr(o, name, a.ctypes.data)
        _check(load_library().mpcb_solve_coupled(self._h, B, C.byref(ci), C.byref(o)))
        return res

    @staticmethod
    def solve_coupled_multi(planners, x0, y0, speed, feet_x, feet_y, swing0, com0=None,
                            outputs=DEFAULT_OUTPUTS, out=None):
        """One batch over several planners (normally one per GPU) through
        mpcb_solve_coupled_multi: numpy in, numpy out, contiguous shards, one host thread and
        one stream per planner inside the library."""
        first = planners[0]
        f64 = np.float64
        arrs = dict(x0=np.ascontiguousarray(x0, f64), y0=np.ascontiguousarray(y0, f64),
                    speed=np.ascontiguousarray(speed, f64), feet_x=np.ascontiguousarray(feet_x, f64),
                    feet_y=np.ascontiguousarray(feet_y, f64),
                    swing0=np.ascontiguousarray(swing0, np.int32),
                    com0=None if com0 is None else np.ascontiguousarray(com0, f64))
        B = arrs["x0"].shape[0]
        ci = CoupledInputs()
        for k, a in arrs.items():
            if a is not None:
                setattr(ci, k, a.ctypes.data)
        res = out if out is not None else first.host_output_arrays(B, outputs)
        o = Outputs()
        o.active_stride = first.active_stride
        for name, a in res.items():
            if a.shape[0] != B or not a.flags.c_contiguous:
                raise ValueError("output array %s does not fit the batch" % name)
            setattr(o, name, a.ctypes.data)
        hs = (C.c_void_p * len(planners))(*[pl._h for pl in planners])
        _check(load_library().mpcb_solve_coupled_multi(hs, len(planners), B, C.byref(ci), C.byref(o)))
        return res

    @staticmethod
    def _group_arrays(planners, prepared, outs):
        n = len(planners)
        hs = (C.c_void_p * n)(*[pl._h for pl in planners])
        Bs = (C.c_int32 * n)(*[p[0] for p in prepared])
        ins = (CoupledInputs * n)()
        os_ = (Outputs * n)()
        for g in range(n):
            C.memmove(C.byref(ins[g]), C.byref(prepared[g][1]), C.sizeof(CoupledInputs))
            C.memmove(C.byref(os_[g]), C.byref(outs[g][1]), C.sizeof(Outputs))
        return hs, Bs, ins, os_

    @staticmethod
    def solve_coupled_multi_device(planners, prepared, outs, streams=None):
        """Shard g = prepared[g] (from planners[g].prepare_coupled) solved with planners[g] into
        outs[g] (from planners[g]._alloc_outputs), all enqueued by one call of
        mpcb_solve_coupled_multi_device; streams[g] is a torch stream of planners[g]'s device or
        None for its default stream.  Asynchronous."""
        n = len(planners)
        hs, Bs, ins, os_ = BatchedPlanner._group_arrays(planners, prepared, outs)
        st = None
        if streams is not None:
            st = (C.c_void_p * n)(*[None if s is None else s.cuda_stream for s in streams])
        _check(load_library().mpcb_solve_coupled_multi_device(hs, n, Bs, ins, os_, st))
        return [o[0] for o in outs]

    @staticmethod
    def solve_coupled_groups(planners, prepared, outs, stream=None):
        """Parameter groups of one device in one call (mpcb_solve_coupled_groups): group g is
        prepared[g] solved with planners[g] into outs[g].  Asynchronous on `stream`."""
        n = len(planners)
        hs, Bs, ins, os_ = BatchedPlanner._group_arrays(planners, prepared, outs)
        _check(load_library().mpcb_solve_coupled_groups(hs, n, Bs, ins, os_, planners[0]._stream(stream)))
        return [o[0] for o in outs]

    def prepare_box(self, x0, zmin, zmax, height=None):
        f64 = self._torch.float64
        ts = (self._dev(x0, f64), self._dev(zmin, f64), self._dev(zmax, f64), self._dev(height, f64))
        B = ts[0].shape[0]
        if ts[3] is not None and tuple(ts[3].shape) != (B,):
            raise ValueError("height must be [B]")
        if tuple(ts[0].shape) != (B, 3) or tuple(ts[1].shape) != (B, self.horizon) or \
                tuple(ts[2].shape) != (B, self.horizon):
            raise ValueError("x0 must be [B,3]; zmin, zmax must be [B,horizon]")
        return B, ts

    def solve_box_prepared(self, prepared, outputs=DEFAULT_OUTPUTS, out=None, stream=None):
        B, ts = prepared
        if out is None:
            out = self._alloc_outputs(B, outputs)
        res, o = out
        _check(load_library().mpcb_solve_box_device(self._h, B, ts[0].data_ptr(), ts[1].data_ptr(),
                                                    ts[2].data_ptr(),
                                                    None if ts[3] is None else ts[3].data_ptr(), C.byref(o),
                                                    self._stream(stream)))
        return res

    def solve_box(self, x0, zmin, zmax, height=None, outputs=DEFAULT_OUTPUTS, stream=None):
        """solveQPconstraint for B one-axis instances; results are in the *_x outputs."""
        return self.solve_box_prepared(self.prepare_box(x0, zmin, zmax, height), outputs=outputs,
                                       stream=stream)

    def solve_unconstrained(self, x0, zmp_ref, outputs=("jerk_x", "zmp_x", "com_x", "com_xd", "status"),
                            stream=None):
        """solveQP for B instances: tracks zmp_ref [B, horizon] from x0 [B, 3]."""
        f64 = self._torch.float64
        tx, tz = self._dev(x0, f64), self._dev(zmp_ref, f64)
        B = tx.shape[0]
        if tuple(tx.shape) != (B, 3) or tuple(tz.shape) != (B, self.horizon):
            raise ValueError("x0 must be [B,3] and zmp_ref [B,horizon]")
        res, o = self._alloc_outputs(B, outputs)
        _check(load_library().mpcb_solve_unconstrained_device(self._h, B, tx.data_ptr(), tz.data_ptr(),
                                                              C.byref(o), self._stream(stream)))
        return res

    def solve_unconstrained_host(self, x0, zmp_ref, height=None, outputs=("jerk_x", "zmp_x", "com_x", "com_xd", "status")):
        """solveQP on host memory (numpy in, numpy out) with an optional CoM height per instance
        (mpcb_solve_unconstrained_heights: instances grouped on distinct heights by the library)."""
        x0 = np.ascontiguousarray(x0, np.float64)
        zr = np.ascontiguousarray(zmp_ref, np.float64)
        B, N = x0.shape[0], self.horizon
        hh = None if height is None else np.ascontiguousarray(height, np.float64)
        res, o = {}, Outputs()
        for name in outputs:
            # (per-instance integers start at -1 so that what the library writes can be told apart)
            res[name] = np.full(B, -1, np.int32) if name in ("status", "iters", "n_active") else np.zeros((B, N))
            setattr(o, name, res[name].ctypes.data)
        _check(load_library().mpcb_solve_unconstrained_heights(
            self._h, B, x0.ctypes.data, zr.ctypes.data, None if hh is None else hh.ctypes.data, C.byref(o)))
        return res

    # -- receding-horizon sweep ---------------------------------------------------
    def prepare_replanning(self, x0, y0, speed, feet_x, feet_y, swing0, disturbance=None, **_):
        """Uploads the initial state of B rollouts (copied: the sweep updates it in place)."""
        f64, i32 = self._torch.float64, self._torch.int32
        ts = dict(x0=self._dev(x0, f64).clone(), y0=self._dev(y0, f64).clone(),
                  speed=self._dev(speed, f64), feet_x=self._dev(feet_x, f64).clone(),
                  feet_y=self._dev(feet_y, f64).clone(), swing0=self._dev(swing0, i32).clone(),
                  disturbance=self._dev(disturbance, f64))
        B = ts["x0"].shape[0]
        ts["com0"] = self._torch.zeros((B, 2), dtype=f64, device=self.device)
        rs = ReplanState()
        for k, t in ts.items():
            if t is not None:
                setattr(rs, k, t.data_ptr())
        return B, rs, ts

    def replan_sweep(self, prepared, n_stages, use_com_correction=True, disturbance_start=4,
                     outputs=DEFAULT_OUTPUTS, out=None, record=True, stream=None):
        """n_stages dependent replans of every rollout, entirely on the device.  Returns
        (last-stage outputs, per-stage dict(cost, status, n_active), evolved state tensors)."""
        torch = self._torch
        B, rs, ts = prepared
        if out is None:
            out = self._alloc_outputs(B, outputs)
        res, o = out
        stage = None
        ptrs = [None, None, None]
        if record:
            stage = dict(cost=torch.empty((n_stages, B), dtype=torch.float64, device=self.device),
                         status=torch.empty((n_stages, B), dtype=torch.int32, device=self.device),
                         n_active=torch.empty((n_stages, B), dtype=torch.int32, device=self.device))
            ptrs = [stage["cost"].data_ptr(), stage["status"].data_ptr(), stage["n_active"].data_ptr()]
        _check(load_library().mpcb_replan_sweep_device(self._h, B, int(n_stages), int(use_com_correction),
                                                       int(disturbance_start), C.byref(rs), C.byref(o),
                                                       ptrs[0], ptrs[1], ptrs[2], self._stream(stream)))
        return res, stage, ts

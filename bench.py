#!/usr/bin/env python3
"""Benchmark of the batched CoM-MPC QP path on MI355X.

One "step" = one pass of the hot path (footstep rollout -> polygon rows -> QP solve ->
trajectory read-out) over one batch whose inputs already sit in HBM.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus N ...      (starts its N ranks itself: one fresh process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (--workload):
  coupled_vel  (default) 8,192 coupled-XY instances per GPU, N=200 knots, 20 steps, n=400
               variables, m=700 polygon rows, velocity formulation R=0.01 Q=1 (constraints
               bind).  At 8 GPUs this is the 65,536-instance batch.
  coupled_jerk same shape, jerk-only formulation.
  coupled_ref  same shape, ZMP-reference formulation with the driver's weights R=0.01 Q=1
               Qs=100 (no constraint binds: shape fidelity).
  decoupled    4,096 X/Y pairs of box-constrained 1-D QPs per GPU, N=100, m=200 per axis.
  replanning   16,384 rollouts per GPU x 8 sequentially dependent replans (131,072 solves
               per step), velocity formulation with Gaussian weights (window 10), CoM step
               correction, per-rollout Y-jerk disturbance.

Instances are independent, so rank r solves instances [r*B, (r+1)*B) of one global batch
and nothing is exchanged during a step (weak scaling).

Before the W warm-up steps the step is repeated, untimed, for --settle-ms (default 60 ms; the
line reports it): a step of the default workload lasts 0.24 ms, and a GPU that has been idle runs
its first 10-20 ms of work ~5 % below the speed it then keeps.  --settle-ms 0 switches it off.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mpc_amd import batches, planner as P  # noqa: E402
from mpc_amd.provenance import kernel_source_sha  # noqa: E402

# fp64 vector peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop (FMA) per clock at the
# 2.4 GHz maximum clock of MI355X_MICROARCH's chip table.
FP64_FLOP_PER_CLK_PER_CU = 128
HBM_PEAK_GBS = 8000.0
CPU_BASELINE_MIN_S = 3.0  # wall clock the host-core baseline runs for at least (all threads busy)


def measured_profile(workload):
    """Counter-derived figures of an earlier rocprofv3 run of this same command, condensed by
    scripts/summarize_profile.py into profiles/<round>/<workload>_summary.json (the newest
    round that has one wins).  Returns (summary of the dominant kernel, file) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", workload + "_summary.json")))
    if not files:
        return None, None
    try:
        data = json.load(open(files[-1]))
        if data.get("kernel_source_sha") != kernel_source_sha():
            return None, os.path.relpath(files[-1], ROOT)  # counters of other code: not used
        # (the kernels that build a handle's tables run once per handle, outside the timed steps)
        table_builders = ("predict_kernel", "hessian_kernel", "cholesky_kernel", "solve_kernel", "product_kernel",
                          "square_tables_kernel", "small_tables_kernel", "phase_tables_kernel", "knot_reference_kernel")
        kernels = {k: v for k, v in data.get("kernels", {}).items() if not any(t in k for t in table_builders)}
        name = max(kernels, key=lambda k: kernels[k].get("avg_us", 0.0) * kernels[k].get("calls", 0))
        k = dict(kernels[name])
        k["kernel"] = name
        return k, os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def fp64_vector_peak_tflops(dev):
    props = torch.cuda.get_device_properties(dev)
    cus = props.multi_processor_count
    clock_hz = getattr(props, "clock_rate", 2400000) * 1e3  # kHz -> Hz
    if clock_hz <= 0:
        clock_hz = 2.4e9
    return cus * clock_hz * FP64_FLOP_PER_CLK_PER_CU / 1e12, cus, clock_hz


class CoupledWorkload:
    unit = "solves/s"

    def __init__(self, form, B, first, dev):
        self.N, self.steps_per_plan = 200, 20
        self.B = B
        self.form = form
        self.name = {P.COUPLED_VEL: "coupled_vel", P.COUPLED_JERK: "coupled_jerk",
                     P.COUPLED_REF: "coupled_ref"}[form]
        self.inputs = batches.coupled_batch(B, first=first)
        self.pl = P.BatchedPlanner(self.N, formulation=form, weight_R=0.01, weight_Q=1.0,
                                   weight_Qs=100.0, number_of_steps=self.steps_per_plan, device=dev)
        self.prepared = self.pl.prepare_coupled(**self.inputs)
        self.out = self.pl._alloc_outputs(B, P.ALL_OUTPUTS)
        self.units = B
        self.launches = 1
        # algorithmic bytes per instance: 25 input doubles + 6N+2 output doubles + ints
        self.alg_bytes = B * (8 * (25 + 6 * self.N + 2) + 4 * (self.pl.active_stride + 4))

    def step(self):
        self.pl.solve_coupled_prepared(self.prepared, out=self.out)

    def results(self):
        return self.out[0]

    def alg_flops(self):
        return float(self.out[0]["alg_flops"].sum().item())

    def describe(self):
        return {"workload": self.name, "instances_per_gpu": self.B, "horizon": self.N,
                "number_of_steps": 20, "variables": 400, "constraints": self.pl.num_constraints,
                "weights": {"R": 0.01, "Q": 1.0}, "seed": batches.SEED}

    def host_buffers(self, steps=5):
        """The same batch through the host-pointer entry point: inputs from and all outputs
        back to host memory over PCIe inside the timed region (caller-owned arrays, reused)."""
        i = {k: (v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)) for k, v in self.inputs.items()}
        args = (i["x0"], i["y0"], i["speed"], i["feet_x"], i["feet_y"], i["swing0"])
        res = {"unit": self.unit, "entry_point": "mpcb_solve_coupled, every output copied back"}
        for kind in ("pageable", "pinned"):
            out = self.pl.host_output_arrays(self.B, P.ALL_OUTPUTS, pinned=kind == "pinned")
            self.pl.solve_coupled_host(*args, out=out)
            t = time.perf_counter()
            for _ in range(steps):
                self.pl.solve_coupled_host(*args, out=out)
            dt = (time.perf_counter() - t) / steps
            res[kind] = {"value": self.B / dt, "ms_per_step": dt * 1e3}
        res["bytes_per_step"] = int(sum(a.nbytes for a in out.values()) + sum(a.nbytes for a in args))
        return res

    def pipelined(self, steps, parts=2):
        """Side measurement: the same batch as `parts` sub-batches with a handle and a stream each, the
        K steps of each issued back to back with no wait between the streams.  The sub-batches drift
        out of phase, so the kernel of one fills the compute units that the uneven end of the
        other's leaves idle (the cost of an instance varies 1.6x; one kernel over the whole batch
        ends with a fifth of its slots empty).  Every step still solves every instance; results
        are compared bit for bit with those of the one-batch solve."""
        per = self.B // parts
        pls, prep, outs, sts = [], [], [], []
        for p in range(parts):
            sl = slice(p * per, (p + 1) * per)
            pl = P.BatchedPlanner(self.N, formulation=self.form, weight_R=0.01, weight_Q=1.0, weight_Qs=100.0,
                                  number_of_steps=self.steps_per_plan, device=self.pl.device.index)
            pls.append(pl)
            prep.append(pl.prepare_coupled(**{k: v[sl] for k, v in self.inputs.items()}))
            outs.append(pl._alloc_outputs(per, P.ALL_OUTPUTS))
            sts.append(torch.cuda.Stream(device=self.pl.device))

        def run(k):
            for _ in range(k):
                for p in range(parts):
                    pls[p].solve_coupled_prepared(prep[p], out=outs[p], stream=sts[p])

        torch.cuda.synchronize()
        run(max(10, steps))
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        same = all(torch.equal(torch.cat([outs[p][0][k] for p in range(parts)]), self.out[0][k])
                   for k in ("status", "iters", "n_active", "cost", "jerk_x", "jerk_y", "com_x", "zmp_y"))
        return {"value": self.B / dt, "unit": self.unit, "ms_per_step": dt * 1e3, "sub_batches": parts,
                "results_identical_to_one_batch": bool(same),
                "note": "same K steps with the batch solved as %d sub-batches on %d handles and streams, steps issued "
                        "back to back without a wait between the streams (the sub-batches drift out of phase and "
                        "fill the uneven end of each other's kernels); every step solves all instances" % (parts, parts)}

    def cpu_baseline(self, threads, sample):
        from oracle import oracle as O
        p = O.params(self.N, R=0.01, Q=1.0, Qs=100.0, height=0.5, dense=1, iter_cap=0)
        oform = {P.COUPLED_VEL: O.COUPLED_VEL, P.COUPLED_JERK: O.COUPLED_JERK,
                 P.COUPLED_REF: O.COUPLED_REF}[self.form]
        i = self.inputs
        t = time.perf_counter()
        ref = O.coupled_batch(p, oform, self.steps_per_plan, 1000, i["x0"][:sample], i["y0"][:sample],
                              i["speed"][:sample], i["feet_x"][:sample], i["feet_y"][:sample],
                              i["swing0"][:sample], nthreads=threads)
        dt = time.perf_counter() - t
        return sample / dt, ref

    def cpu_structured(self, threads, sample):
        from oracle import oracle as O
        p = O.params(self.N, R=0.01, Q=1.0, Qs=100.0, height=0.5, dense=0, iter_cap=0)
        oform = {P.COUPLED_VEL: O.COUPLED_VEL, P.COUPLED_JERK: O.COUPLED_JERK,
                 P.COUPLED_REF: O.COUPLED_REF}[self.form]
        i = self.inputs
        t = time.perf_counter()
        O.coupled_batch(p, oform, self.steps_per_plan, 1000, i["x0"][:sample], i["y0"][:sample],
                        i["speed"][:sample], i["feet_x"][:sample], i["feet_y"][:sample],
                        i["swing0"][:sample], nthreads=threads)
        return sample / (time.perf_counter() - t)


class ReplanningWorkload:
    unit = "solves/s"
    STAGES = 8

    def __init__(self, B, first, dev):
        self.N, self.steps_per_plan = 200, 20
        self.B = B
        self.name = "replanning"
        self.inputs = batches.coupled_batch(B, first=first)
        self.dist = batches.replanning_disturbance(B, first=first)
        self.hip = batches.NOMINAL_HIP_OFFSETS
        self.pl = P.BatchedPlanner(self.N, formulation=P.COUPLED_VEL, weight_R=0.01, weight_Q=1.0,
                                   replanning_window=self.N // self.steps_per_plan,
                                   number_of_steps=self.steps_per_plan, hip_offsets=self.hip, device=dev)
        self.prepared = self.pl.prepare_replanning(disturbance=self.dist, **self.inputs)
        self.pristine = {k: self.prepared[2][k].clone() for k in ("x0", "y0", "feet_x", "feet_y", "swing0")}
        self.out = self.pl._alloc_outputs(B, P.ALL_OUTPUTS)
        self.units = B * self.STAGES
        self.launches = self.STAGES
        self.alg_bytes = self.STAGES * B * (8 * (25 + 6 * self.N + 2) + 4 * (self.pl.active_stride + 4))
        self.stage = None

    def step(self):
        # (ordering a stage by the one a gait cycle -- four stages -- before it pays from about
        # sixteen stages on: 52.9 -> 51.3 ms for 24; with eight it only costs the ordering kernels)
        for k, t in self.pristine.items():  # every sweep starts from the same initial state
            self.prepared[2][k].copy_(t)
        _, self.stage, _ = self.pl.replan_sweep(self.prepared, self.STAGES, out=self.out)

    def results(self):
        return self.out[0]

    def alg_flops(self):
        """Sum over all stages: the sweep is deterministic, so STAGES one-stage sweeps on the
        evolving state (outside the timed region) visit exactly the stages of the timed sweep."""
        for k, t in self.pristine.items():
            self.prepared[2][k].copy_(t)
        total = 0.0
        for _ in range(self.STAGES):
            self.pl.replan_sweep(self.prepared, 1, out=self.out, record=False)
            total += float(self.out[0]["alg_flops"].sum().item())
        return total

    def describe(self):
        return {"workload": self.name, "rollouts_per_gpu": self.B, "stages": self.STAGES, "horizon": self.N,
                "number_of_steps": 20, "replanning_window": 10, "variables": 400,
                "constraints": self.pl.num_constraints, "weights": {"R": 0.01, "Q": 1.0},
                "com_step_correction": True, "seed": batches.SEED}

    def cpu_sample_text(self, threads, sample):
        return "%d rollouts (x %d stages)" % (max(threads, sample // self.STAGES), self.STAGES)

    def cpu_baseline(self, threads, sample):
        from oracle import oracle as O
        p = O.params(self.N, R=0.01, Q=1.0, height=0.5, dense=1, iter_cap=0)
        i = self.inputs
        sample = max(threads, sample // self.STAGES)
        t = time.perf_counter()
        ref = O.replanning_batch(p, self.steps_per_plan, self.STAGES, i["x0"][:sample], i["y0"][:sample],
                                 i["speed"][:sample], i["feet_x"][:sample], i["feet_y"][:sample],
                                 i["swing0"][:sample], self.dist[:sample], hip=self.hip, nthreads=threads)
        dt = time.perf_counter() - t
        return sample * self.STAGES / dt, ref


class DecoupledWorkload:
    unit = "solves/s"

    def __init__(self, B, first, dev, split=True, parts=0):
        self.N = 100
        if parts:
            self.X_PARTS = parts
        self.B = B
        self.first, self.dev = first, dev
        self.name = "decoupled"
        self.inputs = batches.decoupled_batch(B, N=self.N, steps=10, first=first)
        # one handle per axis: a handle is bound to one stream at a time, and the two axes of a
        # pair are independent QPs, so they run on separate streams (the Y axis fills the CUs that the
        # X axis' second-stage kernel, one workgroup per CU, leaves idle)
        self.pl = P.BatchedPlanner(self.N, formulation=P.BOX_1D, weight_R=1.0, device=dev)
        self.pl_y = P.BatchedPlanner(self.N, formulation=P.BOX_1D, weight_R=1.0, device=dev)
        # The X axis carries the long second stage -- a step lasts as long as its first stage plus
        # its longest instance of the second -- so it gets high-priority streams, and it is solved as
        # X_PARTS sub-batches with a handle and a stream each: the second stage of a sub-batch starts
        # when that sub-batch's first stage is done, not the whole axis'.
        self.parts = self.X_PARTS if split and B % self.X_PARTS == 0 and B >= 1024 else 1
        self.pl_more = [P.BatchedPlanner(self.N, formulation=P.BOX_1D, weight_R=1.0, device=dev)
                        for _ in range(self.parts - 1)]
        self.hi = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(self.parts)]
        self.side = torch.cuda.Stream(device=dev, priority=0)
        # The first sub-batch's stream is also the stream the benchmark runs on (main() makes it
        # the current one).  The streams of a step do not wait for one another: every sub-batch's K
        # steps are issued on its own stream and the timed region ends with a device-wide
        # synchronisation.  (Forking from and joining into one stream around every step puts
        # cross-queue waits on the device; on this platform's HIP runtime such a chain, issued
        # without host synchronisation, stops for good once one of its streams shares a hardware
        # queue with another stream of the process -- scripts/stream_forkjoin_torch_only.py.)
        self.main_stream = self.hi[0]
        i = self.inputs
        self.px = None
        self.py = self.pl_y.prepare_box(i["y0"], i["ymin"], i["ymax"])
        self.outx = self.pl._alloc_outputs(B, P.ALL_OUTPUTS)
        self.outy = self.pl._alloc_outputs(B, P.ALL_OUTPUTS)
        self._prepare_x()
        # three free-running streams: no one stream's events bracket "the dominant kernel's launches",
        # so its duration is taken from the committed kernel trace of the same sources
        self.kernel_time_from_profile = True
        self.units = B  # one unit = one X/Y pair (two scalar QPs)
        self.launches = self.parts + 1  # second-stage launches per step: the X sub-batches' and the Y axis'
        self.alg_bytes = 2 * B * (8 * (3 + 2 * self.N + 2 * self.N + 1) + 4 * (self.pl.active_stride + 4))

    X_PARTS = 2

    def _x_planners(self):
        return [self.pl] + self.pl_more

    def _prepare_x(self):
        """Inputs and output views of the X axis' sub-batches (views into self.outx: the results of the
        axis stay one set of [B, ...] arrays)."""
        i = self.inputs
        n = i["x0"].shape[0]
        per = n // self.parts
        self.px, self.outx_part = [], []
        for p, pl in enumerate(self._x_planners()):
            sl = slice(p * per, (p + 1) * per)
            self.px.append(pl.prepare_box(i["x0"][sl], i["xmin"][sl], i["xmax"][sl]))
            self.outx_part.append(pl.output_views(self.outx, p * per, per))

    def step(self):
        for p, pl in enumerate(self._x_planners()):
            pl.solve_box_prepared(self.px[p], out=self.outx_part[p], stream=self.hi[p])
        self.pl_y.solve_box_prepared(self.py, out=self.outy, stream=self.side)

    def results(self):
        return self.outy[0]

    def alg_flops(self):
        return float(self.outx[0]["alg_flops"].sum().item() + self.outy[0]["alg_flops"].sum().item())

    def describe(self):
        return {"workload": self.name, "xy_pairs_per_gpu": self.B, "horizon": self.N,
                "variables": 100, "constraints": 200, "weights": {"R": 1.0}, "seed": batches.SEED,
                "x_axis_sub_batches": self.parts}

    def second_stage_instances(self):
        self.step()
        torch.cuda.synchronize()
        return {"x_axis": sum(pl.last_handover_count() for pl in self._x_planners()),
                "y_axis": self.pl_y.last_handover_count()}

    def feasible_subset(self, steps=10):
        """Throughput over the pairs whose two QPs are feasible (the infeasible X-axis draws, kept in
        the batch on purpose, each cost hundreds of passes of the second-stage kernel)."""
        ok = ((self.outx[0]["status"] == 0) & (self.outy[0]["status"] == 0)).cpu().numpy()
        i = {k: v[ok] for k, v in self.inputs.items()}
        n = int(ok.sum())
        if n == 0:
            return None
        keep = (self.px, self.py, self.outx, self.outy, self.outx_part, self.inputs, self.parts, self.pl_more)
        self.inputs, self.parts, self.pl_more = i, 1, []  # (an odd number of pairs: one X sub-batch)
        self.py = self.pl_y.prepare_box(i["y0"], i["ymin"], i["ymax"])
        self.outx = self.pl._alloc_outputs(n, P.ALL_OUTPUTS)
        self.outy = self.pl._alloc_outputs(n, P.ALL_OUTPUTS)
        self._prepare_x()
        for _ in range(2):
            self.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        self.px, self.py, self.outx, self.outy, self.outx_part, self.inputs, self.parts, self.pl_more = keep
        return {"pairs": n, "value": n / dt, "unit": "solves/s", "ms_per_step": dt * 1e3}

    def alternating_sets(self, steps):
        """The same K steps issued alternately to two complete sets of handles, streams and outputs,
        so that a step's first stage runs while the previous step's second stage -- one long
        instance on one compute unit for most of its time -- is still finishing.  What a caller that
        streams batches gets by alternating handles; not `value`, whose steps each run on one set."""
        twin = DecoupledWorkload(self.B, self.first, self.dev, split=self.parts > 1, parts=self.parts)
        sets = (self, twin)
        for k in range(4):
            sets[k % 2].step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(steps):
            sets[k % 2].step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        same = all(bool((twin.outx[0][k] == self.outx[0][k]).all().item()) for k in ("status", "iters", "jerk_x"))
        return {"value": self.B / dt, "unit": "solves/s", "ms_per_step": dt * 1e3, "sets": 2,
                "results_identical": same,
                "note": "K steps issued alternately to two sets of handles, streams and output arrays (no waits "
                        "between them): consecutive steps overlap"}

    def cpu_baseline(self, threads, sample):
        from oracle import oracle as O
        p = O.params(self.N, R=1.0, height=0.5, dense=1, iter_cap=3000)
        i = self.inputs
        t = time.perf_counter()
        ref = O.box_batch(p, i["x0"][:sample], i["y0"][:sample], i["xmin"][:sample], i["xmax"][:sample],
                          i["ymin"][:sample], i["ymax"][:sample], nthreads=threads)
        dt = time.perf_counter() - t
        return sample / dt, ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="coupled_vel",
                    choices=("coupled_vel", "coupled_jerk", "coupled_ref", "decoupled", "replanning"))
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (0 = workload default)")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl (= RCCL) for one rank per GPU; gloo lets several ranks share a GPU "
                         "for rehearsals")
    ap.add_argument("--host-steps", type=int, default=5,
                    help="steps of the host-pointer (PCIe-inclusive) side measurement at N=1; 0 = skip")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="instances timed on the host cores (0 disables, -1 = default)")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed repetition of the step before the warm-up, for this long (0 = none)")
    ap.add_argument("--x-split", default="auto", choices=("auto", "on", "off"),
                    help="decoupled workload: X axis as two sub-batches on two high-priority streams (auto: on)")
    ap.add_argument("--x-parts", type=int, default=0, help="decoupled workload: number of X sub-batches (0 = default)")
    ap.add_argument("--plain", action="store_true",
                    help="warm-up and timed steps only (for profiler runs): no side measurements, no CPU baseline")
    args = ap.parse_args()
    if args.plain:
        args.host_steps, args.cpu_sample, args.settle_ms = 0, 0, 0.0

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    started_as_rank = "RANK" in os.environ or world > 1
    if not started_as_rank and args.gpus > 1:
        # `python bench.py --gpus N` by itself: this process becomes the starter of N ranks and
        # nothing else.  It makes no GPU call before or after (counting devices does not
        # initialise the runtime), so no process that touched the GPU is ever replaced or forked.
        from mpc_amd import launch
        have = torch.cuda.device_count()
        if have < args.gpus and args.dist_backend != "gloo":
            sys.stderr.write("bench.py: --gpus %d asked for, %d device(s) visible; one rank per GPU needs %d "
                             "(--dist-backend gloo lets ranks share a GPU for rehearsals)\n"
                             % (args.gpus, have, args.gpus))
            sys.exit(2)
        sys.exit(launch.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    if started_as_rank and args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d does not match WORLD_SIZE=%d of the launcher\n" % (args.gpus, world))
        sys.exit(2)
    # under torch.distributed.run (RANK and MASTER_ADDR set) the process group is used even with
    # one rank, so that a one-GPU box exercises the same code as an eight-GPU node
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    dev = local if distributed else 0
    torch.cuda.set_device(dev)

    if args.workload == "decoupled":
        B = args.batch or 4096
        split = args.x_split != "off"
        wl = DecoupledWorkload(B, rank * B, dev, split=split, parts=args.x_parts)
    elif args.workload == "replanning":
        B = args.batch or 16384
        wl = ReplanningWorkload(B, rank * B, dev)
    else:
        B = args.batch or 8192
        form = {"coupled_vel": P.COUPLED_VEL, "coupled_jerk": P.COUPLED_JERK,
                "coupled_ref": P.COUPLED_REF}[args.workload]
        wl = CoupledWorkload(form, B, rank * B, dev)

    if hasattr(wl, "main_stream"):
        torch.cuda.set_stream(wl.main_stream)  # the workload forks from and joins into this stream

    def barrier():
        if distributed:
            dist.barrier()

    # Device settling, before the W warm-up steps and reported in the line: a step of the headline
    # workload takes 0.24 ms, so W = 5 of them leave a GPU that was idle until now 1.2 ms to
    # raise its clocks, and the first 10-20 ms of work run ~5 % slower than everything after.
    # The same steps are repeated for a fixed time first; nothing of it is timed or counted.
    # (The first collective of a process group sets up its communicator, which takes far longer
    # than a step and leaves the GPU idle meanwhile: it happens here, so that the barrier in front
    # of the timed region is a short one.)
    barrier()
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        wl.step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        wl.step()
    # (the first timing event of a process is expensive to record: not inside the timed region)
    ev_init = torch.cuda.Event(enable_timing=True)
    ev_init.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        wl.step()
    t_issued = time.perf_counter()
    ev1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()  # this rank's K steps are done; the MAX over ranks below is the job's time
    barrier()
    elapsed = t1 - t0
    dev_ms = ev0.elapsed_time(ev1)  # device time of the K steps on the launch stream
    if distributed:
        tt = torch.tensor([elapsed, dev_ms], dtype=torch.float64,
                          device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = tt[0].item(), tt[1].item()

    # Side measurement (one GPU only, after the timed region): the same steps with
    # mpcb_set_dispatch_hint(LONGEST_FIRST).  Every step solves the identical batch, so the order
    # taken from the previous step's running times is as good as it can get -- a caller with
    # unrelated batches gains nothing from the hint, which is why this is not `value`.
    longest_first = None
    second_pass = None
    pipelined = None
    base_hint = getattr(wl, "dispatch_hint", P.DISPATCH_IN_ORDER)
    if world == 1 and not args.plain:
        planners = [getattr(wl, a) for a in ("pl", "pl_y") if hasattr(wl, a)] + list(getattr(wl, "pl_more", []))
        # (a workload may set `dispatch_hint` as part of its configuration: the comparison is then
        # with index order)
        other = P.DISPATCH_LONGEST_FIRST if base_hint == P.DISPATCH_IN_ORDER else P.DISPATCH_IN_ORDER
        for pl in planners:
            pl.set_dispatch_hint(other)
        for _ in range(max(3, args.warmup)):
            wl.step()
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        torch.cuda.synchronize()
        dt_h = (time.perf_counter() - t_h) / args.steps
        for pl in planners:
            pl.set_dispatch_hint(base_hint)
        for _ in range(2):
            wl.step()  # the results reported below are those of the timed configuration
        torch.cuda.synchronize()
        if other == P.DISPATCH_LONGEST_FIRST:
            note = ("same steps with mpcb_set_dispatch_hint(LONGEST_FIRST): the first-stage kernel starts the "
                    "instances in decreasing order of the running times the previous step recorded for the "
                    "identical batch (scheduling only, results identical)")
        else:
            note = "same steps with every stage dispatched in index order (mpcb_set_dispatch_hint(0))"
        longest_first = {"value": wl.units / dt_h, "unit": wl.unit, "ms_per_step": dt_h * 1e3, "dispatch_hint": other,
                         "note": note}
        # ... and the timed steps once more, now that the GPU has been busy for a while: on a
        # GPU that was idle until the warm-up the first milliseconds of work run a few percent
        # slower than the steady state
        t_r = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        torch.cuda.synchronize()
        dt_r = (time.perf_counter() - t_r) / args.steps
        second_pass = {"value": wl.units / dt_r, "unit": wl.unit, "ms_per_step": dt_r * 1e3,
                       "note": "the same K steps timed again at the end of the run"}
        if hasattr(wl, "pipelined") and wl.B % 2 == 0 and wl.B >= 2048:
            pipelined = wl.pipelined(args.steps)

    res = wl.results()
    epilogue = None
    if distributed:
        # optional epilogue, outside the timed region: the only collective of the job
        from mpc_amd import distributed as D
        t_e = time.perf_counter()
        best, where = D.best_cost(res["cost"], res["status"], rank * wl.B)
        torch.cuda.synchronize()
        epilogue = {"best_cost": best, "instance": where, "ms": (time.perf_counter() - t_e) * 1e3}
    status = res["status"].cpu().numpy()
    n_active = res["n_active"].cpu().numpy()
    iters = res["iters"].cpu().numpy()
    alg_flops = wl.alg_flops()

    if rank == 0:
        peak_tf, cus, clk = fp64_vector_peak_tflops(dev)
        kernel_s = dev_ms / 1e3 / (args.steps * wl.launches)  # average launch duration
        achieved_tf = alg_flops / wl.launches / kernel_s / 1e12
        value = wl.units * world / (elapsed / args.steps)
        line = {
            "metric": "constrained CoM QP solves/sec",
            "value": value,
            "unit": wl.unit,
            "n_gpus": world,
            "devices_visible": torch.cuda.device_count(),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": wl.describe(),
            # rank 0's clocks of the timed region: wall, the device's span between the two events
            # around the K steps, and when the host had issued the last launch
            "settle_ms": args.settle_ms,
            "timing": {"wall_ms": (t1 - t0) * 1e3, "device_span_ms": ev0.elapsed_time(ev1),
                       "all_launches_issued_ms": (t_issued - t0) * 1e3},
            "roofline": {
                "bound": "fp64_vector",
                # filled in below from the counter profile of THIS source (fp64 operations the
                # dominant kernel really issued / its duration measured in this run); stays null
                # when no profile of the current kernel sources is committed
                "achieved": None,
                "peak": peak_tf,
                "unit": "TFLOP/s",
                "frac": None,
                "traffic": None,
                "kernel_ms": kernel_s * 1e3,
                "peak_basis": "%d CUs x %.2f GHz x %d fp64 flop/clk/CU" % (cus, clk / 1e9,
                                                                         FP64_FLOP_PER_CLK_PER_CU),
                "hbm_achieved_GBs": wl.alg_bytes / wl.launches / kernel_s / 1e9,
                "hbm_frac": wl.alg_bytes / wl.launches / kernel_s / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": wl.alg_bytes / wl.launches,
                # speed of the algorithm, NOT utilisation: operations the dense dual active-set
                # method (the reference's) would spend along each instance's pivot sequence, per
                # second of this kernel.  The kernel avoids almost all of them, so the ratio to
                # the peak exceeds 1.
                "alg_gflop_per_launch": alg_flops / wl.launches / 1e9,
                "alg_tflops": achieved_tf,
                "alg_speed_vs_peak": achieved_tf / peak_tf,
            },
            "solver": {
                "status_counts": np.bincount(status, minlength=4).tolist(),
                "working_set_histogram": np.bincount(n_active).tolist(),
                "working_set_mean": float(n_active.mean()),
                "working_set_max": int(n_active.max()),
                "outer_iterations_mean": float(iters.mean()),
                # instances of the last call that needed the dense second-stage kernel
                "second_stage_instances": (wl.second_stage_instances() if hasattr(wl, "second_stage_instances")
                                           else wl.pl.last_handover_count()),
            },
        }
        if world > torch.cuda.device_count():
            line["ranks_share_a_gpu"] = ("%d ranks on %d device(s) (gloo rehearsal of the N>1 code path): `value` is "
                                         "not a scaling figure" % (world, torch.cuda.device_count()))
        if epilogue is not None:
            line["epilogue_best_cost"] = epilogue
        if longest_first is not None:
            line["longest_first_dispatch" if longest_first["dispatch_hint"] else "index_order_dispatch"] = longest_first
        if second_pass is not None:
            line["second_pass"] = second_pass
        if pipelined is not None:
            line["pipelined_sub_batches"] = pipelined
        if world == 1 and args.host_steps > 0 and hasattr(wl, "host_buffers"):
            line["host_buffers"] = wl.host_buffers(args.host_steps)
        if hasattr(wl, "alternating_sets") and not args.plain and world == 1:
            line["alternating_sets"] = wl.alternating_sets(args.steps)
        if hasattr(wl, "feasible_subset") and not args.plain:
            line["feasible_subset"] = wl.feasible_subset()
        prof, prof_file = measured_profile(args.workload)
        r = line["roofline"]
        r["kernel_source_sha"] = kernel_source_sha()
        if prof is None:
            r["note"] = ("no counter profile of the current kernel sources under profiles/ (%s): executed "
                         "flops and HBM traffic unknown for this run" % (prof_file or "none found"))
        else:
            r["profile_source"] = prof_file
            r["profiled_kernel"] = prof.get("kernel")
            r["profiled_kernel_avg_ms"] = (prof.get("avg_us") or 0.0) / 1e3 or None
            r["traffic"] = prof.get("hbm_traffic_bytes_per_launch")
            r["traffic_unit"] = "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE), rocprofv3 pass of the same source"
            ex = prof.get("executed_f64_flops_per_launch")
            # the counters belong to the profiled batch; scale to this run's units
            scale = wl.units / wl.launches / max(1.0, float(prof.get("units_per_launch") or wl.units / wl.launches))
            if ex and getattr(wl, "kernel_time_from_profile", False) and r["profiled_kernel_avg_ms"]:
                # (its launches run on several free-running streams: no event pair of this run times them)
                r["kernel_ms"] = r["profiled_kernel_avg_ms"]
                r["executed_f64_gflop_per_launch"] = ex / 1e9
                r["achieved"] = ex / (r["profiled_kernel_avg_ms"] / 1e3) / 1e12
                r["frac"] = r["achieved"] / peak_tf
                r["note"] = ("achieved = fp64 operations the dominant kernel issued per launch (SQ_INSTS_VALU_{FMA,ADD,"
                             "MUL}_F64 x 64 lanes, FMA = 2) / its average launch duration, both from the rocprofv3 passes "
                             "of the same kernel sources (profile_source): the step runs on three streams that do not "
                             "wait for one another, so no event pair of this run brackets that kernel's launches")
            elif ex:
                r["executed_f64_gflop_per_launch"] = ex * scale / 1e9
                r["achieved"] = ex * scale / kernel_s / 1e12
                r["frac"] = r["achieved"] / peak_tf
                if r["profiled_kernel_avg_ms"]:
                    # the same arithmetic with the profile's own duration (rocprofv3 adds a few microseconds
                    # to every launch): what someone recomputing from profiles/ alone arrives at
                    r["frac_profile_duration"] = ex / (r["profiled_kernel_avg_ms"] / 1e3) / 1e12 / peak_tf
                r["note"] = ("achieved = fp64 operations the kernel issued (SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 x 64 "
                             "lanes, FMA = 2; counter pass of the same kernel sources, see profile_source) / "
                             "average launch duration measured in this run with HIP events")
            c = prof.get("counters_per_launch", {})
            if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_WAVE_CYCLES"):
                r["valu_busy_per_wave"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
                r["wave_wait_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
                r["fp64_share_of_valu"] = ((c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) +
                                            c.get("SQ_INSTS_VALU_MUL_F64", 0)) / max(1.0, c.get("SQ_INSTS_VALU", 1.0)))
        sample = args.cpu_sample
        default_sample = sample < 0
        if default_sample:
            sample = 4096 if args.workload == "decoupled" else 1024
        if world == 1 and sample > 0:
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            sample = min(sample, wl.B)
            # the sample is solved again until a few seconds of wall clock (x `threads` cores of
            # CPU work) have gone by: one pass of the short workloads lasts a fraction of a second,
            # too little for a steady figure on a shared host
            reps, cpu_s, solved = 0, 0.0, 0.0
            while reps < (12 if default_sample else 1) and cpu_s < CPU_BASELINE_MIN_S:
                t_c = time.perf_counter()
                rate1, _ = wl.cpu_baseline(threads, sample)
                dt_c = time.perf_counter() - t_c
                solved += rate1 * dt_c  # units the pass solved (a workload may round the sample)
                cpu_s += dt_c
                reps += 1
            rate = solved / cpu_s
            line["cpu_baseline"] = {
                "value": rate, "unit": wl.unit, "cores": threads, "kind": "port",
                "sample": "first %s of the same batch, solved %d time(s) (%.1f s wall on %d threads), "
                          "dense restatement of MPCPlanner + eiquadprog (oracle, dense=1), one planner per "
                          "thread" % (wl.cpu_sample_text(threads, sample) if hasattr(wl, "cpu_sample_text")
                                      else "%d instances" % sample, reps, cpu_s, threads),
                "per_core": rate / threads,
            }
            line["speedup_vs_cpu_baseline"] = value / rate
            if hasattr(wl, "cpu_structured"):
                srate = wl.cpu_structured(threads, sample)
                line["cpu_structured"] = {
                    "value": srate, "unit": wl.unit, "cores": threads,
                    "note": "same restatement with its zero-skipping matrix products (identical "
                            "results); shows how much of the ratio is avoided arithmetic",
                }
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

/*
 * mpc_amd.h -- C ABI of the batched CoM-MPC QP solver for AMD MI355X (gfx950).
 *
 * The reference (mfocchi/mpc) has no plugin or FFI layer; the boundary this
 * library sits behind is the public surface of class MPCPlanner
 * (include/crawl_planner/MPCPlanner.h:47-143).  Each entry point below names the
 * MPCPlanner methods whose work it performs for a whole batch of independent
 * planner calls.  Plain C: pointers, sizes, no C++ or torch types.
 *
 * Conventions
 *   - Per-instance vectors are rows: array[instance][component], contiguous.
 *     One wavefront works on one instance, so a row is read/written by 64 lanes
 *     at consecutive addresses.
 *   - Legs are in iit::dog::LegID order LF=0, RF=1, LH=2, RH=3
 *     (include/crawl_planner/leg_data_map.h:19).
 *   - `*_device` entry points take DEVICE pointers, enqueue work on `stream`
 *     (a hipStream_t passed as void*; NULL = default stream) and return without
 *     synchronising.  The other entry points take HOST pointers, copy, solve and
 *     synchronise.
 *   - Return value: 0 (or a non-negative count where stated) on success, a
 *     negative MPCB_E_* code on failure; mpcb_last_error() gives the message.
 *   - Thread compatibility: one handle may be used by one host thread at a time, and by one
 *     stream at a time: the hand-over list and the dense workspace belong to the handle, so
 *     two `*_device` calls of one handle in flight on different streams would share them.
 *     Distinct handles (also on one device) are independent.
 */
#ifndef MPC_AMD_H_
#define MPC_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Formulations.  The numbering is this library's own. */
enum {
    MPCB_BOX_1D       = 0, /* MPCPlanner::solveQPconstraint              (MPCPlanner.cpp:240-300) */
    MPCB_COUPLED_JERK = 1, /* MPCPlanner::solveQPConstraintCoupled, 7-arg (MPCPlanner.cpp:427-490) */
    MPCB_COUPLED_VEL  = 2, /* MPCPlanner::solveQPConstraintCoupled, 9-arg (MPCPlanner.cpp:493-628) */
    MPCB_COUPLED_REF  = 3, /* MPCPlanner::solveQPConstraintCoupledRef     (MPCPlanner.cpp:632-783) */
    MPCB_UNCONSTRAINED = 4,/* MPCPlanner::solveQP                         (MPCPlanner.cpp:223-237) */
    MPCB_BOX_SLACK    = 5  /* MPCPlanner::solveQPconstraintSlack          (MPCPlanner.cpp:303-424) */
};

enum { MPCB_MODE_AUTO = 0, MPCB_MODE_DENSE = 1 };

/* Where the instance-independent tables of a handle are built (mpcb_params.table_build). */
enum {
    MPCB_TABLES_DEVICE = 0, /* on the GPU, double-double arithmetic (default)                  */
    MPCB_TABLES_HOST   = 1  /* on the host in long double, one thread: the checker of the above */
};

/* Per-instance status written to mpcb_outputs.status. */
enum {
    MPCB_OK         = 0,
    MPCB_INFEASIBLE = 1, /* solve_quadprog returned +inf (eiquadprog.hpp:350-356); outputs zero-filled */
    MPCB_ITER_CAP   = 2  /* iteration cap hit (the reference has no cap and would keep going) */
};

/* Error codes (negative returns). */
enum {
    MPCB_E_ARG    = -1, /* bad argument / unsupported size */
    MPCB_E_HIP    = -2, /* a HIP runtime call failed */
    MPCB_E_NUMERIC = -3, /* the Hessian is not positive definite */
    MPCB_E_NOGPU  = -4  /* no usable device */
};

/* Constructor arguments.  Replaces MPCPlanner::MPCPlanner (MPCPlanner.cpp:22-70),
 * setWeights (:94-98), setHipOffsets (:1352-1355) and the `actual_height`,
 * `replanningWindow`, `number_of_steps` arguments of the solve/computeSteps calls,
 * which are the same for every instance of a batch. */
typedef struct {
    int32_t horizon;            /* horizon_size_                                   */
    int32_t formulation;        /* MPCB_*                                          */
    double  Ts;                 /* sample time                                     */
    double  gravity;
    double  height;             /* actual_height                                   */
    double  weight_R;           /* jerk weight                                     */
    double  weight_Q;           /* velocity weight                                 */
    double  weight_Qs;          /* ZMP-reference weight (COUPLED_REF), slack weight (BOX_SLACK) */
    int32_t replanning_window;  /* 1000 = constant velocity weights (MPCPlanner.cpp:535);
                                   otherwise Gaussian weights, makeGaussian (:1161-1171) */
    int32_t number_of_steps;    /* computeSteps' number_of_steps (coupled only).  0 = no
                                   footstep schedule: every knot k owns rows 4k .. 4k+3
                                   (m = 4N); with rows_pqr the caller fills them, e.g. from
                                   a time-based gait, a row that does not exist being
                                   (0, 0, +INFINITY); without rows_pqr all knots get the
                                   polygon of the initial feet                       */
    double  hip_offsets[8];     /* [leg][x,y]; the reference reads them uninitialised
                                   unless setHipOffsets was called; here explicit  */
    int32_t iter_cap;           /* cap on outer iterations and on step-direction passes;
                                   0 = default (50*(n+m))                          */
    int32_t device;             /* HIP device ordinal                              */
    int32_t solver_mode;        /* MPCB_MODE_AUTO: structure-exploiting kernel first, dense
                                   kernel for what it hands over; MPCB_MODE_DENSE: every
                                   instance through the dense kernel (slow; cross-check)   */
    int32_t table_build;        /* MPCB_TABLES_DEVICE (default) or MPCB_TABLES_HOST        */
} mpcb_params;

typedef struct mpcb_handle_s *mpcb_handle;

/* Inputs of a batch of coupled planner calls: the arguments of computeSteps
 * (MPCPlanner.cpp:1188-1193) and of solveQPConstraintCoupled[Ref].            */
typedef struct {
    const double  *x0;        /* [B][3] initial_state_x (pos, vel, acc)                */
    const double  *y0;        /* [B][3] initial_state_y                                 */
    const double  *speed;     /* [B][2] userSpeed == targetSpeed (per-step displacement
                                        in computeSteps, velocity target in the QP)    */
    const double  *feet_x;    /* [B][4] initial_feet_x                                  */
    const double  *feet_y;    /* [B][4] initial_feet_y                                  */
    const int32_t *swing0;    /* [B]    swing_leg_index (LegID)                         */
    const double  *com0;      /* [B][2] initialCoM; (0,0) = no step correction; may be NULL */
    /* Optional.  When rows_pqr is given the library does NOT run computeSteps: the polygon
     * rows are the caller's, one (p, q, r) per row of the A, b that computeSteps returned
     * (A(row, k) = p, A(row, k + N) = q, b(row) = r; MPCPlanner.cpp:1072-1079), in the same
     * row order (mpcb_num_constraints rows); feet_x, feet_y, swing0, com0 are then unused. */
    const double  *rows_pqr;  /* [B][m][3] or NULL                                       */
    /* Optional ZMP reference for COUPLED_REF, constant over each phase of computeSteps
     * (2*number_of_steps + 1 phases: four-stance, three-stance, ..., trailing knots);
     * NULL = centroid of the stance feet (computeCentroid).  With rows_pqr give this or
     * zref_knot.                                                                          */
    const double  *zref_phase; /* [B][nphase][2] or NULL                                 */
    /* Optional CoM height per instance: the `actual_height` argument the reference takes on
     * every solve call (MPCPlanner.cpp:493-502,516).  NULL = mpcb_params.height for all.
     * COUPLED_JERK and COUPLED_VEL: their Hessian does not depend on the height, so one
     * handle serves every height (any entry point).  COUPLED_REF: the Hessian contains
     * Qs Zu'Zu, so every distinct height is a parameter group of its own; the host-pointer
     * entry point mpcb_solve_coupled sorts the instances by height, builds the tables of
     * heights it has not seen before (together, on the GPU; at most 1024 are kept) and
     * solves the groups in one go; the device-pointer entry points take such a batch as
     * explicit groups (mpcb_create_groups + mpcb_solve_coupled_groups).                    */
    const double  *height;     /* [B] or NULL                                             */
    /* Optional ZMP reference for COUPLED_REF that changes from knot to knot: the zmp_ref_x,
     * zmp_ref_y vectors of solveQPConstraintCoupledRef (MPCPlanner.cpp:632-640), any values.
     * Excludes zref_phase.  Costs one N x N table sweep per instance where the per-phase form
     * needs nphase rows; the first call that passes it builds its tables (one-off).          */
    const double  *zref_knot;  /* [B][N][2] or NULL                                       */
    /* Optional: the caller's own support polygons, one per footstep phase, in place of
     * computeSteps -- what a caller of the reference builds with buildPolygonMatrix
     * (MPCPlanner.cpp:1040-1082) from footholds of its own (a margin, an edited polygon, another
     * foothold planner) and passes to solveQPConstraintCoupled as A, b.  Edge e of phase ph is
     * (p, q, r) = phase_pqr[i][ph][e][0..2]; knot k of phase ph owns the rows
     * p zmp_x[k] + q zmp_y[k] + r >= 0 for its edges, in the row order of computeSteps
     * (mpcb_num_constraints rows: four edges for a four-stance phase -- the even ones and the
     * trailing one -- three for a swing phase, whose fourth entry is ignored).  Needs
     * number_of_steps > 0; excludes rows_pqr; every edge that the row order counts must be a real
     * half-plane (the stopping rule's threshold is proportional to the number of rows: use rows_pqr
     * with (0, 0, +INFINITY) entries for polygons with fewer edges); feet_x, feet_y, swing0, com0 are
     * then unused, and
     * COUPLED_REF needs zref_phase or zref_knot.  The polygons are read into the kernels' per-phase
     * table, so a batch runs at the speed of the library's own rows (rows_pqr, which may differ
     * from knot to knot, are fetched from memory in every iteration: about half that speed).  */
    const double  *phase_pqr;  /* [B][nphase][4][3] or NULL                               */
} mpcb_coupled_inputs;

/* Outputs.  Any pointer may be NULL (that output is skipped). */
typedef struct {
    double  *jerk_x, *jerk_y;   /* [B][N] jerk_vector_x / _y                              */
    double  *com_x, *com_y;     /* [B][N] computeCOMtrajectory(..., POSITION)   (:196-205) */
    double  *com_xd, *com_yd;   /* [B][N] computeCOMtrajectory(..., VELOCITY)   (:206-209) */
    double  *zmp_x, *zmp_y;     /* [B][N] computeZMPtrajectory                  (:169-174) */
    double  *min_margin;        /* [B][N] getConstraintViolation                (:1132-1153) */
    double  *cost;              /* [B]    value returned by solve_quadprog (1/2 x'Gx + g0'x) */
    int32_t *status;            /* [B]    MPCB_OK / _INFEASIBLE / _ITER_CAP                */
    int32_t *iters;             /* [B]    outer iterations (eiquadprog.hpp:240)            */
    int32_t *n_active;          /* [B]    size of the final working set                    */
    int32_t *active_idx;        /* [B][active_stride] constraint rows of the final working
                                          set in the order the solver holds them, -1 padded */
    double  *multipliers;       /* [B][active_stride] their Lagrange multipliers          */
    int32_t  active_stride;
    double  *alg_flops;         /* [B]    diagnostic: floating-point operations the dense
                                          dual active-set method (eiquadprog.hpp) spends along
                                          this instance's pivot sequence; used for roofline
                                          reporting, not a count of what the GPU executed      */
    double  *slacks;            /* [B][2N] BOX_SLACK only: the slack variables (lower-bound
                                          rows first), the `slacks` member of MPCPlanner.h:191 */
} mpcb_outputs;

/* Builds every instance-independent matrix of one parameter set -- what MPCPlanner::buildMatrix
 * (MPCPlanner.cpp:101-156), the Hessian assembly of the solve methods (:233, :523-559, :723-739)
 * and the factorisation inside solve_quadprog (eiquadprog.hpp:165-176) recompute on every call:
 * prediction matrices, condensed Hessian block, its Cholesky factor, L^-T and the response
 * matrices -- on the GPU (mpcb_params.table_build). */
int  mpcb_create(const mpcb_params *params, mpcb_handle *out);
/* The same for n parameter sets at once (horizons, weights, heights, formulations may all differ;
 * one device): the tables of all of them are built by one chain of launches, one grid row per
 * set, so a sweep over H heights or weights costs about what one set costs.  out[0..n) receive
 * the handles; on failure none is created.  Each handle is destroyed with mpcb_destroy. */
int  mpcb_create_groups(const mpcb_params *params, int32_t n, mpcb_handle *out);
void mpcb_destroy(mpcb_handle h);
const char *mpcb_last_error(void);

/* Number of inequality rows m of one instance (700 for N=200, 20 steps; 2N for BOX_1D). */
int  mpcb_num_constraints(mpcb_handle h);
/* Number of footstep phases of one coupled instance (2*number_of_steps + 1). */
int  mpcb_num_phases(mpcb_handle h);

/* Coupled XY plan: computeSteps -> [computeCentroid] -> solveQPConstraintCoupled[Ref]
 * -> getConstraintViolation -> computeZMP/COMtrajectory for B instances. */
int  mpcb_solve_coupled_device(mpcb_handle h, int32_t B, const mpcb_coupled_inputs *in,
                               const mpcb_outputs *out, void *stream);
int  mpcb_solve_coupled(mpcb_handle h, int32_t B, const mpcb_coupled_inputs *in,
                        const mpcb_outputs *out);

/* The same as mpcb_solve_coupled_device for callers that cannot allocate device memory --
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  No
 * counterpart in the reference.  mpcb_solve_coupled_device sizes its hand-over list and the
 * dense second stage's J / R slots with hipMalloc on first use and when a batch outgrows them;
 * this entry point keeps them in a static device array instead (47.25 MiB per GPU, held from
 * the moment the library is loaded on that GPU until the process ends) and calls no hipMalloc,
 * hipFree or other allocating routine.  Limits, rejected with MPCB_E_ARG: B <= 65536 and at
 * most 400 variables per QP (horizon <= 200); the dense stage has 8 slots, so a batch that
 * hands many instances over runs that stage 8 at a time.  Every call on a GPU -- from any
 * handle, stream or thread -- uses that GPU's one array: calls that overlap on one GPU race;
 * serialise them (one stream, or events).  Results are bit-identical to
 * mpcb_solve_coupled_device.  mpcb_last_handover_count does not see these calls. */
int  mpcb_solve_coupled_device_noalloc(mpcb_handle h, int32_t B, const mpcb_coupled_inputs *in,
                                       const mpcb_outputs *out, void *stream);

/* One batch over several GPUs from one host thread's point of view (HOST pointers): instance i
 * is solved with handles[floor(i * n_handles / B)] -- contiguous shards, no exchange between
 * devices, one internal host thread and one HIP stream per handle; returns when every shard is
 * back.  The handles must have been created with the same shape and formulation, normally one
 * per device ordinal (mpcb_params.device); two handles on the same ordinal also work.  The
 * reference is single-threaded and has no counterpart; this is the multi-device form of
 * mpcb_solve_coupled.  Returns the number of non-OK instances or a negative error. */
int  mpcb_solve_coupled_multi(const mpcb_handle *handles, int32_t n_handles, int32_t B,
                              const mpcb_coupled_inputs *in, const mpcb_outputs *out);

/* The device-pointer form of mpcb_solve_coupled_multi: shard g -- B[g] instances whose inputs and
 * outputs (DEVICE pointers on handles[g]'s device) are in[g] and out[g] -- is enqueued with
 * handles[g] on streams[g] (a hipStream_t of that device; streams == NULL or streams[g] == NULL:
 * that device's default stream).  Nothing is copied, no host thread is started and nothing is
 * waited for: the call returns when every shard is enqueued, each device then works through its
 * own shard and the caller synchronises the streams (or devices) it passed.  This is how a C
 * caller reaches the HBM-resident rate on several GPUs at once; the handles may differ in every
 * parameter.  No counterpart in the reference.  Returns 0 or a negative error (shards enqueued
 * before the failing one still run). */
int  mpcb_solve_coupled_multi_device(const mpcb_handle *handles, int32_t n_handles, const int32_t *B,
                                     const mpcb_coupled_inputs *in, const mpcb_outputs *out,
                                     void *const *streams);

/* Several parameter groups in one call on ONE device: group g is B[g] instances solved with
 * handles[g] (DEVICE pointers in[g], out[g]).  The reference changes horizon, weights and height
 * from one planner object or call to the next (MPCPlanner.cpp:78-98, 493-502); a batch that mixes
 * them is a list of groups, one handle per distinct parameter set (mpcb_create_groups builds the
 * handles' tables together).  The groups are enqueued back to back on `stream`.  With
 * MPCB_GROUP_LANES=2 in the environment they are spread over `stream` and two library-owned
 * high-priority streams with no wait between groups (kernels of different groups overlap and fill
 * each other's uneven ends: 7 % for four groups of 2,048), `stream` continuing only after all of
 * them (one fork before the first group, one join after the last).  That is opt-in: on this
 * platform's HIP runtime a fork/join issued call after call without a host synchronisation wedges
 * the device queues once the calling stream shares a hardware queue with another stream, i.e. in
 * a process with more streams than hardware queues (four per priority level by default).  All
 * handles must live on the same device and be distinct.  Returns 0 or a negative error. */
int  mpcb_solve_coupled_groups(const mpcb_handle *handles, int32_t n_groups, const int32_t *B,
                               const mpcb_coupled_inputs *in, const mpcb_outputs *out, void *stream);

/* State of B receding-horizon rollouts between two replans (device pointers, updated in
 * place): what the loop of tests/replanning/testPlannerXYconstraintMPC.cpp keeps in
 * actual_state_x/y, initial_feet_x/y and replanning_schedule. */
typedef struct {
    double        *x0, *y0;       /* [B][3] actual CoM state                                  */
    const double  *speed;         /* [B][2] userSpeed                                         */
    double        *feet_x, *feet_y; /* [B][4]                                                 */
    int32_t       *swing0;        /* [B]    leg the next plan starts swinging                 */
    double        *com0;          /* [B][2] scratch: CoM handed to computeSteps               */
    const double  *disturbance;   /* [B]    added to the Y jerk from sample disturbance_start
                                            on (testPlannerXYconstraintMPC.cpp:104,207); NULL = none */
} mpcb_replan_state;

/* n_stages sequentially dependent replans of B rollouts without leaving the device
 * (testPlannerXYconstraintMPC.cpp:108-209 with optimizeVelocity on): per stage
 * computeSteps from the actual CoM (step correction if use_com_correction) ->
 * solveQPConstraintCoupled -> computeCOMupdate over floor(N/number_of_steps) samples ->
 * feet taken from the plan, schedule advanced.  `out` receives the LAST stage's plan and
 * must provide jerk_x, jerk_y, cost and status; stage_cost / stage_status / stage_nactive
 * ([n_stages][B], any may be NULL) record every stage.  The handle must be COUPLED_VEL or
 * COUPLED_JERK.  Infeasible stages contribute zero jerk. */
int  mpcb_replan_sweep_device(mpcb_handle h, int32_t B, int32_t n_stages, int32_t use_com_correction,
                              int32_t disturbance_start, const mpcb_replan_state *state,
                              const mpcb_outputs *out, double *stage_cost, int32_t *stage_status,
                              int32_t *stage_nactive, void *stream);

/* 1-D box-constrained plan: solveQPconstraint for B instances (one axis each).
 * x0 [B][3], zmin/zmax [B][N] (BoxLimits, MPCPlanner.h:29-34).  Outputs: the *_x
 * members; active_idx < N is the lower bound of that knot, >= N the upper bound.
 * With a BOX_SLACK handle the same call is solveQPconstraintSlack: 3N variables (jerk and
 * one slack per bound), 4N rows (active_idx >= 2N: the sign constraint of slack
 * active_idx - 2N); the working set holds about 2N rows, so give active_stride >= 4N to
 * receive all of it; min_margin is the distance of the ZMP to the nearer bound.
 * height [B]: actual_height per instance, or NULL for mpcb_params.height (both handle kinds). */
int  mpcb_solve_box_device(mpcb_handle h, int32_t B, const double *x0, const double *zmin,
                           const double *zmax, const double *height, const mpcb_outputs *out, void *stream);
int  mpcb_solve_box(mpcb_handle h, int32_t B, const double *x0, const double *zmin,
                    const double *zmax, const double *height, const mpcb_outputs *out);

/* Unconstrained ZMP tracking: solveQP (MPCPlanner.cpp:223-237) for B instances.
 * x0 [B][3], zmp_ref [B][N].  Outputs: jerk_x, zmp_x, com_x, com_xd (status = OK, iters = 0,
 * n_active = 0 where asked for; no other output is written). */
int  mpcb_solve_unconstrained_device(mpcb_handle h, int32_t B, const double *x0, const double *zmp_ref,
                                     const mpcb_outputs *out, void *stream);
int  mpcb_solve_unconstrained(mpcb_handle h, int32_t B, const double *x0, const double *zmp_ref,
                              const mpcb_outputs *out);
/* solveQP takes `actual_height` with every call (MPCPlanner.cpp:223-226) and its Hessian
 * Zu'Zu + (R/Q) I depends on it: height [B] (HOST pointers throughout) gives every instance its
 * own; the instances are grouped on distinct heights as mpcb_solve_coupled does for COUPLED_REF
 * (tables of heights not seen before are built together on the GPU, at most 1024 are kept).
 * height == NULL: mpcb_solve_unconstrained. */
int  mpcb_solve_unconstrained_heights(mpcb_handle h, int32_t B, const double *x0, const double *zmp_ref,
                                      const double *height, const mpcb_outputs *out);

/* Diagnostic, no counterpart in the reference: how many instances of the handle's most recent
 * solve ended in the dense second-stage kernel's full method (working set beyond the capacity of
 * everything in front of it, or a numerically singular reduced system).  Waits for the device. */
int  mpcb_last_handover_count(mpcb_handle h);

/* Diagnostic, no counterpart in the reference: how many instances of the handle's most recent
 * coupled solve outgrew the working-set rows of the first-stage kernel (24; 32 with more than 256 knots) and were taken over by
 * its wide edition (60 rows, a quarter of the occupancy); those beyond 60 rows are counted by
 * mpcb_last_block_count as well.  Always 0 for box problems.  Waits for the device. */
int  mpcb_last_wide_count(mpcb_handle h);

/* Diagnostic, no counterpart in the reference: how many instances of the handle's most recent
 * coupled solve outgrew the wide edition too and went to the block edition of the same method
 * (one workgroup per QP, up to 160 rows; horizons up to 256).  What that could not finish either
 * is part of mpcb_last_handover_count.  Waits for the device. */
int  mpcb_last_block_count(mpcb_handle h);

/* Scheduling only, no counterpart in the reference (which solves one QP at a time): the order in
 * which the first-stage kernel starts the instances of a batch.  Results do not depend on it.
 * mode 0 = MPCB_DISPATCH_IN_ORDER (default): by instance index.  mode k = 1 .. MPCB_DISPATCH_MAX_LAG:
 * every solve records the running time of each instance, and a solve of the same batch size starts
 * the instances that took longest in the k-th solve before it first (index order while there is no
 * such record).  A QP's running time spans a factor of two to three within a batch, so with a few
 * instances per wave slot the launch otherwise ends waiting for long instances that started
 * late.  k = 1 (MPCB_DISPATCH_LONGEST_FIRST) is for callers that solve a slowly changing batch
 * again and again: 10-25 % at 8,192 instances; k = the number of legs in the gait cycle (4 for
 * the crawl) is for a loop that re-plans every rollout once per footstep, as
 * testPlannerXYconstraintMPC.cpp does -- consecutive replans then have unrelated running times
 * (another leg swings first), those a gait cycle apart correlate at 0.8-0.97.  Unrelated batches
 * gain nothing and lose nothing.  Costs (MPCB_DISPATCH_MAX_LAG + 1) B ints of device memory,
 * allocated by the next solve.  mpcb_replan_sweep_device uses the records of its own stages only
 * (stage s is ordered by stage s - k of the same call).  The _noalloc entry point always runs in
 * index order, and so do batches that bring caller-built rows (rows_pqr) or per-instance heights. */
enum { MPCB_DISPATCH_IN_ORDER = 0, MPCB_DISPATCH_LONGEST_FIRST = 1, MPCB_DISPATCH_MAX_LAG = 8 };
int  mpcb_set_dispatch_hint(mpcb_handle h, int32_t mode);

/* Diagnostic builds only (library compiled with -DMPCB_STAMPS): device buffer of 16 cycle
 * counters per instance that the kernels fill; a no-op pointer store in the normal build. */
int  mpcb_debug_set_stamp_buffer(mpcb_handle h, long long *dev);

/* Copy an instance-independent matrix to the host (tests, diagnostics).
 * which: 0 = S (ZMP response, N x N), 1 = W^T (jerk response), 2 = CoM position
 * response, 3 = CoM velocity response, 4 = Kz (N x 4), 5 = Kx (N x 4); the prediction
 * matrices of buildMatrix: 6 = first column of Zu (N), 7 = Zx (N x 3), 8 / 9 = the same for
 * Xpu / Xpx, 10 / 11 = for Xvu / Xvx; 12 = L^-T of the Hessian block (N x N, column-major);
 * 13 = f_unc matrix F (4 x 4); COUPLED_REF: 14 = Sp, 15 = Wp (nphase x N), 16 = F2
 * (4 x nphase), 17 = F3 (nphase x nphase); handles that take per-instance heights:
 * 18 = Spp, 19 = Sps, 20 = Saa, 21 = Wtp, 22 = Wta (N x N), 23 = Kzp, 24 = Kza (N x 4).
 * Returns the number of doubles written (dst == NULL: only that number) or a negative error. */
int  mpcb_get_matrix(mpcb_handle h, int32_t which, double *dst);

/* Diagnostic, needs no GPU: the same matrix from the host long-double construction (what a
 * handle created with MPCB_TABLES_HOST uploads), computed on the spot for `params`.  Ids as
 * mpcb_get_matrix (0-17; the height-free parts 18-24 are not offered); 25 = {trace(G1),
 * trace(L^-T)}.  The checker of the GPU construction; no counterpart in the reference.
 * Returns the number of doubles written (dst == NULL: only that number) or a negative error. */
int  mpcb_debug_host_table(const mpcb_params *params, int32_t which, double *dst);

#ifdef __cplusplus
}
#endif
#endif /* MPC_AMD_H_ */

// Device-side argument blocks and launchers shared by the C ABI and the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mpc_amd.h"

namespace mpcb {

// Internal per-instance status: the fast kernel could not certify the instance
// (working set larger than its LDS-resident factor, or the reduced system became
// numerically singular).  Such instances are re-solved by the dense kernel; the
// value never leaves the library.
constexpr int STATUS_UNRESOLVED = 3;
// Internal as well: the first edition of the wave-per-QP kernel ran out of rows for the working
// set and leaves the instance to its wide edition.
constexpr int STATUS_TO_WIDE = 4;
// ... and the wide edition ran out of rows in turn: the instance goes to the block edition (one
// workgroup per QP, zspace_block.h), which the dense kernel's workgroups run before the full method.
constexpr int STATUS_TO_BLOCK = 5;

struct DevSetup {
    int N, m, axes, form;
    int nvar;                         // variables of one QP (axes * N; 3N with slacks)
    double slack_ls, slack_g0;        // BOX_SLACK: sqrt(Qs) and the linear cost 1000 * Qs
    int steps, phase_knots, nphase;
    int iter_cap;
    const double *S, *Wt, *Sc, *Sv;   // N x N response matrices, row k = knot k
    const double *Kx, *Kz, *Kc, *Kv;  // N x 4 unconstrained responses
    // ZMP-reference formulation only (else NULL): responses to a unit reference on the
    // knots of one phase, [nphase][N], weight Qs folded in; cost terms F2 [4][nphase],
    // F3 [nphase][nphase]
    const double *Sp, *Wp, *F2, *F3;
    int nref;                         // segments of the ZMP reference: rows of Sp / Wp, columns of F2, order of
                                      // F3 -- the footstep phases (nphase) or, for a free reference, the knots (N)
    // per-instance heights (else NULL): S = Spp - c Sps + c^2 Saa, W^T = Wtp - c Wta,
    // Kz = Kzp - c Kza with c = height / gravity
    const double *Spp, *Sps, *Saa, *Wtp, *Wta, *Kzp, *Kza;
    double inv_gravity;
    const int *knot_phase, *knot_row0; // [N]
    double F[16];
    double psi_thr;                   // m * eps * c1 * c2 * 100 (eiquadprog.hpp:267)
    double traceG1;
    double Ts;                        // sample time of the prediction model
    double hip[8];
    long long *dbg;                   // cycle stamps (diagnostic builds only), else NULL
};

struct DevIn {
    const double *x0, *y0, *speed, *feet_x, *feet_y, *com0;
    const int *swing0;
    const double *zmin, *zmax;        // BOX_1D
    const double *rows_pqr;           // optional [B][m][3]: caller-built polygon rows
    const double *zref_phase;         // optional [B][nref][2]: ZMP reference per segment
    const double *height;             // optional [B]: CoM height per instance (needs the height tables)
    int zref_knot_mode;               // zref_phase holds [B][N][2], one reference per knot
    const double *phase_pqr;          // optional [B][nphase][4][3]: caller-built polygon of every footstep phase
};

// Hand-over from the wave-per-QP kernel to the dense kernel, and the dense kernel's
// tables and HBM workspace.
// Record the first stage leaves for an instance it hands over -- box problems: to the
// LDS-resident dense second stage; coupled forms: to the wide edition of the wave-per-QP kernel
// (else the pointers are NULL): the state at the top of the outer iteration it could not finish
// -- working set in order, multipliers, objective and the counters -- so that the next stage
// rebuilds its factors for that working set and carries on from there instead of repeating
// those iterations.  (Coupled forms pack (knot << 2 | slot) << 12 | row into the row entries.)
constexpr int WARM_QL = 32;       // rows a record can hold (>= the first stage's capacity)
constexpr int WARM_STRIDE = 36;   // per instance: [0, WARM_QL) rows resp. multipliers, then
constexpr int WARM_N = 32;        //   warm_i: number of rows, outer iterations, passes
constexpr int WARM_ITERS = 33;    //   warm_d: objective, operation count, R_norm
constexpr int WARM_PASSES = 34;
constexpr int WARM_F = 32, WARM_FALG = 33, WARM_RNORM = 34;

// Counter block of a solve (DenseWs::count points at its first entry).  The first stage counts its
// hand-overs in [CNT_HANDOVER] and [CNT_WIDE], its wide edition those to the block edition in
// [CNT_BLOCK]; the dense kernel, always the last launch of a solve, has its last workgroup out copy
// them to the CNT_LAST_* entries (which the host reads) and zero the block again, so that a solve
// finds the counters at zero without a memset of its own.  [CNT_LAST_HANDOVER] counts the instances
// the full method solved: its own list plus what the block edition passed on ([CNT_FELL]).
constexpr int CNT_HANDOVER = 0, CNT_WIDE = 1, CNT_DONE = 2, CNT_LAST_HANDOVER = 3, CNT_LAST_WIDE = 4;
constexpr int CNT_BLOCK = 5, CNT_LAST_BLOCK = 6, CNT_FELL = 7;
constexpr int COUNTER_INTS = 8;

struct DenseWs {
    int *list;                 // instance indices the first stage could not certify
    int *count;                // how many: the first of COUNTER_INTS ints, see CNT_* below
    int *host_count;           // page-locked host int that receives the hand-over count of the solve, or NULL
    int *wide_list;            // coupled forms: instances the first edition of the wave-per-QP kernel
    int *wide_count;           //   leaves to its wide edition (working set beyond its 24 or 32 rows), or NULL
    int *block_list;           // coupled forms: instances the wide edition leaves to the block edition (beyond
                               //   60 rows; counted in count[CNT_BLOCK]), or NULL
    int block_rows;            // rows the block edition may hold before it passes an instance on (its
                               //   capacity, or less: MPCB_DEBUG_BLOCK_ROWS, for tests of the fall-through)
    int *warm_i;               // [B][WARM_STRIDE] hand-over records (see above), or NULL
    double *warm_d;            // [B][WARM_STRIDE]
    int *order;                // dispatch order of the first stage: workgroup w solves instance order[w]
                               //   (a permutation of 0..B-1), or NULL for w itself
    int *order_record;         // [ORDER_HEADER + B] running-time record: the order is built from what the
                               //   previous solve left in it, then this solve overwrites it (NULL: none)
    double *J, *R;             // workspace: one n x n slot of each per resident workgroup
    const double *J1;          // L1^-T (N x N, column-major, upper triangular)
    const double *zu_col, *xp_col, *xv_col; // Toeplitz first columns of Zu, Xpu, Xvu
    const double *zx, *xpx, *xvx;           // N x 3 state matrices
    const int *row_knot, *row_slot;         // [m] (polygon rows only)
};

// Enqueues the structure-exploiting dual active-set kernel for B instances.  Instances it
// cannot certify are appended to ws.list (ws.count must be zero on entry).
hipError_t launch_zspace(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, int B,
                         const DenseWs &ws, hipStream_t stream);

// Builds ws.order from ws.order_record (identity unless that record is of a solve of B instances),
// stamps the record for the coming solve and clears the two counters at ws.count.
constexpr int ORDER_BINS = 256;
constexpr int ORDER_HEADER = 4; // ints before the per-instance bins; [0] = batch size of the recorded solve
hipError_t launch_dispatch_order(const DenseWs &ws, int B, hipStream_t stream);

// Enqueues the wide edition (60-row capacity, a quarter of the occupancy) over ws.wide_list with
// `blocks` single-wave workgroups; what outgrows it too joins ws.block_list (if there is one), what
// it cannot finish for another reason joins ws.list.  A no-op for box
// problems and when ws.wide_list is NULL.  LDS bytes one of its workgroups needs.
hipError_t launch_zspace_wide(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, int blocks,
                              const DenseWs &ws, hipStream_t stream);
size_t zspace_wide_lds_bytes(const DevSetup &sd);

// Fills ws.list with 0..B-1 and ws.count with B (dense-only mode).
hipError_t launch_list_all(const DenseWs &ws, int B, hipStream_t stream);

// Enqueues the dense kernel over ws.list with `blocks` workgroups (each owns one slot of
// ws.J / ws.R and loops over list entries blockIdx.x, blockIdx.x + blocks, ...).
hipError_t launch_dense(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, const DenseWs &ws,
                        int blocks, hipStream_t stream);

// The LDS-resident low-latency variant of the dense kernel for small box problems
// (box_dense.hip): whether it covers this shape, its LDS need, and its launcher (same
// contract as launch_dense).
bool box_dense_applies(const DevSetup &sd);
size_t box_dense_lds_bytes(const DevSetup &sd);
hipError_t launch_box_dense(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, const DenseWs &ws,
                            int blocks, hipStream_t stream);

// LDS bytes one workgroup of the dense kernel needs for this shape (dynamic + static part; with
// `with_block` also what the block edition, which its workgroups run first, needs);
// *in_lds: whether it keeps J and R in LDS, in which case ws.J / ws.R are never touched.
size_t dense_lds_bytes_needed(const DevSetup &sd, bool *in_lds, bool with_block = false);
// Whether the block edition covers this shape (coupled form, HBM-resident dense factors, the knots
// fit one workgroup).
bool block_edition_applies(const DevSetup &sd);

// MPCPlanner::solveQP for B instances (handle built with MPCB_UNCONSTRAINED).
hipError_t launch_unconstrained(const DevSetup &sd, const double *x0, const double *zref,
                                const mpcb_outputs &out, int B, hipStream_t stream);

// Receding-horizon sweep: mutable per-rollout state and the per-stage arguments of the
// advance kernel.
struct ReplanState {
    double *x0, *y0;            // [B][3]
    const double *speed;        // [B][2]
    double *feet_x, *feet_y;    // [B][4]
    int *swing0;                // [B]
    double *com0;               // [B][2]
    const double *disturbance;  // [B] or NULL
};

struct ReplanStep {
    int stage, window, N, disturbance_start, use_com_correction, feet_move;
    double Ts;
    double hip[8];
    const double *jerk_x, *jerk_y, *cost;
    const int *status, *n_active;
    double *stage_cost;
    int *stage_status, *stage_nactive;
};

hipError_t launch_replan_init(const ReplanState &st, int B, int use_com_correction, hipStream_t stream);
hipError_t launch_replan_advance(const ReplanState &st, const ReplanStep &sp, int B, hipStream_t stream);

} // namespace mpcb

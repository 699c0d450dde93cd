// Structure-exploiting dual active-set QP kernel: ONE WAVEFRONT PER QP.
//
// The reference solves  min 1/2 x'Gx + g0'x  s.t. CI'x + ci0 >= 0  with the dense
// Goldfarb-Idnani method (eiquadprog.hpp:130-447) after assembling G (n x n) and
// CI (n x m) densely (MPCPlanner.cpp:427-783).  Here the same method -- same pivot
// rules, tolerances and step formulas -- runs on the multipliers u of the working
// set A only, because for this planner
//   * G = blkdiag(G1, G1) is the same for every instance, and
//   * row j of CI' is a half-plane on the ZMP at ONE knot k_j:
//         s_j = p_j zx[k_j] + q_j zy[k_j] + r_j,   zx = Zx x0 + Zu jerk_x.
// With W = G1^-1 Zu' and S = Zu W (instance-independent, read through L2):
//         zmp = zmp_unc + sum_a (u_a p_a) S[k_a, :]
//         n_i' G^-1 n_j = (p_i p_j + q_i q_j) S[k_i, k_j]
// so a solve needs only the q x q triangular factor R of the working set
// (R'R = N_A' G^-1 N_A, which is also the R of the reference's QR update) and its
// inverse.  Both live in LDS; nothing n x n or m x n is ever formed.
//
// Layout: one 64-thread workgroup (a single wave) per QP, no workgroup barrier anywhere.
// Lane l owns knots l, l+64, ... and the (at most 4) constraint rows of each of
// them, in registers.  Cross-lane data goes through the wave's private LDS slice
// or through wave shuffles.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "device_common.h"
#include "kernels.h"

namespace mpcb {
namespace {

// Capacity of the LDS-resident factor: 32 rows (24 in the slim configuration, WaveConfig).  The coupled forms keep no R (see
// drop_constraint), the box form does (drop_constraint_rotations).  (Three waves per SIMD for the
// coupled forms -- 28 rows, 13 KB of LDS per wave, 168 registers -- were tried: the compiler does
// not get this kernel below ~95 spilled registers at that budget, and with spills it runs at two
// thirds of the speed of the two-wave build.)
// Box problems hand over earlier than their storage would force them to.  What fills the
// factor there are the numerically infeasible draws, which run to ~100 active rows whatever the
// capacity; the dense second stage (LDS-resident, four waves on one instance, built for the
// latency of one hard instance) takes over where this kernel stopped, so nothing is repeated,
// and one of its passes takes about half as long as one of this kernel's at 20-odd rows.  Against
// that, every instance handed over occupies a compute unit of its own.  Measured on the
// 4,096-pair batch, ms per step: 20 -> 2.07, 24 -> 2.10, 26 -> 3.15, 28 -> 2.76 -- the step is as
// long as its longest draw, and which draw that is changes with the point of the hand-over (the
// two outliers of ~290 outer iterations at 28 rows do not occur at 20 and 24, where the longest
// takes ~210); 24 keeps regular instances of up to 24 rows in this kernel.
constexpr int QL_BOX = 24;

template <int QL_, bool KEEP_R>
struct WaveStateT {
    static constexpr int QL = QL_, LDI = QL_ + 1, TRI = QL_ * (QL_ + 1) / 2;
    double R[KEEP_R ? TRI : 1]; // packed upper triangle, column-major
    // R^-1 as a full square with leading dimension LDI (odd: a column read by rows is
    // conflict-free).  Everything outside the leading iq x iq upper triangle is kept zero, so
    // the products with it need no masks.
    double Rinv[QL * LDI];
    double wsU[QL + 1], wsP[QL + 1], wsQ[QL + 1];       // working set: multiplier, p, q
    double oldU[QL + 1];                                // copy taken at step 1 (eiquadprog.hpp:275-277)
    double cx[QL + 4], cy[QL + 4];                      // row coefficients, zero-padded to a multiple of 4
    double topF, topFalg;                               // scalars as of the top of the current outer
    int topPasses, topIq;                               //   iteration (for the hand-over record)
    int wsRow[QL + 1], wsKS[QL + 4];                    // constraint row; (knot << 2) | slot
    int oldRow[QL + 1], oldKS[QL + 1];
    int pad; // (4 QL + 9 ints above: an odd number)
};
// WIDE: the second edition of this kernel, which takes over the instances whose working set
// outgrows the first edition's rows (coupled forms): 60 rows -- with the four padding entries that is one lane
// per working-set position -- at a quarter of the occupancy (38 KB of LDS per wave).
// SLIM: the first edition of the coupled forms at horizons up to 256 knots runs THREE waves per
// SIMD: 24 rows (10.8 KB of LDS per wave at 41 phases, twelve waves per compute unit) and a budget
// of 168 registers, which the compiler meets with a few spilled dwords that only rare paths reload.
// The third wave hides more of the per-iteration trips to L2 and LDS than the eight rows are worth:
// working sets beyond 24 rows are rare at the benchmark's weights and are taken over mid-solve by
// the wide edition.  (With a height per instance three table parts are read per response entry:
// those variants keep two rows' loads in flight per group instead of four to stay in the budget.)
// The box form hands over at 24 rows (QL_BOX), so at horizons up to 256 knots its state is cut to
// 24 rows as well: 9 KB of LDS per wave where 32 rows take 14.5, so a compute unit holds twelve of
// its waves -- the limit of the kernel's 161 registers -- instead of eleven.  (Beyond 256 knots the factor
// storage doubles as the buffer of the jerk exchange and keeps its 32 rows.)
template <bool POLY, int NPL, bool WIDE, bool HV>
struct WaveConfig {
    static constexpr bool SLIM = POLY && !WIDE && NPL <= 4;
    static constexpr int QL = WIDE ? 60 : (SLIM || (!POLY && NPL <= 4)) ? 24 : 32;
    typedef WaveStateT<QL, !POLY> State;
    static_assert(POLY || QL <= WARM_QL, "a hand-over record holds the whole working set");
    static_assert(QL + 4 <= 64 && QL % 4 == 0, "one lane per working-set position; products run four terms per trip");
    static_assert(sizeof(State) % 8 == 0, "tab must stay 8-byte aligned");
};

__device__ __forceinline__ int tri(int col) { return col * (col + 1) / 2; }

// value of v on lane `src` (wave-uniform), without going through LDS
__device__ __forceinline__ double lane_bcast(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
}

// Remove working-set position qq (delete_constraint, eiquadprog.hpp:516-592), the reference's way:
// R loses column qq and is re-triangularised by Givens rotations of rows (j, j+1);
// the same rotations applied to the columns of R^-1, minus row qq and the last
// column, give the new inverse -- what the reference does to the columns of J.
// (Kept for the box problems.  Its rotation angles come from R, whose update is backward stable,
// so T picks up only rounding-level errors per drop; the closed form below takes the angles from
// T itself, whose errors then compound when the working set is nearly dependent.  Every regular
// instance is solved identically by both, but the numerically infeasible box draws lose the
// reduced system's accuracy after a handful of rows with the closed form and are handed to the
// second stage almost from scratch -- 3.9 instead of 2.6 ms per decoupled step.)
template <class WaveState>
__device__ __forceinline__ void drop_constraint_rotations(WaveState &W, int lane, int &iq, int qq) {
    constexpr int QL = WaveState::QL, LDI = WaveState::LDI;
    // working-set arrays: positions qq+1 .. iq-1 move down by one
    double tu = 0, tp = 0, tq = 0;
    int trow = 0, tks = 0;
    const bool mv = lane > qq && lane < iq;
    if (mv) {
        tu = W.wsU[lane];
        tp = W.wsP[lane];
        tq = W.wsQ[lane];
        trow = W.wsRow[lane];
        tks = W.wsKS[lane];
    }
    wsync();
    if (mv) {
        W.wsU[lane - 1] = tu;
        W.wsP[lane - 1] = tp;
        W.wsQ[lane - 1] = tq;
        W.wsRow[lane - 1] = trow;
        W.wsKS[lane - 1] = tks;
    }
    const int last = iq - 2; // new column / rotation index runs qq .. iq-2
    const bool ri = lane < iq;
    if (qq <= last) {
        // R: lane k builds new column k from old column k+1
        const bool rk = lane >= qq && lane <= last;
        const int lq = lane < QL - 1 ? lane : QL - 2;
        const int offn = tri(lq), offo = tri(lq + 1);
        // rows above qq are not touched by the rotations: plain copy, eight rows per group
        // (all loads of a group are issued before its stores, the wave runs in lockstep)
        for (int i0 = 0; i0 < qq; i0 += 8) {
            double v[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; q8++) v[q8] = (rk && i0 + q8 < qq) ? W.R[offo + i0 + q8] : 0.0;
            wsync();
#pragma unroll
            for (int q8 = 0; q8 < 8; q8++)
                if (rk && i0 + q8 < qq) W.R[offn + i0 + q8] = v[q8];
        }
        double top = rk ? W.R[offo + qq] : 0.0;
        // R^-1: lane i carries row i; rows below qq move up by one
        const int li = lane < QL ? lane : QL - 1;
        double tcur = (ri && lane <= qq) ? W.Rinv[li * LDI + qq] : 0.0;
        const int inew = lane < qq ? lane : lane - 1;
        wsync();
        for (int j = qq; j <= last; j++) {
            const bool act = rk && lane >= j;
            const double bot = act ? W.R[offo + j + 1] : 0.0;
            const double t2 = (ri && lane <= j + 1) ? W.Rinv[li * LDI + j + 1] : 0.0;
            const double tj = lane_bcast(top, j), bj = lane_bcast(bot, j);
            // h = hypot(tj, bj) (distance(), eiquadprog.hpp:90-108) and 1/h from one chain
            const double hh = tj * tj + bj * bj;
            double h = 0.0, ih = 0.0;
            if (hh > 0.0) sqrt_rsqrt(hh, h, ih);
            wsync();
            if (h == 0.0) {
                if (act) {
                    W.R[offn + j] = top;
                    top = bot;
                }
                if (ri && lane != qq && inew <= j) W.Rinv[inew * LDI + j] = tcur;
                tcur = t2;
                continue;
            }
            double cc = tj * ih, ss = bj * ih, hd = h;
            if (cc < 0.0) {
                cc = -cc;
                ss = -ss;
                hd = -h;
            }
            if (act) {
                if (lane == j) {
                    W.R[offn + j] = hd;
                } else {
                    W.R[offn + j] = cc * top + ss * bot;
                    top = ss * top - cc * bot;
                }
            }
            if (ri && lane != qq && inew <= j) W.Rinv[inew * LDI + j] = cc * tcur + ss * t2;
            tcur = ss * tcur - cc * t2;
        }
    }
    wsync();
    if (ri) W.Rinv[lane * LDI + iq - 1] = 0.0; // the vacated last column
    iq--;
    wsync();
}

// Remove working-set position qq (delete_constraint, eiquadprog.hpp:516-592).
// The reference deletes column qq of R and restores the triangle with plane rotations of the rows
// (j, j+1), j = qq .. iq-2, applying the same rotations to the columns of J.  Here only T = R^-1
// is kept, and those rotations are the ones that move row qq of T -- entries g_j = T(qq, j),
// j = qq .. iq-1 -- into its last entry.  With the prefix norms rho_j = ||g[qq .. j]|| and the
// prefix sums Q_j = sum_{c = qq .. j} g_c col_c of the old columns, the whole sequence has the
// closed form
//     new col_j = g_{j+1} / (rho_{j+1} rho_j) Q_j - (rho_j / rho_{j+1}) col_{j+1},   qq <= j < iq-1
// (up to the signs of columns, which cancel in every product the method forms); row qq and the
// last column drop out.  All rotation coefficients come from one prefix scan and one reciprocal
// square root per lane; what remains sequential is a chain of two fused multiply-adds per
// column in every row.
template <class WaveState>
__device__ __forceinline__ void drop_constraint(WaveState &W, int lane, int &iq, int qq) {
    constexpr int QL = WaveState::QL, LDI = WaveState::LDI;
    // working-set arrays: positions qq+1 .. iq-1 move down by one
    double tu = 0, tp = 0, tq = 0;
    int trow = 0, tks = 0;
    const bool mv = lane > qq && lane < iq;
    if (mv) {
        tu = W.wsU[lane];
        tp = W.wsP[lane];
        tq = W.wsQ[lane];
        trow = W.wsRow[lane];
        tks = W.wsKS[lane];
    }
    wsync();
    if (mv) {
        W.wsU[lane - 1] = tu;
        W.wsP[lane - 1] = tp;
        W.wsQ[lane - 1] = tq;
        W.wsRow[lane - 1] = trow;
        W.wsKS[lane - 1] = tks;
    }
    const int last = iq - 1; // columns qq .. last take part, column last goes
    const bool ri = lane < iq;
    const int li = lane < QL ? lane : QL - 1;
    if (qq < last) {
        // lane j: g_j, rho_j and the coefficients of column j
        const bool gj = lane >= qq && lane <= last;
        const double g = gj ? W.Rinv[qq * LDI + li] : 0.0;
        const double p = wave_scan(g * g);
        double rho = 0.0, irho = 0.0;
        sqrt_rsqrt(p > 0.0 ? p : 1.0, rho, irho);
        rho = p > 0.0 ? rho : 0.0;
        irho = p > 0.0 ? irho : 0.0;
        const double gn = dpp_f64<0x130, 0xF>(0.0, g), irn = dpp_f64<0x130, 0xF>(0.0, irho); // of lane + 1
        const bool cj = lane >= qq && lane < last;
        const double a = cj ? gn * (irn * irho) : 0.0, b = cj ? -rho * irn : 0.0;
        // lane i: row i of T; rows below qq move up by one
        const int inew = lane < qq ? lane : lane - 1;
        const bool wr = ri && lane != qq;
        double cur = ri ? W.Rinv[li * LDI + qq] : 0.0, q = 0.0;
        for (int j = qq; j < last; j++) {
            const double nxt = ri ? W.Rinv[li * LDI + j + 1] : 0.0;
            q = fma(lane_bcast(g, j), cur, q);
            const double o = fma(lane_bcast(a, j), q, lane_bcast(b, j) * nxt);
            if (wr) W.Rinv[inew * LDI + j] = o;
            cur = nxt;
            wsync(); // column j + 1 of every row is read before anything is stored over it
        }
    }
    wsync();
    if (ri) W.Rinv[lane * LDI + last] = 0.0; // the vacated last column
    iq--;
    wsync();
}

// Diagnostic build only (-DMPCB_STAMPS): cycles per section, summed per instance into
// sd.dbg[inst*16 + section].  No stamp executes in the normal build.
#ifdef MPCB_STAMPS
#define STAMP(i)                                         \
    do {                                                 \
        const long long t_ = __builtin_amdgcn_s_memtime(); \
        stamp_acc[i] += t_ - stamp_t;                    \
        stamp_t = t_;                                    \
    } while (0)
#else
#define STAMP(i) \
    do {         \
    } while (0)
#endif

// ROWS: polygon rows come from the caller (in.rows_pqr, the (p, q, r) of every row of the
// A, b that computeSteps returned, possibly edited) instead of being built from the feet.
// HV: every instance has its own CoM height (in.height): the response tables are combined from
// their height-free parts as they are read, S = Spp - c Sps + c^2 Saa, W^T = Wtp - c Wta,
// Kz = Kzp - c Kza with c = height / gravity (three loads per entry instead of one).
// One QP, solved by the calling wave in its LDS slice `smem` (see the head of the file).
// Instances this edition cannot finish are appended to wide_list (capacity reached and a wide
// edition follows) or to unresolved_list (for the dense kernel).  RECORD: how long the instance
// took is noted in order_record (see dispatch_order_kernel).
// The three argument blocks every kernel of this file starts with, in parameter order: where the
// outputs block sits in the kernel-argument segment.
struct LeadingArgs {
    DevSetup sd;
    DevIn in;
    mpcb_outputs out;
};

template <int NPL, bool POLY, bool ROWS, bool HV, bool WIDE, bool RECORD>
__device__ __forceinline__ void solve_instance(const DevSetup &sd, const DevIn &in, const mpcb_outputs &,
                                               const int inst, const int lane, double *smem, int *unresolved_list,
                                               int *unresolved_count, int *warm_i, double *warm_d, int *wide_list,
                                               int *wide_count, int *order_record) {
    constexpr int ES = POLY ? 4 : 2;
    unsigned t_start = 0;
    if constexpr (RECORD) t_start = (unsigned)__builtin_amdgcn_s_memtime();
    typedef typename WaveConfig<POLY, NPL, WIDE, HV>::State WaveState;
    constexpr bool SLIM = WaveConfig<POLY, NPL, WIDE, HV>::SLIM;
    constexpr int QL = WaveState::QL, LDI = WaveState::LDI;
    const int N = sd.N;
#ifdef MPCB_STAMPS
    long long stamp_acc[16] = {0};
    long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
    // polygon table (one padding phase after the last) + the ZMP reference per segment
    const int tab_doubles = POLY ? (sd.nphase + 1) * 12 + 2 * max(sd.nphase, sd.nref) : 0;
    WaveState &W = *reinterpret_cast<WaveState *>(smem);
    double *tab = smem + sizeof(WaveState) / 8; // [phase][edge][p,q,r]
    double *cref = tab + (POLY ? (sd.nphase + 1) * 12 : 0);              // [phase][x,y] ZMP reference

    // ---- owned knots ------------------------------------------------------
    // Lane l owns knots l + 64 c.  Only what the main loop reads stays in registers for the
    // whole solve (kcb, the packed phase/row below); the knot numbers themselves are
    // recomputed where the prologue, the rare paths and the epilogue need them.
    int lane_k = lane; // (laundered after the main loop: nothing derived from it stays live across it)
    auto KN = [&](int c) -> int { return lane_k + 64 * c; };
    auto KC = [&](int c) -> int { return KN(c) < N ? KN(c) : N - 1; };
    unsigned kcb[NPL];   // byte offset of the (clamped) knot inside a table row
    unsigned validm = 0; // bit (c*4+e): the slot exists
#pragma unroll
    for (int c = 0; c < NPL; c++) kcb[c] = (unsigned)KC(c) * 8u;
    // phase and first row of the owned knots: asked for here, used after the polygons are built
    // (from clamped -- always valid -- knots: conditional loads would be issued and waited for one
    // chunk after the other)
    int ph_k[NPL], row0_k[NPL];
    if constexpr (POLY) {
#pragma unroll
        for (int c = 0; c < NPL; c++) {
            ph_k[c] = sd.knot_phase[KC(c)];
            row0_k[c] = sd.knot_row0[KC(c)];
        }
    }
    // Entries of the N x N response tables are fetched by buffer loads: the table is a buffer
    // resource (four scalar registers), the wave-uniform part of the offset (a row, or a column)
    // rides in a scalar register and the lane adds a 32-bit byte offset -- no per-lane 64-bit
    // address arithmetic.  (An offset past the table would read 0; none is generated.)
    const unsigned tab_bytes = (unsigned)N * (unsigned)N * 8u;
    auto table = [&](const double *t) -> __amdgpu_buffer_rsrc_t {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(t), 0, tab_bytes, 0x00020000);
    };
    auto tab_ld = [](__amdgpu_buffer_rsrc_t t, unsigned uniform_bytes, unsigned lane_bytes) -> double {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(t, lane_bytes, uniform_bytes, 0);
        return __hiloint2double((int)v.y, (int)v.x);
    };
    const unsigned rowB = (unsigned)N * 8u; // bytes of a table row
    const __amdgpu_buffer_rsrc_t tS = table(HV ? sd.Spp : sd.S), tW = table(HV ? sd.Wtp : sd.Wt);
    const __amdgpu_buffer_rsrc_t tS1 = table(sd.Sps), tS2 = table(sd.Saa), tW1 = table(sd.Wta); // HV only

    // ---- instance inputs ---------------------------------------------------
    // Every scalar input of the instance is fetched by ONE vector load (a lane per value) and
    // handed round by readlane: as wave-uniform scalar loads they were issued one array after
    // the other, each a full miss to HBM at the head of the kernel.
    double wx[4], wy[4];
    double feetx[4] = {0, 0, 0, 0}, feety[4] = {0, 0, 0, 0}, com0x = 0.0, com0y = 0.0;
    int swing_first = 0;
    {
        const double *src = nullptr;
        if (lane < 3) src = in.x0 + (size_t)inst * 3 + lane;
        if constexpr (POLY) {
            if (lane >= 3 && lane < 6) src = in.y0 + (size_t)inst * 3 + (lane - 3);
            if (lane >= 6 && lane < 8) src = in.speed + (size_t)inst * 2 + (lane - 6);
            if constexpr (!ROWS) {
                if (!in.phase_pqr) { // (the caller's polygons replace the footstep inputs)
                    if (lane >= 8 && lane < 12) src = in.feet_x + (size_t)inst * 4 + (lane - 8);
                    if (lane >= 12 && lane < 16) src = in.feet_y + (size_t)inst * 4 + (lane - 12);
                    if (lane >= 16 && lane < 18 && in.com0) src = in.com0 + (size_t)inst * 2 + (lane - 16);
                }
            }
        }
        int sw = 0;
        if constexpr (POLY && !ROWS) {
            if (lane == 18 && !in.phase_pqr) sw = in.swing0[inst];
        }
        const double v = src ? *src : 0.0;
        wx[0] = lane_bcast(v, 0);
        wx[1] = lane_bcast(v, 1);
        wx[2] = lane_bcast(v, 2);
        wx[3] = 0.0;
        wy[0] = wy[1] = wy[2] = wy[3] = 0.0;
        if constexpr (POLY) {
            wy[0] = lane_bcast(v, 3);
            wy[1] = lane_bcast(v, 4);
            wy[2] = lane_bcast(v, 5);
            wx[3] = lane_bcast(v, 6);
            wy[3] = lane_bcast(v, 7);
            if constexpr (!ROWS) {
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    feetx[l] = lane_bcast(v, 8 + l);
                    feety[l] = lane_bcast(v, 12 + l);
                }
                com0x = lane_bcast(v, 16);
                com0y = lane_bcast(v, 17);
                swing_first = __builtin_amdgcn_readlane(sw, 18);
            }
        }
    }

    double ch = 0.0, ch2 = 0.0; // HV: c and c^2
    if constexpr (HV) {
        ch = ufirst(in.height[inst] * sd.inv_gravity);
        ch2 = ufirst(ch * ch);
    }
    // entries of the response tables at a flat offset (row * N + column)
    // S[row][col] with a wave-uniform col and a per-lane row, and W^T[row][.] at the lane's
    // byte offset with a wave-uniform row: uniform part in scalar registers, 32-bit lane part
    auto S_col = [&](int row, int col) -> double {
        const unsigned bo = (unsigned)row * rowB, uo = (unsigned)col * 8u;
        if constexpr (HV) return fma(ch2, tab_ld(tS2, uo, bo), fma(-ch, tab_ld(tS1, uo, bo), tab_ld(tS, uo, bo)));
        else return tab_ld(tS, uo, bo);
    };
    // S[a][b] for a per-lane knot a and a wave-uniform knot b, read as S[b][a]: the table is stored
    // exactly symmetric, and this way the lanes' reads fall into ONE row (1.6 KB at N = 200, the
    // row the next outer iteration sweeps anyway if the candidate is added) instead of one cache
    // line in each of up to iq rows
    auto S_pair = [&](int lane_knot, int uniform_knot) -> double {
        const unsigned uo = (unsigned)uniform_knot * rowB, bo = (unsigned)lane_knot * 8u;
        if constexpr (HV) return fma(ch2, tab_ld(tS2, uo, bo), fma(-ch, tab_ld(tS1, uo, bo), tab_ld(tS, uo, bo)));
        else return tab_ld(tS, uo, bo);
    };
    auto Wt_row = [&](int row, unsigned byte_off) -> double {
        const unsigned ro = (unsigned)row * rowB;
        if constexpr (HV) return fma(-ch, tab_ld(tW1, ro, byte_off), tab_ld(tW, ro, byte_off));
        else return tab_ld(tW, ro, byte_off);
    };

    // ---- constraint rows ---------------------------------------------------
    // phr[c]: phase (10 bits; < 1024) | first row << 10 of knot c; tabo[c]: where the phase's
    // (p, q, r) triples start in tab
    int phr[NPL], tabo[NPL];
    double lo[NPL], hi[NPL];
#pragma unroll
    for (int c = 0; c < NPL; c++) {
        phr[c] = KN(c) << 10;
        tabo[c] = 0;
        lo[c] = hi[c] = 0.0;
    }
    auto PH = [&](int c) -> int { return phr[c] & 1023; };
    auto ROW0 = [&](int c) -> int { return phr[c] >> 10; };
    if constexpr (POLY) {
        // computeSteps + buildPolygonMatrix: lane l builds the polygon of phase l
        if constexpr (!ROWS) {
            if (in.phase_pqr) {
                // the caller's polygon of every phase (buildPolygonMatrix run by the caller), copied as
                // it stands: 12 doubles per phase, consecutive lanes on consecutive addresses
                const double *src = in.phase_pqr + (size_t)inst * sd.nphase * 12;
                for (int i = lane; i < sd.nphase * 12; i += 64) tab[i] = src[i];
            } else {
                for (int pbase = 0; pbase < sd.nphase; pbase += 64)
                    polygon_phase(sd, feetx, feety, com0x, com0y, swing_first, wx[3], wy[3], pbase + lane, tab, cref);
            }
        }
        if (in.zref_phase)
            for (int i = lane; i < 2 * sd.nref; i += 64) cref[i] = in.zref_phase[(size_t)inst * 2 * sd.nref + i];
        if constexpr (!ROWS) {
            // Rows that do not exist evaluate to +inf without a test: the fourth edge of a
            // three-foot phase, and every edge of the padding phase that the lanes past
            // the horizon point at, read (p, q, r) = (0, 0, +inf).
            for (int pb = lane; pb <= sd.nphase; pb += 64) {
                const bool three = (pb & 1) && pb < 2 * sd.steps;
                for (int e = (pb == sd.nphase ? 0 : three ? 3 : 4); e < 4; e++) {
                    tab[(pb * 4 + e) * 3] = 0.0;
                    tab[(pb * 4 + e) * 3 + 1] = 0.0;
                    tab[(pb * 4 + e) * 3 + 2] = DINF;
                }
            }
        }
        int ph[NPL], row0[NPL];
#pragma unroll
        for (int c = 0; c < NPL; c++) {
            ph[c] = ph_k[c];
            row0[c] = row0_k[c];
        }
#pragma unroll
        for (int c = 0; c < NPL; c++) {
            if (KN(c) < N) {
                const int E = ((ph[c] & 1) && ph[c] < 2 * sd.steps) ? 3 : 4;
                validm |= ((1u << E) - 1u) << (4 * c);
            } else if (!ROWS) {
                ph[c] = sd.nphase;
            }
            phr[c] = ph[c] | (row0[c] << 10);
            tabo[c] = ph[c] * 12;
        }
    } else {
#pragma unroll
        for (int c = 0; c < NPL; c++) {
            if (KN(c) < N) {
                lo[c] = in.zmin[(size_t)inst * N + KN(c)];
                hi[c] = in.zmax[(size_t)inst * N + KN(c)];
                validm |= 3u << (4 * c);
            } else { // rows past the horizon evaluate to +inf
                lo[c] = -DINF;
                hi[c] = DINF;
            }
        }
    }
    STAMP(11);
    if (lane < QL + 4) W.wsKS[lane] = 0; // padding entries must name a valid knot
    for (int i = lane; i < QL * LDI; i += 64) W.Rinv[i] = 0.0;
    wsync();

    // v += sum_a coef_a * M[knot_a, :] over the first n working-set rows.  The loads of a
    // whole group of rows (eight while more than four remain, else four) are in flight
    // together; cx/cy are zero beyond n, so the padding rows add 0.
    // (wtag: 0 = the ZMP response S, 1 = the jerk response W^T)
    auto accum_group = [&](auto wtag, int a0, int ksv, double *vx, double *vy, auto gtag) {
        constexpr int WH = decltype(wtag)::value;
        constexpr int G = decltype(gtag)::value;
        const __amdgpu_buffer_rsrc_t T0 = WH == 0 ? tS : tW;
        unsigned off[G];
        double ca[G], cb[G], sv[G][NPL];
#pragma unroll
        for (int j = 0; j < G; j++) {
            off[j] = (unsigned)(__builtin_amdgcn_readlane(ksv, a0 + j) >> 2) * rowB;
            ca[j] = W.cx[a0 + j];
            cb[j] = W.cy[a0 + j];
        }
#pragma unroll
        for (int j = 0; j < G; j++)
#pragma unroll
            for (int c = 0; c < NPL; c++) sv[j][c] = tab_ld(T0, off[j], kcb[c]);
        if constexpr (HV) {
            const __amdgpu_buffer_rsrc_t T1 = WH == 0 ? tS1 : tW1;
            double s1[G][NPL];
#pragma unroll
            for (int j = 0; j < G; j++)
#pragma unroll
                for (int c = 0; c < NPL; c++) s1[j][c] = tab_ld(T1, off[j], kcb[c]);
#pragma unroll
            for (int j = 0; j < G; j++)
#pragma unroll
                for (int c = 0; c < NPL; c++) sv[j][c] = fma(-ch, s1[j][c], sv[j][c]);
            if constexpr (WH == 0) {
#pragma unroll
                for (int j = 0; j < G; j++)
#pragma unroll
                    for (int c = 0; c < NPL; c++) s1[j][c] = tab_ld(tS2, off[j], kcb[c]);
#pragma unroll
                for (int j = 0; j < G; j++)
#pragma unroll
                    for (int c = 0; c < NPL; c++) sv[j][c] = fma(ch2, s1[j][c], sv[j][c]);
            }
        }
#pragma unroll
        for (int j = 0; j < G; j++)
#pragma unroll
            for (int c = 0; c < NPL; c++) {
                vx[c] += ca[j] * sv[j][c];
                if constexpr (POLY) vy[c] += cb[j] * sv[j][c];
            }
    };
    // (The knots of all positions come from ONE LDS read, a lane per position, and are handed
    // to the scalar unit by readlane: position by position each read was a round trip of its own.)
    auto ws_knots = [&]() -> int { return W.wsKS[lane < QL + 4 ? lane : 0]; };
    auto accum_rows = [&](auto wtag, int n, int ksv, double *vx, double *vy) {
        int a0 = 0;
        if constexpr (NPL <= 4 && !HV && !SLIM) { // (eight rows' loads do not fit the slim register budget)
            for (; n - a0 > 4; a0 += 8) accum_group(wtag, a0, ksv, vx, vy, std::integral_constant<int, 8>{});
        }
        if constexpr (SLIM && HV) { // (three table parts per entry: two rows per group at this register budget)
            for (; a0 < n; a0 += 2) accum_group(wtag, a0, ksv, vx, vy, std::integral_constant<int, 2>{});
        } else {
            for (; a0 < n; a0 += 4) accum_group(wtag, a0, ksv, vx, vy, std::integral_constant<int, 4>{});
        }
    };
    constexpr std::integral_constant<int, 0> TAB_S{};
    constexpr std::integral_constant<int, 1> TAB_W{};
    // v += sum over phases of (reference of the phase) * Mp[phase, :], eight phases' loads in
    // flight at a time (ZMP-reference formulation)
    auto accum_phases = [&](const double *Mp, double *vx, double *vy) {
        const int P = sd.nref;
        for (int p0 = 0; p0 < P; p0 += 8) {
            double cxr[8], cyr[8], mv[8][NPL];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int p2 = p0 + j < P ? p0 + j : P - 1;
                cxr[j] = p0 + j < P ? cref[2 * p2] : 0.0;
                cyr[j] = p0 + j < P ? cref[2 * p2 + 1] : 0.0;
#pragma unroll
                for (int c = 0; c < NPL; c++) mv[j][c] = Mp[(size_t)p2 * N + KC(c)];
            }
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int c = 0; c < NPL; c++) {
                    vx[c] += cxr[j] * mv[j][c];
                    vy[c] += cyr[j] * mv[j][c];
                }
        }
    };
    // coefficients of the working-set rows: coef * (p, q), zero-padded
    auto set_coefs = [&](double coef, int n) {
        if (lane < QL + 4) {
            const bool in = lane < n;
            W.cx[lane] = in ? coef * W.wsP[lane < QL ? lane : QL] : 0.0;
            W.cy[lane] = in ? coef * W.wsQ[lane < QL ? lane : QL] : 0.0;
        }
        wsync();
    };

    // caller-built rows: row0[c] + e may run past the knot's last row for a slot that does
    // not exist; clamp so the (masked) read stays inside the instance's rows
    const double *rows = ROWS ? in.rows_pqr + (size_t)inst * sd.m * 3 : nullptr;
    auto rowIdx = [&](int c, int e) -> int {
        // (recomputed at every use: the twelve sums would otherwise stay in registers throughout)
        int packed = phr[c];
        asm volatile("" : "+v"(packed));
        const int r = (packed >> 10) + e;
        return r < sd.m ? r : sd.m - 1;
    };
    auto slotP = [&](int c, int e) -> double {
        if constexpr (ROWS) return rows[rowIdx(c, e) * 3];
        else if constexpr (POLY) return tab[tabo[c] + e * 3];
        else return e == 0 ? 1.0 : -1.0;
    };
    auto slotQ = [&](int c, int e) -> double {
        if constexpr (ROWS) return rows[rowIdx(c, e) * 3 + 1];
        else if constexpr (POLY) return tab[tabo[c] + e * 3 + 1];
        else return 0.0;
    };
    auto slotR = [&](int c, int e) -> double {
        if constexpr (ROWS) return rows[rowIdx(c, e) * 3 + 2];
        else if constexpr (POLY) return tab[tabo[c] + e * 3 + 2];
        else return e == 0 ? -lo[c] : hi[c];
    };

    STAMP(15);
    // ---- unconstrained solution (x = -G^-1 g0, eiquadprog.hpp:188-191) ----
    double zxu[NPL], zyu[NPL];
#pragma unroll
    for (int c = 0; c < NPL; c++) {
        double kz[4];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            if constexpr (HV) kz[a] = fma(-ch, sd.Kza[(size_t)KC(c) * 4 + a], sd.Kzp[(size_t)KC(c) * 4 + a]);
            else kz[a] = sd.Kz[(size_t)KC(c) * 4 + a];
        }
        zxu[c] = kz[0] * wx[0] + kz[1] * wx[1] + kz[2] * wx[2] + kz[3] * wx[3];
        zyu[c] = kz[0] * wy[0] + kz[1] * wy[1] + kz[2] * wy[2] + kz[3] * wy[3];
    }
    double f;
    {
        double fp = 0.0;
        for (int pb = 0; pb < (sd.F2 ? sd.nref : 1); pb += 64)
            fp += unconstrained_cost_part(sd, wx, cref, 0, pb + lane, pb + lane == 0) +
                  unconstrained_cost_part(sd, wy, cref, 1, pb + lane, pb + lane == 0);
        f = wave_sum(fp);
    }
    if constexpr (POLY) {
        if (sd.Sp) accum_phases(sd.Sp, zxu, zyu); // ZMP-reference formulation: response to the per-phase reference
    }

    // Stopping threshold m * eps * c1 * c2 * 100 (eiquadprog.hpp:267): m is the number of rows
    // of the problem.  Caller-built rows may leave slots empty ((0, 0, +inf)); those are not
    // rows of the caller's A and do not count.
    double psi_thr = sd.psi_thr;
    if constexpr (ROWS) {
        int mine = 0;
#pragma unroll
        for (int c = 0; c < NPL; c++)
#pragma unroll
            for (int e = 0; e < ES; e++)
                if ((validm & (1u << (4 * c + e))) && slotR(c, e) < DINF) mine++;
        const int m_eff = (int)wave_sum((double)mine);
        psi_thr = ufirst(sd.psi_thr * ((double)m_eff / (double)sd.m));
    }

    // ---- dual active-set loop ----------------------------------------------
    int iq = 0, iters = 0, passes = 0, status = MPCB_OK;
    // operation count of the dense method along this pivot sequence (diagnostic)
    const long long nn = N * (POLY ? 2 : 1), mm = sd.m;
    long long falg = 2 * nn * nn;
    unsigned actm = 0, exclm = 0;
    double R_norm = 1.0;
    double zx[NPL], zy[NPL];
    bool pending = false;       // a candidate's partial steps are part of x (cap exit)
    bool zmp_current = false;   // zx, zy hold the ZMP of the final multipliers
    int ipRow = 0, ipKS = 0, ipPh = 0;
    double pip = 0.0, qip = 0.0, ucand = 0.0, s_ip = 0.0;
    if constexpr (WIDE) {
        // Taking over mid-solve: the first edition left the state at the top of the outer iteration
        // it could not finish (working set in order, multipliers, objective, counters).  The factor
        // T = R^-1 of that working set is rebuilt row by row, exactly as add_constraint appends
        // rows (d = T'g, r = T d, new column (-r, 1) / rho), without any of the step-length logic;
        // the solve then carries on with that iteration instead of repeating the ~32 before it.
        const int ntop = warm_i ? ufirst(warm_i[(size_t)inst * WARM_STRIDE + WARM_N]) : 0;
        if (ntop > 0 && ntop <= WARM_QL) {
            const int *wi = warm_i + (size_t)inst * WARM_STRIDE;
            const double *wd = warm_d + (size_t)inst * WARM_STRIDE;
            if (lane < ntop) {
                const int packed = wi[lane], row = packed & 4095, ks = packed >> 12;
                double p, q;
                if constexpr (ROWS) {
                    p = rows[row * 3];
                    q = rows[row * 3 + 1];
                } else {
                    const int ph = sd.knot_phase[ks >> 2];
                    p = tab[(ph * 4 + (ks & 3)) * 3];
                    q = tab[(ph * 4 + (ks & 3)) * 3 + 1];
                }
                W.wsRow[lane] = row;
                W.wsKS[lane] = ks;
                W.wsP[lane] = p;
                W.wsQ[lane] = q;
                W.wsU[lane] = wd[lane];
            }
            wsync();
            const int ksv = ws_knots();
            bool sound = true;
            for (int a = 0; a < ntop; a++) {
                const int ka = __builtin_amdgcn_readlane(ksv, a) >> 2;
                const double pa = ufirst(W.wsP[a]), qa = ufirst(W.wsQ[a]);
                const double g = lane < a ? (W.wsP[lane] * pa + W.wsQ[lane] * qa) * S_pair(ksv >> 2, ka) : 0.0;
                const double mii = ufirst((pa * pa + qa * qa) * S_col(ka, ka));
                const int lq = lane < QL ? lane : QL - 1;
                const double *col = W.Rinv + lq, *rowp = W.Rinv + lq * LDI;
                double da = 0.0, ra = 0.0;
                for (int j = 0; j < a; j += 4) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) da = fma(col[(j + q4) * LDI], lane_bcast(g, j + q4), da);
                }
                da = lane < a ? da : 0.0;
                for (int k = 0; k < a; k += 4) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) ra = fma(rowp[k + q4], lane_bcast(da, k + q4), ra);
                }
                const double zn = ufirst(mii - wave_sum(da * da));
                if (!(zn > 1e-7 * mii)) {
                    sound = false;
                    break;
                }
                double rho, irho;
                sqrt_rsqrt(zn, rho, irho);
                irho = ufirst(irho);
                if (lane < a) W.Rinv[lane * LDI + a] = -ra * irho;
                else if (lane == a) W.Rinv[a * LDI + a] = irho;
                wsync();
            }
            if (sound) {
                for (int a = 0; a < ntop; a++) {
                    const int ks = __builtin_amdgcn_readlane(ksv, a);
                    if (lane == ((ks >> 2) & 63)) actm |= 1u << (4 * ((ks >> 2) >> 6) + (ks & 3));
                }
                iq = ntop;
                iters = ufirst(wi[WARM_ITERS]);
                passes = ufirst(wi[WARM_PASSES]);
                f = ufirst(wd[WARM_F]);
                falg = (long long)ufirst(wd[WARM_FALG]);
                R_norm = ufirst(wd[WARM_RNORM]);
            } else {
                // (not expected: the first edition factorised this very working set) start afresh
                for (int i = lane; i < QL * LDI; i += 64) W.Rinv[i] = 0.0;
                wsync();
            }
        }
    }
    STAMP(0);
    for (;;) { // step 1 (l1)
        iters++;
        if (iters > sd.iter_cap) {
            status = MPCB_ITER_CAP;
            break;
        }
        // zmp at x = x_unc + G^-1 N_A u
        set_coefs(lane < iq ? W.wsU[lane] : 0.0, iq);
#pragma unroll
        for (int c = 0; c < NPL; c++) {
            zx[c] = zxu[c];
            zy[c] = zyu[c];
        }
        const int ks_top = ws_knots(); // lane a: (knot, slot) of working-set position a, as of step 1
        accum_rows(TAB_S, iq, ks_top, zx, zy);
        STAMP(1);
        // s(x) for every row (:255-261), fused with the search for the most violated row
        // (step 2, :279-287, first pass: ss = 0).  Row values are not kept: the rare
        // restart of step 2 recomputes them from zx, zy.
        // mloc / mslot: this lane's lowest row value and the first slot (in row order: rows
        // ascend with (knot, edge) for the polygon rows, with (side, knot) for the box rows)
        // that attains it
        double mloc = DINF;
        int mslot = -1;
        double rowv[NPL][ES]; // the row values, for the sum of the violations where it is needed (below)
        auto visit = [&](int c, int e, double v) {
            rowv[c][e] = v;
            const bool better = v < mloc;
            mloc = hw_min(mloc, v); // (= better ? v : mloc, in one instruction instead of two selects)
            mslot = better ? 4 * c + e : mslot;
        };
        if constexpr (POLY) {
            // one knot's slots at a time; the (p, q, r) of the next knot are loaded while this
            // knot's rows are evaluated.  Slots that do not exist evaluate to +inf (through
            // their table entries, or masked for caller-built rows).
            double sp[2][ES], sq[2][ES], sr[2][ES];
#pragma unroll
            for (int e = 0; e < ES; e++) {
                sp[0][e] = slotP(0, e);
                sq[0][e] = slotQ(0, e);
                sr[0][e] = slotR(0, e);
            }
#pragma unroll
            for (int c = 0; c < NPL; c++) {
                if (c + 1 < NPL) {
#pragma unroll
                    for (int e = 0; e < ES; e++) {
                        sp[(c + 1) & 1][e] = slotP(c + 1, e);
                        sq[(c + 1) & 1][e] = slotQ(c + 1, e);
                        sr[(c + 1) & 1][e] = slotR(c + 1, e);
                    }
                }
#pragma unroll
                for (int e = 0; e < ES; e++) {
                    double v = fma(sp[c & 1][e], zx[c], fma(sq[c & 1][e], zy[c], sr[c & 1][e]));
                    if constexpr (ROWS) v = (validm & (1u << (4 * c + e))) ? v : DINF;
                    visit(c, e, v);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < ES; e++)
#pragma unroll
                for (int c = 0; c < NPL; c++) visit(c, e, e == 0 ? zx[c] - lo[c] : hi[c] - zx[c]);
        }
        // Lowest row value of the wave.  The sum of the violations psi (:259-261) is needed only for
        // the stopping test |psi| <= threshold (:267), and psi -- a sum of non-positive terms, the
        // lowest row value among them -- is at most that value, in floating point as well (adding
        // non-positive numbers never rounds upwards past a term).  While the most violated row alone
        // exceeds the threshold the test fails whatever the sum is, so the sum (a minimum and an
        // addition per row plus a wave reduction) is formed only when that row is within it: on the
        // last iteration or two of a solve.  Same terms, same order, same reduction as ever.
        const double mw_all = wave_min(mloc);
        falg += 2 * mm * nn;
        exclm = 0xffffffffu; // iaexcl[i] = true (:257)
        if (mw_all >= -psi_thr) {
            double psi = 0.0;
            if constexpr (POLY) {
#pragma unroll
                for (int c = 0; c < NPL; c++)
#pragma unroll
                    for (int e = 0; e < ES; e++) psi += fmin(0.0, rowv[c][e]);
            } else {
#pragma unroll
                for (int e = 0; e < ES; e++)
#pragma unroll
                    for (int c = 0; c < NPL; c++) psi += fmin(0.0, rowv[c][e]);
            }
            psi = wave_sum(psi);
            if (fabs(psi) <= psi_thr) { // :267-272
                zmp_current = true;
                break;
            }
        }
        {
            // (a fresh copy of the lane number: the addresses of these arrays would otherwise be
            // computed once in front of the loop and kept -- in the slim build, spilled -- through it)
            int lo_ = lane;
            asm volatile("" : "+v"(lo_));
            if (lo_ < iq) { // :275-277
                W.oldU[lo_] = W.wsU[lo_];
                W.oldRow[lo_] = W.wsRow[lo_];
                W.oldKS[lo_] = W.wsKS[lo_];
            }
        }
        // (coupled forms hand over with a record only when the working set is full at a full step,
        // and inside an outer iteration it can only shrink before that step)
        if (warm_i && (!POLY || iq >= QL) && lane == 0) {
            W.topPasses = passes;
            W.topIq = iq;
            W.topF = f;
            W.topFalg = (double)(falg - 2 * mm * nn); // (this iteration's row evaluation is counted again by whoever redoes it)
        }
        STAMP(2);
        double ssv = 0.0;
        bool done = false, restart, restart_seen = false;
        do { // step 2 (l2), :279-299
            restart = false;
            // Most violated row (:279-287).  First pass: the wave minimum of all row values,
            // then the lowest row attaining it (rows ascend with (knot, edge) for the polygon
            // rows and with (side, knot) for the box rows).  If that row is in the working set
            // -- its value is a rounding residue -- the masked search below decides instead.
            // The winner's phase and first row come from the registers of the lane that owns
            // its knot (one readlane each) rather than from the per-knot tables in memory: a
            // dependent round trip to L2 per candidate would sit on the critical path.
            double bv = DINF;
            int brow = 0x7fffffff, bks = 0, bph = 0, brow0 = 0;
            bool masked = restart_seen;
            if (!masked) {
                const double mw = mw_all;
                if (mw < 0.0) {
                    const int slot = mloc == mw ? mslot : -1;
                    const int ks = ((lane + 64 * (slot >> 2)) << 2) | (slot & 3);
                    const int key = slot < 0 ? 0x7fffffff : POLY ? ks : lane + 64 * (slot >> 2) + (slot & 3) * N;
                    // (whether this lane's candidate is in the working set, taken here: past the tie
                    // test only the key and this bit have to survive, not the slot)
                    const int myact = (int)((actm >> (slot & 31)) & 1u);
                    // one lane attains the minimum unless rows tie exactly; only then does the
                    // lowest row have to be found across lanes
                    unsigned long long win = __ballot(slot >= 0);
                    int kmin;
                    if (__popcll(win) > 1) {
                        kmin = wave_min(key);
                        win = __ballot(key == kmin);
                    }
                    const int src = __ffsll((long long)win) - 1;
                    kmin = __builtin_amdgcn_readlane(key, src);
                    const int act = __builtin_amdgcn_readlane(myact, src);
                    bks = POLY ? kmin : __builtin_amdgcn_readlane(ks, src); // (the polygon rows' key is (knot, slot) itself)
                    if constexpr (POLY) {
                        // which of the winner lane's knots: a wave-uniform choice among
                        // readlanes (a per-lane choice would be an indexed register array)
                        const int cs = bks >> 8;
                        int pr = __builtin_amdgcn_readlane(phr[0], src);
#pragma unroll
                        for (int c = 1; c < NPL; c++) {
                            const int t = __builtin_amdgcn_readlane(phr[c], src);
                            pr = cs == c ? t : pr;
                        }
                        bph = pr & 1023;
                        brow0 = pr >> 10;
                    }
                    brow = kmin;
                    bv = mw;
                    masked = act != 0;
                }
            }
            if (masked) {
                // rows below the running minimum ss that are neither active nor excluded (the
                // second pass of step 2 after a degenerate add ends up here as well)
                bv = DINF;
                brow = 0x7fffffff;
                bks = 0;
                int lph = 0, lrow0 = 0;
                // (This search is rare.  Its inputs pass through empty asm statements so that
                // nothing derived from them counts as loop-invariant: hoisted out of the main
                // loop, the per-slot masks and keys would sit in registers -- or spill -- for
                // the whole solve.)
                // (rows the library builds itself: slots that do not exist evaluate to +inf through their
                // table entries and are never below ss, so the mask of existing slots is not needed --
                // and not kept alive through the main loop)
                unsigned elig = ((POLY && !ROWS) ? 0xffffffffu : validm) & ~actm & exclm;
                int lk = lane_k;
                asm volatile("" : "+v"(elig), "+v"(lk));
#pragma unroll
                for (int c = 0; c < NPL; c++)
#pragma unroll
                    for (int e = 0; e < ES; e++) {
                        const double v = fma(slotP(c, e), zx[c], fma(slotQ(c, e), zy[c], slotR(c, e)));
                        const int ks = ((lk + 64 * c) << 2) | e;
                        const int row = POLY ? ks : lk + 64 * c + e * N;
                        const bool take = ((elig >> (4 * c + e)) & 1u) && v < ssv && (v < bv || (v == bv && row < brow));
                        bv = take ? v : bv;
                        brow = take ? row : brow;
                        bks = take ? ks : bks;
                        lph = take ? PH(c) : lph;
                        lrow0 = take ? ROW0(c) : lrow0;
                    }
                const int src = wave_argmin(bv, brow, bks);
                bph = __builtin_amdgcn_readlane(lph, src);
                brow0 = __builtin_amdgcn_readlane(lrow0, src);
            }
            STAMP(3);
            if (bv < ssv) {
                ssv = bv;
                ipKS = bks;
                ipPh = bph;
                ipRow = POLY ? brow0 + (bks & 3) : brow;
                s_ip = bv;
            }
            if (ssv >= 0.0) {
                zmp_current = !restart_seen; // after a degenerate restart u was restored, zx is stale
                done = true;
                break;
            }
            const int kip = ipKS >> 2, eip = ipKS & 3;
            if constexpr (ROWS) {
                pip = ufirst(rows[ipRow * 3]);
                qip = ufirst(rows[ipRow * 3 + 1]);
            } else if constexpr (POLY) {
                pip = ufirst(tab[(ipPh * 4 + eip) * 3]);
                qip = ufirst(tab[(ipPh * 4 + eip) * 3 + 1]);
            } else {
                pip = eip == 0 ? 1.0 : -1.0;
                qip = 0.0;
            }
            ucand = 0.0; // u(iq) = 0 (:297)
            pending = true;
            // n_ip' G^-1 n_ip; the load is consumed after the gather of the first pass has been
            // issued, so that the two memory round trips overlap
            const double skk = S_col(kip, kip);
            // g_a = n_a' G^-1 n_ip depends on the pair (a, ip) only: it is gathered once per
            // candidate, and a drop merely closes the gap (lanes above the dropped position move
            // down by one) instead of paying another round trip to the table
            double ga = 0.0;
            bool have_g = false;

            for (;;) { // step 2a (l2a), :306-446
                passes++;
                if (passes > sd.iter_cap) {
                    status = MPCB_ITER_CAP;
                    done = true;
                    break;
                }
                falg += 2 * nn * nn + 2 * nn * (nn - iq) + (long long)iq * iq;
                // g = N_A' G^-1 n_ip ; d = R^-T g (= J1' n_ip) ; r = R^-1 d
                double da = 0.0, ra = 0.0, ua = 0.0;
                if (lane < iq) {
                    if (!have_g) { // (the working set is still that of step 1: ks_top is current)
                        const int ka = ks_top >> 2;
                        ga = (W.wsP[lane] * pip + W.wsQ[lane] * qip) * S_pair(ka, kip);
                    }
                    ua = W.wsU[lane];
                }
                have_g = true;
                const double mii = ufirst((pip * pip + qip * qip) * skk);
                STAMP(4);
                {
                    // d_i = sum_j Rinv[j][i] g_j, r_i = sum_k Rinv[i][k] d_k: plain sums, the
                    // zeros of Rinv take care of the triangle and of lanes >= iq.  g_j and d_k are
                    // handed round by readlane (scalar operands of the FMAs), not through LDS.
                    const int lq = lane < QL ? lane : QL - 1;
                    const double *col = W.Rinv + lq, *rowp = W.Rinv + lq * LDI;
                    // (four terms per trip, written out: the trip count is rounded up, the extra
                    // terms meet zeros of Rinv)
                    if constexpr (POLY) {
                        // (g and d travel through the coefficient arrays, dead between the row sums of
                        // step 1 and the next ones: a broadcast LDS read per pair of terms where a
                        // readlane pair per term cost two vector instructions -- the kernel is bound by
                        // vector issue, the LDS pipeline is not)
                        if (lane < QL + 4) W.cx[lane] = lane < iq ? ga : 0.0;
                        wsync();
                        for (int j = 0; j < iq; j += 4) {
                            double rv[4], gv[4];
#pragma unroll
                            for (int q4 = 0; q4 < 4; q4++) {
                                rv[q4] = col[(j + q4) * LDI];
                                gv[q4] = W.cx[j + q4];
                            }
#pragma unroll
                            for (int q4 = 0; q4 < 4; q4++) da = fma(rv[q4], gv[q4], da);
                        }
                        da = lane < iq ? da : 0.0;
                        if (lane < QL + 4) W.cy[lane] = da;
                        wsync();
                        for (int k = 0; k < iq; k += 4) {
                            double rv[4], dv[4];
#pragma unroll
                            for (int q4 = 0; q4 < 4; q4++) {
                                rv[q4] = rowp[k + q4];
                                dv[q4] = W.cy[k + q4];
                            }
#pragma unroll
                            for (int q4 = 0; q4 < 4; q4++) ra = fma(rv[q4], dv[q4], ra);
                        }
                    } else {
                    for (int j = 0; j < iq; j += 4) {
                        double rv[4], gv[4];
#pragma unroll
                        for (int q4 = 0; q4 < 4; q4++) {
                            rv[q4] = col[(j + q4) * LDI];
                            gv[q4] = lane_bcast(ga, j + q4);
                        }
#pragma unroll
                        for (int q4 = 0; q4 < 4; q4++) da = fma(rv[q4], gv[q4], da);
                    }
                    da = lane < iq ? da : 0.0;
                    for (int k = 0; k < iq; k += 4) {
                        double rv[4], dv[4];
#pragma unroll
                        for (int q4 = 0; q4 < 4; q4++) {
                            rv[q4] = rowp[k + q4];
                            dv[q4] = lane_bcast(da, k + q4);
                        }
#pragma unroll
                        for (int q4 = 0; q4 < 4; q4++) ra = fma(rv[q4], dv[q4], ra);
                    }
                    }
                    ra = lane < iq ? ra : 0.0;
                }
                STAMP(5);
                // (d, r and the step lengths live on the first iq lanes: while the working set fits
                // one row of 16 lanes the reductions over them skip the cross-row steps)
                // (The box form keeps the full-wave sum: the paths of its numerically infeasible draws
                // -- ~200 outer iterations each, and with them the running time of the second stage --
                // change with the last bit of zn, and a different contraction of da * da into the
                // first addition is enough for that.)
                const bool small_ws = iq <= 16;
                const double zn = ufirst(mii - ((POLY && small_ws) ? row0_sum(da * da) : wave_sum(da * da))); // z . n_ip
                // rho = sqrt(zn) is the new diagonal entry of R if the constraint is added; its
                // reciprocal also gives the full step length t2 = -s_ip / zn without a division
                // (computed without a branch so that this chain and the division chain of the
                // step length below are scheduled into each other's gaps)
                double rho, irho;
                sqrt_rsqrt(zn > 0.0 ? zn : 1.0, rho, irho);
                rho = zn > 0.0 ? ufirst(rho) : 0.0;
                irho = zn > 0.0 ? ufirst(irho) : 0.0;
                // step 2b: t1 = min u_k / r_k over r_k > 0, first minimum wins (:322-334)
                double t1 = (lane < iq && ra > 0.0) ? ua / ra : DINF;
                int apos = lane;
                {
                    const double tmin = small_ws ? row0_min(t1) : wave_min(t1);
                    // lowest position among the lanes that attain the minimum (all hold +inf if
                    // no r is positive: position 0 then, which nothing uses)
                    const unsigned long long win = __ballot(t1 == tmin);
                    apos = win ? __ffsll((long long)win) - 1 : 0;
                    t1 = tmin;
                }
                STAMP(6);
                // z != 0 test (:336): ||z||^2 >= zn / lambda_max(G1) >= zn / trace(G1)
                bool znz;
                if (zn > 1e-5 * mii && zn > 64.0 * EPS * sd.traceG1) {
                    znz = true;
                } else {
                    // explicit z = G^-1 (n_ip - N_A r)
                    set_coefs(-ra, iq);
                    double acc = 0.0;
                    double zvx[NPL], zvy[NPL];
#pragma unroll
                    for (int c = 0; c < NPL; c++) {
                        const double w = Wt_row(kip, kcb[c]);
                        zvx[c] = pip * w;
                        zvy[c] = qip * w;
                    }
                    accum_rows(TAB_W, iq, ws_knots(), zvx, zvy);
#pragma unroll
                    for (int c = 0; c < NPL; c++)
                        if (KN(c) < N) acc += zvx[c] * zvx[c] + zvy[c] * zvy[c];
                    znz = fabs(wave_sum(acc)) > EPS;
                    wsync();
                }
                if (znz && !(zn > (POLY ? 1e-7 : 1e-11) * mii)) {
                    // n_ip is numerically dependent on the working set although z is
                    // not negligible: the reduced system has lost its accuracy.
                    status = STATUS_UNRESOLVED;
                    done = true;
                    break;
                }
                const double t2 = znz ? ufirst(-s_ip * irho * irho) : DINF;
                const double t = fmin(t1, t2);
                STAMP(9);
                if (t >= DINF) { // :350-356
                    status = MPCB_INFEASIBLE;
                    f = DINF;
                    done = true;
                    break;
                }
                if (t2 >= DINF) { // dual step only, drop (:358-373)
                    if (lane < iq) W.wsU[lane] = ua - t * ra;
                    ucand = ufirst(ucand + t);
                    const int ksd = ufirst(W.wsKS[apos]);
                    if (lane == ((ksd >> 2) & 63)) actm &= ~(1u << (4 * ((ksd >> 2) >> 6) + (ksd & 3)));
                    wsync();
                    {
                        const long long cdrop = iq - 1 - apos;
                        falg += 6 * nn * cdrop + 3 * cdrop * (cdrop - 1);
                    }
                    if constexpr (POLY) drop_constraint(W, lane, iq, apos);
                    else drop_constraint_rotations(W, lane, iq, apos);
                    {
                        const double up = dpp_f64<0x130, 0xF>(0.0, ga); // wave_shl:1 -- value of lane + 1
                        ga = lane >= apos ? up : ga;
                    }
                    continue;
                }
                // step in primal and dual space (:377-382)
                f = ufirst(f + t * zn * (0.5 * t + ucand));
                if (lane < iq) W.wsU[lane] = ua - t * ra;
                ucand = ufirst(ucand + t);
                if (t == t2) { // full step: add the constraint (:392-426, add_constraint :450-513)
                    if (iq >= (POLY ? QL : QL_BOX)) {
                        // (a wide edition, where one follows, takes the instances that are merely too big)
                        // (`wide_list`: the next edition's list -- the wide edition's for the first, the
                        // block edition's for the wide one)
                        status = wide_list ? (WIDE ? STATUS_TO_BLOCK : STATUS_TO_WIDE) : STATUS_UNRESOLVED;
                        done = true;
                        break;
                    }
                    falg += 6 * nn * (nn - iq - 1);
                    if (rho <= EPS * R_norm) {
                        // degenerate: exclude ip, restore u, A (and x) of step 1, retry step 2
                        const int bit = 4 * (kip >> 6) + eip;
                        if (lane == (kip & 63)) exclm &= ~(1u << bit);
                        wsync();
                        if (lane < iq) {
                            W.wsU[lane] = W.oldU[lane];
                            // (p, q) of the restored row, from where they were first taken
                            const int oks = W.oldKS[lane];
                            double op, oq;
                            if constexpr (ROWS) {
                                op = rows[W.oldRow[lane] * 3];
                                oq = rows[W.oldRow[lane] * 3 + 1];
                            } else if constexpr (POLY) {
                                const int oph = sd.knot_phase[oks >> 2];
                                op = tab[(oph * 4 + (oks & 3)) * 3];
                                oq = tab[(oph * 4 + (oks & 3)) * 3 + 1];
                            } else {
                                op = (oks & 3) == 0 ? 1.0 : -1.0;
                                oq = 0.0;
                            }
                            W.wsP[lane] = op;
                            W.wsQ[lane] = oq;
                            W.wsRow[lane] = W.oldRow[lane];
                            W.wsKS[lane] = W.oldKS[lane];
                        }
                        wsync();
                        actm = 0;
                        for (int a = 0; a < iq; a++) {
                            const int ks = ufirst(W.wsKS[a]);
                            if (lane == ((ks >> 2) & 63)) actm |= 1u << (4 * ((ks >> 2) >> 6) + (ks & 3));
                        }
                        pending = false;
                        restart = true;
                        restart_seen = true;
                        break;
                    }
                    R_norm = fmax(R_norm, rho);
                    STAMP(10);
                    if constexpr (!POLY) { // R is kept for drop_constraint_rotations
                        if (lane < iq) W.R[tri(iq) + lane] = da;
                        else if (lane == iq) W.R[tri(iq) + iq] = rho;
                    }
                    if (lane < iq) {
                        W.Rinv[lane * LDI + iq] = -ra * irho;
                    } else if (lane == iq) {
                        W.Rinv[iq * LDI + iq] = irho;
                        W.wsU[iq] = ucand;
                        W.wsP[iq] = pip;
                        W.wsQ[iq] = qip;
                        W.wsRow[iq] = ipRow;
                        W.wsKS[iq] = ipKS;
                    }
                    if (lane == (kip & 63)) actm |= 1u << (4 * (kip >> 6) + eip);
                    iq++;
                    pending = false;
                    wsync();
                    break; // goto l1
                }
                // partial step: drop the blocking constraint and try again (:434-446)
                s_ip = ufirst(s_ip + t * zn);
                {
                    const int ksd = ufirst(W.wsKS[apos]);
                    if (lane == ((ksd >> 2) & 63)) actm &= ~(1u << (4 * ((ksd >> 2) >> 6) + (ksd & 3)));
                }
                wsync();
                {
                    const long long cdrop = iq - 1 - apos;
                    falg += 6 * nn * cdrop + 3 * cdrop * (cdrop - 1);
                }
                if constexpr (POLY) drop_constraint(W, lane, iq, apos);
                else drop_constraint_rotations(W, lane, iq, apos);
                {
                    const double up = dpp_f64<0x130, 0xF>(0.0, ga); // wave_shl:1 -- value of lane + 1
                    ga = lane >= apos ? up : ga;
                }
            }
        } while (restart);
        STAMP(7);
        if (done) break;
    }
    STAMP(7);

    // ---- outputs -------------------------------------------------------------
    // (fresh copies of the lane and instance numbers: nothing derived from them before the
    // main loop -- 64-bit offsets, mostly -- is kept in registers, or spilled, across it)
    int inst_e = inst;
    asm volatile("" : "+v"(lane_k), "+s"(inst_e));
    const int lane_e = lane_k;
    // The output pointers are fetched from the kernel-argument segment here, where they are first
    // needed: taken from the kernel's parameters they are loaded at its entry and then sit in
    // scalar registers -- in practice in lanes of spill registers -- through the whole solve.
    mpcb_outputs out;
    {
        typedef const __attribute__((address_space(4))) char *kernarg_bytes;
        kernarg_bytes ka = (kernarg_bytes)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const __attribute__((address_space(4))) mpcb_outputs *ko =
            (const __attribute__((address_space(4))) mpcb_outputs *)(ka + offsetof(LeadingArgs, out));
        out.jerk_x = ko->jerk_x; out.jerk_y = ko->jerk_y; out.com_x = ko->com_x; out.com_y = ko->com_y;
        out.com_xd = ko->com_xd; out.com_yd = ko->com_yd; out.zmp_x = ko->zmp_x; out.zmp_y = ko->zmp_y;
        out.min_margin = ko->min_margin; out.cost = ko->cost; out.status = ko->status; out.iters = ko->iters;
        out.n_active = ko->n_active; out.active_idx = ko->active_idx; out.multipliers = ko->multipliers;
        out.active_stride = ko->active_stride; out.alg_flops = ko->alg_flops; out.slacks = ko->slacks;
    }
    const bool solved = status == MPCB_OK || status == MPCB_ITER_CAP;
    if (status == STATUS_UNRESOLVED && lane_e == 0 && unresolved_list)
        unresolved_list[atomicAdd(unresolved_count, 1)] = inst_e; // re-solved by the dense kernel
    if (status == (WIDE ? STATUS_TO_BLOCK : STATUS_TO_WIDE) && lane_e == 0) wide_list[atomicAdd(wide_count, 1)] = inst_e;
    if constexpr (RECORD) {
        if (lane_e == 0) {
            // bin = running time on a logarithmic scale, 16 bins to the octave (hand-overs count as
            // the longest)
            unsigned dt = (unsigned)__builtin_amdgcn_s_memtime() - t_start;
            dt = dt < 16u ? 16u : dt;
            const int oct = 31 - __builtin_clz(dt);
            int bin = (oct - 4) * 16 + (int)((dt >> (oct - 4)) & 15u);
            bin = bin < ORDER_BINS - 2 ? bin : ORDER_BINS - 2;
            if (status == STATUS_UNRESOLVED || status == STATUS_TO_WIDE) bin = ORDER_BINS - 1;
            order_record[ORDER_HEADER + inst_e] = bin;
        }
    }
    if (!WIDE && warm_i && status == (POLY ? STATUS_TO_WIDE : STATUS_UNRESOLVED)) {
        // where the next stage takes over (box problems: the dense kernel; coupled forms: the wide
        // edition of this kernel): the top of this outer iteration
        int *wi = warm_i + (size_t)inst_e * WARM_STRIDE;
        double *wd = warm_d + (size_t)inst_e * WARM_STRIDE;
        const int ntop = ufirst(W.topIq);
        if (lane_e < ntop) {
            // (coupled forms: the row and its (knot, slot), 12 bits each)
            wi[lane_e] = POLY ? (W.oldRow[lane_e] | (W.oldKS[lane_e] << 12)) : W.oldRow[lane_e];
            wd[lane_e] = W.oldU[lane_e];
        }
        if (lane_e == 0) {
            wi[WARM_N] = ntop;
            wi[WARM_ITERS] = iters - 1;
            wi[WARM_PASSES] = W.topPasses;
            wd[WARM_F] = W.topF;
            wd[WARM_FALG] = W.topFalg;
            wd[WARM_RNORM] = R_norm;
        }
    }
    if (lane_e == 0) {
        if (out.cost) out.cost[inst_e] = f;
        if (out.status) out.status[inst_e] = status;
        if (out.iters) out.iters[inst_e] = iters;
        if (out.n_active) out.n_active[inst_e] = iq;
        if (out.alg_flops) out.alg_flops[inst_e] = (double)(falg + (POLY ? 2 * mm * nn : 0));
    }
    if (out.active_idx || out.multipliers) {
        for (int a = lane_e; a < out.active_stride; a += 64) {
            const bool in_set = a < iq && a <= QL;
            if (out.active_idx) out.active_idx[(size_t)inst_e * out.active_stride + a] = in_set ? W.wsRow[a] : -1;
            if (out.multipliers) out.multipliers[(size_t)inst_e * out.active_stride + a] = in_set ? W.wsU[a] : 0.0;
        }
    }
    // multipliers -> trajectories through the response matrices
    int nresp = iq;
    set_coefs(lane_e < iq ? W.wsU[lane_e] : 0.0, iq);
    if (pending && status == MPCB_ITER_CAP && iq < QL) {
        // partial steps of an unfinished candidate are part of x
        if (lane_e == 0) {
            W.cx[iq] = ucand * pip;
            W.cy[iq] = ucand * qip;
            W.wsKS[iq] = ipKS;
        }
        nresp = iq + 1;
        wsync();
    }
    auto store_pair = [&](double *ox, double *oy, const double *vx, const double *vy) {
#pragma unroll
        for (int c = 0; c < NPL; c++)
            if (KN(c) < N) {
                if (ox) ox[(size_t)inst_e * N + KN(c)] = solved ? vx[c] : 0.0;
                if (oy) oy[(size_t)inst_e * N + KN(c)] = solved ? vy[c] : 0.0;
            }
    };
    STAMP(12);
    const bool want_com = out.com_x || out.com_y || out.com_xd || out.com_yd;
    if (out.jerk_x || out.jerk_y || want_com) {
        // jerk = unconstrained response + response to the multipliers
        double jx[NPL], jy[NPL];
#pragma unroll
        for (int c = 0; c < NPL; c++) {
            const double *k4 = sd.Kx + (size_t)KC(c) * 4;
            jx[c] = k4[0] * wx[0] + k4[1] * wx[1] + k4[2] * wx[2] + k4[3] * wx[3];
            jy[c] = k4[0] * wy[0] + k4[1] * wy[1] + k4[2] * wy[2] + k4[3] * wy[3];
        }
        if constexpr (POLY) {
            if (sd.Wp) accum_phases(sd.Wp, jx, jy);
        }
        accum_rows(TAB_W, nresp, ws_knots(), jx, jy);
        store_pair(out.jerk_x, out.jerk_y, jx, jy);
        STAMP(13);
        if (want_com) {
            // computeCOMtrajectory (MPCPlanner.cpp:196-209) without its N x N products.  The
            // triple integrator's response to the jerk is a convolution with a quadratic in
            // the knot distance p = k - j:  position T^3 (1/6 + p/2 + p^2/2), velocity
            // T^2 (1/2 + p).  With the running moments S0 = sum u_j, S1 = sum j u_j,
            // S2 = sum j^2 u_j over j <= k:
            //   sum p u_j = k S0 - S1,   sum p^2 u_j = k (k S0 - S1) - (k S1 - S2).
            // For this the jerk is re-dealt through LDS (the factor's storage is free by now)
            // so that every lane_e holds NPL consecutive knots: the moments run serially inside
            // a lane_e and need one prefix scan across the lanes each.
            double *sx = W.Rinv, *sy = W.Rinv + 64 * NPL;
            static_assert(2 * 64 * NPL <= QL * LDI, "jerk exchange does not fit the factor storage");
            wsync();
#pragma unroll
            for (int c = 0; c < NPL; c++) {
                sx[lane_e + 64 * c] = jx[c];
                if constexpr (POLY) sy[lane_e + 64 * c] = jy[c];
            }
            wsync();
            const int k0 = NPL * lane_e;
            const double T1 = sd.Ts, T2 = T1 * T1, T3 = T2 * T1;
            auto integrate = [&](const double *su, const double *w, double *op, double *ov) {
                if (!op && !ov) return;
                double u[NPL], l0[NPL], l1[NPL], l2[NPL];
#pragma unroll
                for (int i = 0; i < NPL; i++) u[i] = su[k0 + i];
                double m0 = 0.0, m1 = 0.0, m2 = 0.0;
#pragma unroll
                for (int i = 0; i < NPL; i++) {
                    const double k = (double)(k0 + i);
                    m0 += u[i];
                    m1 = fma(k, u[i], m1);
                    m2 = fma(k * k, u[i], m2);
                    l0[i] = m0;
                    l1[i] = m1;
                    l2[i] = m2;
                }
                // moments of the lanes before this one
                const double e0 = wave_scan(m0) - m0, e1 = wave_scan(m1) - m1, e2 = wave_scan(m2) - m2;
                double pos[NPL], vel[NPL];
#pragma unroll
                for (int i = 0; i < NPL; i++) {
                    const double k = (double)(k0 + i), tk = (k + 1.0) * T1;
                    const double S0 = e0 + l0[i], S1 = e1 + l1[i], S2 = e2 + l2[i];
                    const double lx = fma(k, S0, -S1), mx = fma(k, lx, -fma(k, S1, -S2));
                    pos[i] = w[0] + tk * (w[1] + 0.5 * tk * w[2]) + T3 * (S0 * (1.0 / 6.0) + 0.5 * (lx + mx));
                    vel[i] = w[1] + tk * w[2] + T2 * (0.5 * S0 + lx);
                    if (!solved) {
                        pos[i] = 0.0;
                        vel[i] = 0.0;
                    }
                }
                auto store_run = [&](double *o, const double *v) {
                    if (!o) return;
                    double *dst = o + (size_t)inst_e * N + k0;
                    if ((NPL & 1) == 0 && (reinterpret_cast<size_t>(dst) & 15) == 0) {
#pragma unroll
                        for (int i = 0; i + 1 < NPL; i += 2) {
                            if (k0 + i + 1 < N) *reinterpret_cast<double2 *>(dst + i) = make_double2(v[i], v[i + 1]);
                            else if (k0 + i < N) dst[i] = v[i];
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < NPL; i++)
                            if (k0 + i < N) dst[i] = v[i];
                    }
                };
                store_run(op, pos);
                store_run(ov, vel);
            };
            integrate(sx, wx, out.com_x, out.com_xd);
            if constexpr (POLY) {
                integrate(sy, wy, out.com_y, out.com_yd);
            } else {
                // one axis only: the other axis' outputs are identically zero
                double zero[NPL];
#pragma unroll
                for (int c = 0; c < NPL; c++) zero[c] = 0.0;
                store_pair(nullptr, out.com_y, zero, zero);
                store_pair(nullptr, out.com_yd, zero, zero);
            }
        }
    }
    STAMP(14);
    if (out.zmp_x || out.zmp_y || out.min_margin) {
        // On a regular exit zx, zy were evaluated at the final multipliers in step 1.
        if (!zmp_current) {
#pragma unroll
            for (int c = 0; c < NPL; c++) {
                zx[c] = zxu[c];
                zy[c] = zyu[c];
            }
            accum_rows(TAB_S, nresp, ws_knots(), zx, zy);
        }
#pragma unroll
        for (int c = 0; c < NPL; c++)
            if (KN(c) < N) {
                if (out.zmp_x) out.zmp_x[(size_t)inst_e * N + KN(c)] = solved ? zx[c] : 0.0;
                if (out.zmp_y) out.zmp_y[(size_t)inst_e * N + KN(c)] = solved ? zy[c] : 0.0;
                if (out.min_margin) {
                    // getConstraintViolation (MPCPlanner.cpp:1132-1153)
                    double mn = DINF;
#pragma unroll
                    for (int e = 0; e < ES; e++)
                        if ((POLY && !ROWS) || (validm & (1u << (4 * c + e))))
                            mn = fmin(mn, slotP(c, e) * zx[c] + slotQ(c, e) * zy[c] + slotR(c, e));
                    out.min_margin[(size_t)inst_e * N + KN(c)] = solved ? mn : 0.0;
                }
            }
    }
#ifdef MPCB_STAMPS
    STAMP(8);
    if (lane == 0 && sd.dbg)
        for (int i = 0; i < 16; i++) sd.dbg[(size_t)inst * 16 + i] = stamp_acc[i];
#endif
}

// First edition: one QP per single-wave workgroup, instance = workgroup index.
template <int NPL, bool POLY, bool ROWS, bool HV>
__global__ __launch_bounds__(64, NPL <= 4 ? (POLY ? 3 : 2) : 1) void zspace_kernel(DevSetup sd, DevIn in, mpcb_outputs out, int B,
                                                      int *unresolved_list, int *unresolved_count,
                                                      int *warm_i, double *warm_d, int *wide_list,
                                                      int *wide_count) {
    extern __shared__ double smem[];
    const int inst = ufirst((int)blockIdx.x);
    if (inst >= B) return;
    solve_instance<NPL, POLY, ROWS, HV, false, false>(sd, in, out, inst, (int)(threadIdx.x & 63), smem,
                                                      unresolved_list, unresolved_count, warm_i, warm_d, wide_list,
                                                      wide_count, nullptr);
}

// The same with a dispatch order: workgroup w solves instance order[w] and notes its running time
// in order_record.
template <int NPL, bool POLY, bool ROWS, bool HV>
__global__ __launch_bounds__(64, NPL <= 4 ? (POLY ? 3 : 2) : 1) void zspace_ordered_kernel(DevSetup sd, DevIn in, mpcb_outputs out,
                                                      int B, int *unresolved_list, int *unresolved_count,
                                                      int *warm_i, double *warm_d, int *wide_list,
                                                      int *wide_count, const int *order, int *order_record) {
    extern __shared__ double smem[];
    if ((int)blockIdx.x >= B) return;
    const int inst = ufirst(order[blockIdx.x]);
    solve_instance<NPL, POLY, ROWS, HV, false, true>(sd, in, out, inst, (int)(threadIdx.x & 63), smem,
                                                     unresolved_list, unresolved_count, warm_i, warm_d, wide_list,
                                                     wide_count, order_record);
}

// Longest first.  The hardware hands workgroups to free wave slots in index order, and a QP
// takes between half and twice the average time, so with only a few QPs per slot the launch
// ends with most slots idle while the last-started long ones finish.  When the previous solve
// through this record was of the same batch size (rec[0] == B), rec[ORDER_HEADER + i] is the
// running-time bin of its instance i, and this kernel (one workgroup) turns the bins into the
// permutation that starts the highest bin first: a counting sort, position = instances in higher
// bins + arrivals before it in its own.  Otherwise the order is the identity.  It also stamps the
// record with B for the coming solve (which overwrites the bins as its instances finish) and
// clears the two hand-over counters.
constexpr int ORDER_THREADS = 1024;
__global__ __launch_bounds__(ORDER_THREADS) void dispatch_order_kernel(int *rec, int *order, int *counts, int B) {
    __shared__ int cursor[ORDER_BINS];
    const int t = threadIdx.x;
    const bool valid = rec[0] == B;
    if (valid) {
        if (t < ORDER_BINS) cursor[t] = 0;
        __syncthreads();
        for (int i = t; i < B; i += ORDER_THREADS) atomicAdd(&cursor[rec[ORDER_HEADER + i] & (ORDER_BINS - 1)], 1);
        __syncthreads();
        int above = 0;
        if (t < ORDER_BINS)
            for (int b = t + 1; b < ORDER_BINS; b++) above += cursor[b];
        __syncthreads();
        if (t < ORDER_BINS) cursor[t] = above;
        __syncthreads();
        for (int i = t; i < B; i += ORDER_THREADS)
            order[atomicAdd(&cursor[rec[ORDER_HEADER + i] & (ORDER_BINS - 1)], 1)] = i;
    } else {
        for (int i = t; i < B; i += ORDER_THREADS) order[i] = i;
    }
    __syncthreads();
    if (t == 0) {
        rec[0] = B;
        counts[0] = 0;
        counts[1] = 0;
    }
}

// Wide edition: the workgroups (single waves again) share out the list the first edition left.
template <int NPL, bool ROWS, bool HV>
__global__ __launch_bounds__(64, 1) void zspace_wide_kernel(DevSetup sd, DevIn in, mpcb_outputs out,
                                                            const int *work_list, const int *work_count,
                                                            int *unresolved_list, int *unresolved_count,
                                                            int *warm_i, double *warm_d, int *block_list,
                                                            int *block_count) {
    extern __shared__ double smem[];
    const int count = ufirst(*work_count);
    for (int entry = blockIdx.x; entry < count; entry += gridDim.x) {
        solve_instance<NPL, true, ROWS, HV, true, false>(sd, in, out, ufirst(work_list[entry]),
                                                         (int)(threadIdx.x & 63), smem, unresolved_list,
                                                         unresolved_count, warm_i, warm_d, block_list, block_count,
                                                         nullptr);
        wsync();
    }
}

template <bool POLY, int NPL, bool WIDE, bool HV>
size_t wave_lds_bytes(const DevSetup &sd) {
    return sizeof(typename WaveConfig<POLY, NPL, WIDE, HV>::State) +
           (POLY ? ((size_t)(sd.nphase + 1) * 12 + 2 * (size_t)std::max(sd.nphase, sd.nref)) * 8 : 0);
}

template <int NPL, bool POLY, bool ROWS, bool HV>
hipError_t launch_one(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, int B,
                      const DenseWs &ws, hipStream_t stream) {
    // One wave (= one QP) per workgroup: waves never wait for each other, so a workgroup's LDS
    // and wave slot are released the moment its QP is done instead of when the slowest of
    // several is (iteration counts differ by 2-3x between instances).
    const size_t shmem = wave_lds_bytes<POLY, NPL, false, HV>(sd);
    // (the ordered edition exists for the library's own footstep inputs at the handle's height;
    // batches with caller-built rows or per-instance heights run in index order whatever the hint)
    if constexpr (!ROWS && !HV) {
        if (ws.order && ws.order_record) {
            auto kfn = zspace_ordered_kernel<NPL, POLY, ROWS, HV>;
            if (shmem > 48 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(kfn, dim3(B), dim3(64), shmem, stream, sd, in, out, B, ws.list, ws.count, ws.warm_i,
                               ws.warm_d, POLY ? ws.wide_list : nullptr, ws.wide_count, ws.order, ws.order_record);
            return hipGetLastError();
        }
    }
    auto kfn = zspace_kernel<NPL, POLY, ROWS, HV>;
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(B), dim3(64), shmem, stream, sd, in, out, B, ws.list, ws.count, ws.warm_i,
                       ws.warm_d, POLY ? ws.wide_list : nullptr, ws.wide_count);
    return hipGetLastError();
}

template <int NPL, bool ROWS, bool HV>
hipError_t launch_one_wide(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, int blocks,
                           const DenseWs &ws, hipStream_t stream) {
    const size_t shmem = wave_lds_bytes<true, NPL, true, HV>(sd);
    auto kfn = zspace_wide_kernel<NPL, ROWS, HV>;
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(64), shmem, stream, sd, in, out, ws.wide_list, ws.wide_count, ws.list,
                       ws.count, ws.warm_i, ws.warm_d, ws.block_list, ws.count + CNT_BLOCK);
    return hipGetLastError();
}

template <bool POLY, bool ROWS, bool HV>
hipError_t launch_npl(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, int B, const DenseWs &ws,
                      hipStream_t stream) {
    const int npl = (sd.N + 63) / 64;
    if (npl <= 2) return launch_one<2, POLY, ROWS, HV>(sd, in, out, B, ws, stream);
    if (npl <= 4) return launch_one<4, POLY, ROWS, HV>(sd, in, out, B, ws, stream);
    return launch_one<8, POLY, ROWS, HV>(sd, in, out, B, ws, stream);
}

template <bool ROWS, bool HV>
hipError_t launch_npl_wide(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, int blocks, const DenseWs &ws,
                           hipStream_t stream) {
    const int npl = (sd.N + 63) / 64;
    if (npl <= 2) return launch_one_wide<2, ROWS, HV>(sd, in, out, blocks, ws, stream);
    if (npl <= 4) return launch_one_wide<4, ROWS, HV>(sd, in, out, blocks, ws, stream);
    return launch_one_wide<8, ROWS, HV>(sd, in, out, blocks, ws, stream);
}

} // namespace

hipError_t launch_zspace(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, int B,
                         const DenseWs &ws, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    const bool poly = sd.form != MPCB_BOX_1D;
    if (in.height) {
        if (!sd.Spp) return hipErrorInvalidValue; // the height tables were not uploaded
        if (poly && in.rows_pqr) return launch_npl<true, true, true>(sd, in, out, B, ws, stream);
        if (poly) return launch_npl<true, false, true>(sd, in, out, B, ws, stream);
        return launch_npl<false, false, true>(sd, in, out, B, ws, stream);
    }
    if (poly && in.rows_pqr) return launch_npl<true, true, false>(sd, in, out, B, ws, stream);
    if (poly) return launch_npl<true, false, false>(sd, in, out, B, ws, stream);
    return launch_npl<false, false, false>(sd, in, out, B, ws, stream);
}

hipError_t launch_dispatch_order(const DenseWs &ws, int B, hipStream_t stream) {
    if (B <= 0 || !ws.order || !ws.order_record) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dispatch_order_kernel, dim3(1), dim3(ORDER_THREADS), 0, stream, ws.order_record, ws.order,
                       ws.count, B);
    return hipGetLastError();
}

hipError_t launch_zspace_wide(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, int blocks,
                              const DenseWs &ws, hipStream_t stream) {
    if (blocks <= 0 || sd.form == MPCB_BOX_1D || !ws.wide_list) return hipSuccess;
    if (in.height) {
        if (in.rows_pqr) return launch_npl_wide<true, true>(sd, in, out, blocks, ws, stream);
        return launch_npl_wide<false, true>(sd, in, out, blocks, ws, stream);
    }
    if (in.rows_pqr) return launch_npl_wide<true, false>(sd, in, out, blocks, ws, stream);
    return launch_npl_wide<false, false>(sd, in, out, blocks, ws, stream);
}

size_t zspace_wide_lds_bytes(const DevSetup &sd) {
    const int npl = (sd.N + 63) / 64;
    return npl <= 2 ? wave_lds_bytes<true, 2, true, false>(sd) : npl <= 4 ? wave_lds_bytes<true, 4, true, false>(sd)
                                                                   : wave_lds_bytes<true, 8, true, false>(sd);
}

} // namespace mpcb

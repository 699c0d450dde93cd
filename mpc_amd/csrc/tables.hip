// Table construction on the GPU: prediction matrices, condensed Hessian block, Cholesky factor,
// J = L^-T and the response matrices, for a LIST of parameter groups in one chain of launches
// (grid y = group).  What the reference rebuilds densely on every call
// (MPCPlanner::buildMatrix, MPCPlanner.cpp:101-156; the Hessian blocks :233, :523-559, :723-739;
// LLT and J inside solve_quadprog, eiquadprog.hpp:165-176).
//
// Arithmetic: double-double (an unevaluated sum hi + lo of two doubles, ~106 bits) built from
// error-free transformations on the fp64 vector unit.  The Hessian block reaches cond ~ 1e10
// (weight_R = 1e-6, or weight_Qs = 100), where a plain double factorisation leaves the tables
// good to ~1e-6 only -- the level of the parity bar itself; in double-double every table entry
// is a correctly rounded double for all practical purposes.  MFMA does not apply: the matrix
// cores accumulate in plain double and cannot return the rounding error of a product, and the
// work is a few hundred MFLOP per group, once per handle.
#include "tables.h"

#include <cstring>

#include "device_common.h"

// the error-free transformations below rely on every product and sum being rounded on its own
#pragma clang fp contract(off)

namespace mpcb {
namespace {

typedef double2 dd; // x: high part, y: low part

__device__ __forceinline__ dd mk(double a) { return make_double2(a, 0.0); }

__device__ __forceinline__ dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return make_double2(s, (a - (s - bb)) + (b - bb));
}
__device__ __forceinline__ dd quick_two_sum(double a, double b) {
    const double s = a + b;
    return make_double2(s, b - (s - a));
}
__device__ __forceinline__ dd two_prod(double a, double b) {
    const double p = a * b;
    return make_double2(p, __builtin_fma(a, b, -p));
}
__device__ __forceinline__ dd dd_add(dd a, dd b) {
    dd s = two_sum(a.x, b.x);
    const dd t = two_sum(a.y, b.y);
    s.y += t.x;
    s = quick_two_sum(s.x, s.y);
    s.y += t.y;
    return quick_two_sum(s.x, s.y);
}
__device__ __forceinline__ dd dd_neg(dd a) { return make_double2(-a.x, -a.y); }
__device__ __forceinline__ dd dd_sub(dd a, dd b) { return dd_add(a, dd_neg(b)); }
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
    dd p = two_prod(a.x, b.x);
    p.y += a.x * b.y + a.y * b.x;
    return quick_two_sum(p.x, p.y);
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b) {
    dd p = two_prod(a.x, b);
    p.y += a.y * b;
    return quick_two_sum(p.x, p.y);
}
__device__ __forceinline__ dd dd_half(dd a) { return make_double2(0.5 * a.x, 0.5 * a.y); }
__device__ __forceinline__ dd dd_div(dd a, dd b) {
    const double q1 = a.x / b.x;
    dd r = dd_sub(a, dd_mul_d(b, q1));
    const double q2 = r.x / b.x;
    r = dd_sub(r, dd_mul_d(b, q2));
    const double q3 = r.x / b.x;
    return dd_add(quick_two_sum(q1, q2), mk(q3));
}
// sqrt(a) = a x + (a - (a x)^2) x / 2 with x ~ 1/sqrt(a)
__device__ __forceinline__ dd dd_sqrt(dd a) {
    const double x = 1.0 / sqrt(a.x), ax = a.x * x;
    const dd err = dd_sub(a, two_prod(ax, ax));
    return two_sum(ax, err.x * (x * 0.5));
}
__device__ __forceinline__ double dd_round(dd a) { return a.x + a.y; }

enum { T_Z = 0, T_XP = 1, T_XV = 2, T_XA = 3 };

struct TabGroup {
    int N, form, nphase, height, ncol, vel;
    double Ts, Qs, zz;      // zz: factor of the Zu' Zu term of G1 (Qs, 1 for solveQP, else 0)
    dd diag, c;
    // double-double workspace
    dd *col;    // [4][N]     first columns of the Toeplitz matrices Zu, Xpu, Xvu, Xau
    dd *sx;     // [4][N][3]  state matrices Zx, Xpx, Xvx, Xax
    dd *w;      // [N]        velocity weights
    dd *G;      // [N][N]     Hessian block, then (lower triangle) its Cholesky factor L
    dd *Gt;     // [N][N]     L' (upper triangle): rows of it are columns of L, contiguous
    dd *Gm;     // [N][4]     g0 = Gm (pos, vel, acc, target speed)
    dd *gdiag;  // [N]        diagonal of G1
    dd *invd;   // [N]        1 / L[i][i]
    dd *X;      // [N][ncol]  right-hand sides, then solutions
    dd *P;      // [4][N][ncol] Toeplitz products with the solutions
    dd *Spd;    // [nphase][N]  per-phase ZMP response before rounding (COUPLED_REF)
    const int *knot_phase;
    int *flag;  // set when G1 is not positive definite
    // finished tables (double)
    double *S, *Wt, *Sc, *Sv, *Kx, *Kz, *Kc, *Kv, *J0, *zu_col, *xp_col, *xv_col, *zx, *xpx, *xvx;
    double *Sp, *Wp, *F2, *F3, *Spp, *Sps, *Saa, *Wtp, *Wta, *Kzp, *Kza;
    double *scal; // [18] F[16], trace(G1), trace(L^-T)
};

// column ranges of X
__device__ __forceinline__ int col_K(const TabGroup &g) { return g.N; }          // 4: G1^-1 Gm
__device__ __forceinline__ int col_J(const TabGroup &g) { return g.N + 4; }      // N: L^-T
__device__ __forceinline__ int col_P(const TabGroup &g) { return 2 * g.N + 4; }  // N: G1^-1 Xpu'
__device__ __forceinline__ int col_A(const TabGroup &g) { return 3 * g.N + 4; }  // N: G1^-1 Xau'

// Closed forms of buildMatrix's iteration (MPCPlanner.cpp:113-153): with tau = (i+1) Ts,
// A^(i+1) = [[1, tau, tau^2/2], [0, 1, tau], [0, 0, 1]], so row i of the state matrix is C A^(i+1)
// and sub-diagonal k of the input matrix is C A^k B, B = (Ts^3/6, Ts^2/2, Ts)'.
__global__ void __launch_bounds__(64) predict_kernel(const TabGroup *groups) {
    const TabGroup g = groups[blockIdx.y];
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= g.N) return;
    const dd T = mk(g.Ts);
    const dd tau = two_prod(g.Ts, (double)(i + 1)), s = two_prod(g.Ts, (double)i);
    const dd tau2h = dd_half(dd_mul(tau, tau));
    const dd T2h = dd_half(dd_mul(T, T));
    const dd T3s = dd_div(dd_mul(dd_mul(T, T), T), mk(6.0));
    // A^k B with s = k Ts
    const dd b0 = dd_add(dd_add(T3s, dd_mul(s, T2h)), dd_mul(dd_half(dd_mul(s, s)), T));
    const dd b1 = dd_add(T2h, dd_mul(s, T));
    const dd b2 = T;
    for (int t = 0; t < 4; t++) {
        // output rows: Cz = (1, 0, -c), Cx = (1, 0, 0), Cv = (0, 1, 0), Ca = (0, 0, 1) (MPCPlanner.cpp:58-63)
        const dd C0 = mk(t <= T_XP ? 1.0 : 0.0), C1 = mk(t == T_XV ? 1.0 : 0.0);
        const dd C2 = t == T_Z ? dd_neg(g.c) : mk(t == T_XA ? 1.0 : 0.0);
        dd *sx = g.sx + ((size_t)t * g.N + i) * 3;
        sx[0] = C0;
        sx[1] = dd_add(dd_mul(C0, tau), C1);
        sx[2] = dd_add(dd_add(dd_mul(C0, tau2h), dd_mul(C1, tau)), C2);
        g.col[(size_t)t * g.N + i] = dd_add(dd_add(dd_mul(C0, b0), dd_mul(C1, b1)), dd_mul(C2, b2));
    }
}

// G1 = diag + Xvu' W Xvu + zz Zu' Zu (lower triangle) and Gm; one workgroup per row.
__global__ void __launch_bounds__(256) hessian_kernel(const TabGroup *groups) {
    const TabGroup g = groups[blockIdx.y];
    const int N = g.N, i = blockIdx.x;
    if (i >= N) return;
    const dd *z = g.col + (size_t)T_Z * N, *xv = g.col + (size_t)T_XV * N;
    const dd *sxz = g.sx + (size_t)T_Z * N * 3, *sxv = g.sx + (size_t)T_XV * N * 3;
    for (int j = threadIdx.x; j < N + 4; j += 256) {
        if (j < N) {
            if (j > i) continue;
            dd acc = mk(0.0);
            if (g.vel)
                for (int k = i; k < N; k++) acc = dd_add(acc, dd_mul(dd_mul(g.w[k], xv[k - i]), xv[k - j]));
            if (g.zz != 0.0) {
                dd a2 = mk(0.0);
                for (int k = i; k < N; k++) a2 = dd_add(a2, dd_mul(z[k - i], z[k - j]));
                acc = dd_add(acc, dd_mul_d(a2, g.zz));
            }
            if (i == j) {
                acc = dd_add(acc, g.diag);
                g.gdiag[i] = acc;
            }
            g.G[(size_t)i * N + j] = acc;
        } else {
            const int a = j - N;
            dd acc = mk(0.0);
            if (g.vel)
                for (int k = i; k < N; k++) {
                    const dd wx = dd_mul(g.w[k], xv[k - i]);
                    acc = a < 3 ? dd_add(acc, dd_mul(wx, sxv[(size_t)k * 3 + a])) : dd_sub(acc, wx);
                }
            if (g.zz != 0.0 && a < 3) {
                dd a2 = mk(0.0);
                for (int k = i; k < N; k++) a2 = dd_add(a2, dd_mul(z[k - i], sxz[(size_t)k * 3 + a]));
                acc = dd_add(acc, dd_mul_d(a2, g.zz));
            }
            g.Gm[(size_t)i * 4 + a] = acc;
        }
    }
}

// In-place lower Cholesky of G (right-looking, one workgroup per group; the factor stays in
// global memory, which for these sizes means L2): per column a scale and a rank-1 update of the
// trailing triangle, a wave per row.
constexpr int CHOL_THREADS = 1024;
__global__ void __launch_bounds__(CHOL_THREADS) cholesky_kernel(const TabGroup *groups) {
    extern __shared__ dd lcol[];
    const TabGroup g = groups[blockIdx.x];
    const int N = g.N, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int j = 0; j < N; j++) {
        const dd ajj = g.G[(size_t)j * N + j];
        if (!(ajj.x > 0.0)) { // the same value in every thread
            if (tid == 0) *g.flag = 1;
            return;
        }
        const dd d = dd_sqrt(ajj), inv = dd_div(mk(1.0), d);
        for (int i = j + 1 + tid; i < N; i += CHOL_THREADS) {
            const dd l = dd_mul(g.G[(size_t)i * N + j], inv);
            g.G[(size_t)i * N + j] = l;
            g.Gt[(size_t)j * N + i] = l;
            lcol[i] = l;
        }
        __syncthreads();
        if (tid == 0) {
            g.G[(size_t)j * N + j] = d;
            g.invd[j] = inv;
        }
        for (int i = j + 1 + wave; i < N; i += CHOL_THREADS / 64) {
            const dd li = lcol[i];
            for (int k = j + 1 + lane; k <= i; k += 64) {
                dd *p = g.G + (size_t)i * N + k;
                *p = dd_sub(*p, dd_mul(li, lcol[k]));
            }
        }
        __syncthreads();
    }
}

// Sum of a double-double over the wave (result wave-uniform): the butterfly of wave_sum with
// double-double additions.
__device__ __forceinline__ dd wave_sum_dd(dd v) {
#define DD_STEP_(CTRL) v = dd_add(v, make_double2(dpp_perm_f64<CTRL>(v.x), dpp_perm_f64<CTRL>(v.y)))
    DD_STEP_(0xB1);
    DD_STEP_(0x4E);
    DD_STEP_(0x141);
    DD_STEP_(0x140);
#undef DD_STEP_
    v = dd_add(v, make_double2(dpp_f64<0x142, 0xA>(0.0, v.x), dpp_f64<0x142, 0xA>(0.0, v.y)));
    v = dd_add(v, make_double2(dpp_f64<0x143, 0xC>(0.0, v.x), dpp_f64<0x143, 0xC>(0.0, v.y)));
    return make_double2(lane63(v.x), lane63(v.y));
}

// Forward and backward substitution with L for every right-hand-side column: ONE WAVE PER COLUMN.
// Lane l keeps the entries l, l + 64, ... of the column in registers; row i's inner product with
// the entries already solved is a lane-parallel partial sum (a coalesced read of row i of L, or of
// L' for the backward pass) and one wave reduction -- instead of a serial chain of i dependent
// double-double operations in one thread.
constexpr int SOLVE_SLOTS = 8; // horizons up to 512
__global__ void __launch_bounds__(64) solve_kernel(const TabGroup *groups) {
    const TabGroup g = groups[blockIdx.y];
    const int N = g.N, c = blockIdx.x, lane = threadIdx.x;
    if (c >= g.ncol || *g.flag) return;
    // what the column starts from: a column of Zu', of Gm, of the identity, of Xpu' or of Xau'
    int kind, cc = c;
    const dd *tcol = nullptr;
    if (c < N) {
        kind = 0;
        tcol = g.col + (size_t)T_Z * N;
    } else if (c < col_J(g)) {
        kind = 1;
        cc = c - N;
    } else if (c < col_P(g)) {
        kind = 2;
        cc = c - col_J(g);
    } else {
        kind = 0;
        const bool pos = c < col_A(g);
        cc = c - (pos ? col_P(g) : col_A(g));
        tcol = g.col + (size_t)(pos ? T_XP : T_XA) * N;
    }
    auto rhs = [&](int i) -> dd { // entry i of the right-hand side (any lane may ask for any entry)
        if (kind == 0) return cc >= i ? tcol[cc - i] : mk(0.0); // T'[i][cc] = tcol[cc - i] for cc >= i
        if (kind == 1) return g.Gm[(size_t)i * 4 + cc];
        return mk(i == cc ? 1.0 : 0.0);
    };
    dd xs[SOLVE_SLOTS];
#pragma unroll
    for (int j = 0; j < SOLVE_SLOTS; j++) {
        const int k = lane + 64 * j;
        xs[j] = k < N ? rhs(k) : mk(0.0);
    }
    if (kind != 2) { // (L^-T e_c needs the backward pass only)
        for (int i = 0; i < N; i++) {
            const dd *Li = g.G + (size_t)i * N;
            dd part = mk(0.0);
#pragma unroll
            for (int j = 0; j < SOLVE_SLOTS; j++) {
                const int k = lane + 64 * j;
                if (64 * j < i && k < i) part = dd_add(part, dd_mul(Li[k], xs[j]));
            }
            const dd tot = wave_sum_dd(part);
            const dd xi = dd_mul(dd_sub(rhs(i), tot), g.invd[i]);
#pragma unroll
            for (int j = 0; j < SOLVE_SLOTS; j++)
                if (lane + 64 * j == i) xs[j] = xi;
        }
    }
    for (int i = N - 1; i >= 0; i--) {
        const dd *Lti = g.Gt + (size_t)i * N; // L[k][i], k > i
        dd part = mk(0.0), yi = mk(0.0);
#pragma unroll
        for (int j = 0; j < SOLVE_SLOTS; j++) {
            const int k = lane + 64 * j;
            if (64 * j + 63 > i && k > i && k < N) part = dd_add(part, dd_mul(Lti[k], xs[j]));
            if (k == i) yi = xs[j];
        }
        // (the entry being solved for sits in one lane: it joins the sum with the opposite sign)
        const dd tot = wave_sum_dd(dd_sub(yi, part));
        const dd xi = dd_mul(tot, g.invd[i]);
#pragma unroll
        for (int j = 0; j < SOLVE_SLOTS; j++)
            if (lane + 64 * j == i) xs[j] = xi;
    }
#pragma unroll
    for (int j = 0; j < SOLVE_SLOTS; j++) {
        const int k = lane + 64 * j;
        if (k < N) g.X[(size_t)k * g.ncol + c] = xs[j];
    }
}

// P[t][i][c] = sum_{k <= i} col_t[i - k] X[k][c]: the Toeplitz matrices applied to the solutions.
__global__ void __launch_bounds__(256) product_kernel(const TabGroup *groups) {
    const TabGroup g = groups[blockIdx.y];
    const int N = g.N, i = blockIdx.x;
    if (i >= N || *g.flag) return;
    const size_t ld = (size_t)g.ncol, plane = (size_t)N * ld;
    for (int c = threadIdx.x; c < g.ncol; c += 256) {
        // which products the column needs: W -> Zu, Xpu, Xvu; Kx -> all four; the height columns -> Xpu, Xau
        int mask;
        if (c < N) mask = 0x7;
        else if (c < col_J(g)) mask = 0xF;
        else if (c < col_P(g)) continue;
        else mask = 0xA;
        dd acc[4] = {mk(0.0), mk(0.0), mk(0.0), mk(0.0)};
        for (int k = 0; k <= i; k++) {
            const dd x = g.X[(size_t)k * ld + c];
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (mask >> t & 1) acc[t] = dd_add(acc[t], dd_mul(g.col[(size_t)t * N + (i - k)], x));
        }
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (mask >> t & 1) g.P[t * plane + (size_t)i * ld + c] = acc[t];
    }
}

__device__ __forceinline__ dd sym2(const dd *M, size_t ld, int i, int j) {
    return dd_half(dd_add(M[(size_t)i * ld + j], M[(size_t)j * ld + i]));
}

// Rounds the N x N tables to double in the layouts the solve kernels read (row k = knot k).
__global__ void __launch_bounds__(256) square_tables_kernel(const TabGroup *groups) {
    const TabGroup g = groups[blockIdx.y];
    const int N = g.N, i = blockIdx.x;
    if (i >= N || *g.flag) return;
    const size_t ld = (size_t)g.ncol, plane = (size_t)N * ld;
    const dd *Pz = g.P + T_Z * plane, *Pp = g.P + T_XP * plane, *Pv = g.P + T_XV * plane, *Pa = g.P + T_XA * plane;
    for (int j = threadIdx.x; j < N; j += 256) {
        const size_t o = (size_t)i * N + j;
        // S = Zu G1^-1 Zu' is symmetric in exact arithmetic: the rounded table is exactly symmetric
        g.S[o] = dd_round(sym2(Pz, ld, i, j));
        g.Wt[o] = dd_round(g.X[(size_t)j * ld + i]);
        g.Sc[o] = dd_round(Pp[(size_t)j * ld + i]);
        g.Sv[o] = dd_round(Pv[(size_t)j * ld + i]);
        g.J0[o] = dd_round(g.X[(size_t)j * ld + col_J(g) + i]); // column-major: element (row j, column i)
        if (g.height) {
            const int cp = col_P(g), ca = col_A(g);
            g.Spp[o] = dd_round(sym2(Pp + cp, ld, i, j));
            g.Saa[o] = dd_round(sym2(Pa + ca, ld, i, j));
            g.Sps[o] = dd_round(dd_half(dd_add(dd_add(Pp[(size_t)i * ld + ca + j], Pa[(size_t)i * ld + cp + j]),
                                               dd_add(Pp[(size_t)j * ld + ca + i], Pa[(size_t)j * ld + cp + i]))));
            g.Wtp[o] = dd_round(g.X[(size_t)j * ld + cp + i]);
            g.Wta[o] = dd_round(g.X[(size_t)j * ld + ca + i]);
        }
    }
}

// N x 4 unconstrained responses, the rounded prediction matrices and the scalars.
__global__ void __launch_bounds__(256) small_tables_kernel(const TabGroup *groups) {
    const TabGroup g = groups[blockIdx.x];
    const int N = g.N, tid = threadIdx.x;
    const size_t ld = (size_t)g.ncol, plane = (size_t)N * ld;
    const int ck = col_K(g);
    for (int i = tid; i < N; i += 256) {
        g.zu_col[i] = dd_round(g.col[(size_t)T_Z * N + i]);
        g.xp_col[i] = dd_round(g.col[(size_t)T_XP * N + i]);
        g.xv_col[i] = dd_round(g.col[(size_t)T_XV * N + i]);
        for (int a = 0; a < 3; a++) {
            g.zx[i * 3 + a] = dd_round(g.sx[((size_t)T_Z * N + i) * 3 + a]);
            g.xpx[i * 3 + a] = dd_round(g.sx[((size_t)T_XP * N + i) * 3 + a]);
            g.xvx[i * 3 + a] = dd_round(g.sx[((size_t)T_XV * N + i) * 3 + a]);
        }
        if (*g.flag) continue;
        for (int a = 0; a < 4; a++) {
            // X holds G1^-1 Gm = -Kx; K* = (Toeplitz matrix) Kx + (state matrix)
            g.Kx[i * 4 + a] = -dd_round(g.X[(size_t)i * ld + ck + a]);
            dd k[4];
            for (int t = 0; t < 4; t++) {
                k[t] = dd_neg(g.P[t * plane + (size_t)i * ld + ck + a]);
                if (a < 3) k[t] = dd_add(k[t], g.sx[((size_t)t * N + i) * 3 + a]);
            }
            g.Kz[i * 4 + a] = dd_round(k[T_Z]);
            g.Kc[i * 4 + a] = dd_round(k[T_XP]);
            g.Kv[i * 4 + a] = dd_round(k[T_XV]);
            if (g.height) {
                g.Kzp[i * 4 + a] = dd_round(k[T_XP]);
                g.Kza[i * 4 + a] = dd_round(k[T_XA]);
            }
        }
    }
    if (*g.flag) return;
    if (tid < 16) { // f_unc = 1/2 g0 . x_unc = w' (1/2 Gm' Kx) w
        const int a = tid >> 2, b = tid & 3;
        dd s = mk(0.0);
        for (int i = 0; i < N; i++) s = dd_sub(s, dd_mul(g.Gm[(size_t)i * 4 + a], g.X[(size_t)i * ld + ck + b]));
        g.scal[tid] = dd_round(dd_half(s));
    } else if (tid == 16 || tid == 17) {
        const dd *v = tid == 16 ? g.gdiag : g.invd;
        dd s = mk(0.0);
        for (int i = 0; i < N; i++) s = dd_add(s, v[i]);
        g.scal[tid] = dd_round(s);
    }
}

// COUPLED_REF: the reference of computeCentroid (MPCPlanner.cpp:1106-1130) is constant over a
// footstep phase, so its effect is a sum over phases of (value on the phase) x (response summed
// over the phase's knots).  g0 gains -Qs Zu' zref (:731-735).  One workgroup per phase.
__global__ void __launch_bounds__(256) phase_tables_kernel(const TabGroup *groups) {
    const TabGroup g = groups[blockIdx.y];
    const int N = g.N, P = g.nphase, ph = blockIdx.x, tid = threadIdx.x;
    if (g.form != MPCB_COUPLED_REF || ph >= P || *g.flag) return;
    const size_t ld = (size_t)g.ncol, plane = (size_t)N * ld;
    const dd *Pz = g.P + T_Z * plane;
    for (int i = tid; i < N; i += 256) {
        dd s = mk(0.0), w = mk(0.0);
        for (int k = 0; k < N; k++)
            if (g.knot_phase[k] == ph) {
                s = dd_add(s, sym2(Pz, ld, k, i));
                w = dd_add(w, g.X[(size_t)i * ld + k]);
            }
        s = dd_mul_d(s, g.Qs);
        g.Spd[(size_t)ph * N + i] = s;
        g.Sp[(size_t)ph * N + i] = dd_round(s);
        g.Wp[(size_t)ph * N + i] = dd_round(dd_mul_d(w, g.Qs));
    }
    if (tid < 4) { // F2[a][ph] = -Qs sum_{k in ph} (Zu Kx)[k][a], and P holds -(Zu Kx)
        dd s = mk(0.0);
        for (int k = 0; k < N; k++)
            if (g.knot_phase[k] == ph) s = dd_add(s, Pz[(size_t)k * ld + col_K(g) + tid]);
        g.F2[(size_t)tid * P + ph] = dd_round(dd_mul_d(s, g.Qs));
    }
    __syncthreads();
    // F3[ph][ph2] = -Qs^2/2 sum_{k in ph, k2 in ph2} S[k][k2] = -Qs/2 sum_{k2 in ph2} Spd[ph][k2]
    for (int p2 = tid; p2 < P; p2 += 256) {
        dd s = mk(0.0);
        for (int k = 0; k < N; k++)
            if (g.knot_phase[k] == p2) s = dd_add(s, g.Spd[(size_t)ph * N + k]);
        g.F3[(size_t)ph * P + p2] = -dd_round(dd_half(dd_mul_d(s, g.Qs)));
    }
}

__global__ void __launch_bounds__(256) knot_reference_kernel(int N, double Qs, const double *S, const double *Wt,
                                                             const double *Kz, const double *zx, double *Sp,
                                                             double *Wp, double *F2, double *F3) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, NN = (size_t)N * N;
    if (i < NN) {
        Sp[i] = Qs * S[i];
        Wp[i] = Qs * Wt[i];
        F3[i] = -Qs * Qs * S[i] / 2;
    }
    if (i < (size_t)4 * N) {
        const size_t a = i / N, k = i % N;
        F2[i] = -Qs * (Kz[k * 4 + a] - (a < 3 ? zx[k * 3 + a] : 0.0));
    }
}

struct Carver {
    char *base = nullptr;
    size_t used = 0;
    template <class T> T *take(size_t n) {
        T *p = base ? reinterpret_cast<T *>(base + used) : nullptr;
        used += (n * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

// lays one group's workspace (dd) out of `ws` and its finished tables out of `out`
void carve(const TableSpec &s, TabGroup &g, Carver &ws, Carver &out) {
    const size_t N = (size_t)s.N, P = (size_t)s.nphase, nc = (size_t)g.ncol;
    g.col = ws.take<dd>(4 * N);
    g.sx = ws.take<dd>(4 * N * 3);
    g.w = ws.take<dd>(N);
    g.G = ws.take<dd>(N * N);
    g.Gt = ws.take<dd>(N * N);
    g.Gm = ws.take<dd>(N * 4);
    g.gdiag = ws.take<dd>(N);
    g.invd = ws.take<dd>(N);
    g.X = ws.take<dd>(N * nc);
    g.P = ws.take<dd>(4 * N * nc);
    g.Spd = ws.take<dd>(s.form == MPCB_COUPLED_REF ? P * N : 0);
    g.knot_phase = ws.take<int>(N);
    g.flag = ws.take<int>(1);
    g.S = out.take<double>(N * N);
    g.Wt = out.take<double>(N * N);
    g.Sc = out.take<double>(N * N);
    g.Sv = out.take<double>(N * N);
    g.J0 = out.take<double>(N * N);
    g.Kx = out.take<double>(N * 4);
    g.Kz = out.take<double>(N * 4);
    g.Kc = out.take<double>(N * 4);
    g.Kv = out.take<double>(N * 4);
    g.zu_col = out.take<double>(N);
    g.xp_col = out.take<double>(N);
    g.xv_col = out.take<double>(N);
    g.zx = out.take<double>(N * 3);
    g.xpx = out.take<double>(N * 3);
    g.xvx = out.take<double>(N * 3);
    g.scal = out.take<double>(18);
    if (s.form == MPCB_COUPLED_REF) {
        g.Sp = out.take<double>(P * N);
        g.Wp = out.take<double>(P * N);
        g.F2 = out.take<double>(4 * P);
        g.F3 = out.take<double>(P * P);
    }
    if (s.height_tables) {
        g.Spp = out.take<double>(N * N);
        g.Sps = out.take<double>(N * N);
        g.Saa = out.take<double>(N * N);
        g.Wtp = out.take<double>(N * N);
        g.Wta = out.take<double>(N * N);
        g.Kzp = out.take<double>(N * 4);
        g.Kza = out.take<double>(N * 4);
    }
}

} // namespace

#define TAB_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_);         \
            rc = MPCB_E_HIP;                                                           \
            goto done;                                                                 \
        }                                                                              \
    } while (0)

int build_tables_device(const TableSpec *specs, int n, TableSet *out, hipStream_t stream, std::string *err) {
    if (n <= 0) return 0;
    int rc = 0;
    std::vector<TabGroup> groups(n);
    std::vector<std::vector<double2>> w_stage(n); // host copies of the weights, alive until the final wait
    for (int q = 0; q < n; q++) out[q] = TableSet();
    void *ws_dev = nullptr;
    TabGroup *groups_dev = nullptr;
    int maxN = 0, maxcol = 0, maxP = 0;
    size_t ws_bytes = 0;
    // sizes first (null bases), then the same carving over the allocations
    for (int pass = 0; pass < 2; pass++) {
        Carver ws;
        ws.base = static_cast<char *>(ws_dev);
        for (int q = 0; q < n; q++) {
            const TableSpec &s = specs[q];
            TabGroup &g = groups[q];
            if (pass == 0) {
                std::memset(&g, 0, sizeof(g));
                g.N = s.N;
                g.form = s.form;
                g.nphase = s.nphase;
                g.height = s.height_tables ? 1 : 0;
                g.ncol = (s.height_tables ? 4 : 2) * s.N + 4;
                g.vel = s.w_hi.empty() ? 0 : 1;
                g.Ts = s.Ts;
                g.Qs = s.Qs;
                g.zz = s.form == MPCB_COUPLED_REF ? s.Qs : s.form == MPCB_UNCONSTRAINED ? 1.0 : 0.0;
                g.diag = make_double2(s.diag_hi, s.diag_lo);
                g.c = make_double2(s.c_hi, s.c_lo);
                maxN = s.N > maxN ? s.N : maxN;
                maxcol = g.ncol > maxcol ? g.ncol : maxcol;
                maxP = s.nphase > maxP ? s.nphase : maxP;
            }
            Carver o;
            o.base = pass == 0 ? nullptr : static_cast<char *>(out[q].block);
            carve(s, g, ws, o);
            if (pass == 0) TAB_TRY(hipMalloc(&out[q].block, o.used));
        }
        if (pass == 0) {
            ws_bytes = ws.used;
            TAB_TRY(hipMalloc(&ws_dev, ws_bytes));
        }
    }
    TAB_TRY(hipMalloc((void **)&groups_dev, sizeof(TabGroup) * n));
    for (int q = 0; q < n; q++) {
        const TableSpec &s = specs[q];
        TabGroup &g = groups[q];
        if (g.vel) {
            std::vector<double2> &w = w_stage[q];
            w.resize(s.N);
            for (int i = 0; i < s.N; i++) w[i] = make_double2(s.w_hi[i], s.w_lo[i]);
            TAB_TRY(hipMemcpyAsync(g.w, w.data(), sizeof(double2) * s.N, hipMemcpyHostToDevice, stream));
        }
        if (s.form == MPCB_COUPLED_REF)
            TAB_TRY(hipMemcpyAsync(const_cast<int *>(g.knot_phase), s.knot_phase.data(), sizeof(int) * s.N,
                                   hipMemcpyHostToDevice, stream));
        TAB_TRY(hipMemsetAsync(g.flag, 0, sizeof(int), stream));
    }
    TAB_TRY(hipMemcpyAsync(groups_dev, groups.data(), sizeof(TabGroup) * n, hipMemcpyHostToDevice, stream));
    {
        const unsigned un = (unsigned)n;
        hipLaunchKernelGGL(predict_kernel, dim3((maxN + 63) / 64, un), dim3(64), 0, stream, groups_dev);
        hipLaunchKernelGGL(hessian_kernel, dim3(maxN, un), dim3(256), 0, stream, groups_dev);
        hipLaunchKernelGGL(cholesky_kernel, dim3(un), dim3(CHOL_THREADS), sizeof(dd) * (size_t)maxN, stream, groups_dev);
        hipLaunchKernelGGL(solve_kernel, dim3(maxcol, un), dim3(64), 0, stream, groups_dev);
        hipLaunchKernelGGL(product_kernel, dim3(maxN, un), dim3(256), 0, stream, groups_dev);
        hipLaunchKernelGGL(square_tables_kernel, dim3(maxN, un), dim3(256), 0, stream, groups_dev);
        hipLaunchKernelGGL(small_tables_kernel, dim3(un), dim3(256), 0, stream, groups_dev);
        if (maxP > 0)
            hipLaunchKernelGGL(phase_tables_kernel, dim3(maxP, un), dim3(256), 0, stream, groups_dev);
        TAB_TRY(hipGetLastError());
    }
    for (int q = 0; q < n; q++) {
        const TabGroup &g = groups[q];
        double scal[18];
        int flag = 0;
        TAB_TRY(hipMemcpyAsync(scal, g.scal, sizeof(scal), hipMemcpyDeviceToHost, stream));
        TAB_TRY(hipMemcpyAsync(&flag, g.flag, sizeof(int), hipMemcpyDeviceToHost, stream));
        TAB_TRY(hipStreamSynchronize(stream));
        if (flag) {
            if (err) *err = "condensed Hessian is not positive definite";
            rc = MPCB_E_NUMERIC;
            goto done;
        }
        TableSet &t = out[q];
        std::memcpy(t.F, scal, sizeof(t.F));
        t.traceG1 = scal[16];
        t.traceJ1 = scal[17];
        t.S = g.S; t.Wt = g.Wt; t.Sc = g.Sc; t.Sv = g.Sv; t.J0 = g.J0;
        t.Kx = g.Kx; t.Kz = g.Kz; t.Kc = g.Kc; t.Kv = g.Kv;
        t.zu_col = g.zu_col; t.xp_col = g.xp_col; t.xv_col = g.xv_col;
        t.zx = g.zx; t.xpx = g.xpx; t.xvx = g.xvx;
        t.Sp = g.Sp; t.Wp = g.Wp; t.F2 = g.F2; t.F3 = g.F3;
        t.Spp = g.Spp; t.Sps = g.Sps; t.Saa = g.Saa; t.Wtp = g.Wtp; t.Wta = g.Wta;
        t.Kzp = g.Kzp; t.Kza = g.Kza;
    }
done:
    (void)hipStreamSynchronize(stream);
    if (ws_dev) (void)hipFree(ws_dev);
    if (groups_dev) (void)hipFree(groups_dev);
    if (rc)
        for (int q = 0; q < n; q++) {
            if (out[q].block) (void)hipFree(out[q].block);
            out[q] = TableSet();
        }
    return rc;
}

hipError_t launch_knot_reference_tables(int N, double Qs, const double *S, const double *Wt, const double *Kz,
                                        const double *zx, double *Sp, double *Wp, double *F2, double *F3,
                                        hipStream_t stream) {
    const size_t NN = (size_t)N * N;
    hipLaunchKernelGGL(knot_reference_kernel, dim3((unsigned)((NN + 255) / 256)), dim3(256), 0, stream, N, Qs, S, Wt,
                       Kz, zx, Sp, Wp, F2, F3);
    return hipGetLastError();
}

} // namespace mpcb

// Dense dual active-set kernel for SMALL box problems (n = N <= ~104 variables): one QP per
// workgroup, every matrix in LDS, built for the latency of one pass rather than throughput.
//
// Second stage behind the wave-per-QP kernel for MPCB_BOX_1D: the instances that kernel cannot
// certify (working sets beyond its capacity, numerically singular reduced systems -- in
// practice the numerically infeasible draws, whose working set grows to n over ~200 outer
// iterations) are solved here with the reference's method (eiquadprog.hpp:130-592): the full
// J = L^-T Q and a triangular factor of the working set, same pivot rules, tolerances and step
// formulas.  What differs from a transliteration is how the two factors are updated:
//
//  * add_constraint (eiquadprog.hpp:450-513) rotates the pairs (j-1, j), j = n-1 .. iq+1, so
//    that d2 = J2' n+ collapses into its first entry.  With the suffix sums
//    P_j = sum_{i>=j} J[:,i] d_i and nu_j = ||d[j..]|| the result of that whole sequence is
//        new J[:,iq] = P_iq / nu_iq                      ( = z / ||d2|| )
//        new J[:,j]  = (nu_j/nu_{j-1}) J[:,j-1] - d_{j-1}/(nu_{j-1} nu_j) P_j ,  j > iq
//    (up to the signs of columns, which cancel in every product the method forms).  P_iq is
//    the step direction z itself, so the pass that computes z in four column segments (one
//    per wave) leaves exactly the partial sums the update needs: the update is a second sweep
//    of independent fused multiply-adds, four waves on four column segments, with no
//    sequential rotation chain.
//  * Instead of R the kernel keeps T = R^-1 (packed upper triangle).  r = R^-1 d1 is then a
//    product, split over the four waves, instead of a back substitution chain of length iq;
//    an added constraint appends the column [-r; 1] / nu_iq; a dropped constraint qq needs
//    the plane rotations that move row qq of T (entries qq .. iq-1 =: g) into its last entry,
//    applied to the columns of T and of J (delete_constraint, eiquadprog.hpp:516-592, does
//    the same to R's rows and J's columns).  With prefix sums Q_j = sum_{i<=j} g_i col_i and
//    rho_j = ||g[qq..j]|| that sequence is again closed-form:
//        new col_j = g_{j+1}/(rho_{j+1} rho_j) Q_j - (rho_j/rho_{j+1}) col_{j+1}
//    and the freed direction is Q_{iq-1} / rho_{iq-1}.
//
// All four waves evaluate the scalar decisions (most violated row, step lengths, blocking
// constraint) redundantly from the same LDS partial sums, so every wave holds x, u, the
// working set and the decisions in registers and the workgroup meets at three barriers per
// outer iteration (one more per dropped constraint) -- the previous arrangement needed a
// barrier per reduction and serialised on one wave's back substitution.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_common.h"
#include "kernels.h"

namespace mpcb {
namespace {

// Diagnostic build only (-DMPCB_STAMPS): cycles per section into sd.dbg[inst*16 + section].
#ifdef MPCB_STAMPS
#define BSTAMP(i)                                          \
    do {                                                   \
        const long long t_ = __builtin_amdgcn_s_memtime(); \
        stamp_acc[i] += t_ - stamp_t;                      \
        stamp_t = t_;                                      \
    } while (0)
#else
#define BSTAMP(i) \
    do {          \
    } while (0)
#endif

constexpr int BT = 256, BW = 4, HALF = 64, PSTR = 128; // threads, waves, lanes, stride of the partial-sum arrays

__device__ __forceinline__ double bcast(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
}

// Inclusive suffix sum over the lanes (lane i receives v_i + ... + v_63), accumulated from the
// far end: shifts towards lower lanes inside each row of 16, then the totals of the rows above.
__device__ __forceinline__ double wave_suffix_scan(double v, int lane) {
    v += dpp_f64<0x101, 0xF>(0.0, v); // row_shl:1
    v += dpp_f64<0x102, 0xF>(0.0, v); // row_shl:2
    v += dpp_f64<0x104, 0xF>(0.0, v); // row_shl:4
    v += dpp_f64<0x108, 0xF>(0.0, v); // row_shl:8
    const double t1 = bcast(v, 16), t2 = bcast(v, 32), t3 = bcast(v, 48);
    const int row = lane >> 4;
    const double above = row == 0 ? (t3 + t2) + t1 : row == 1 ? t3 + t2 : row == 2 ? t3 : 0.0;
    return v + above;
}

__device__ __forceinline__ double shift_down(double a0, int lane) { // value of lane + 1
    return dpp_f64<0x130, 0xF>(0.0, a0);
}
__device__ __forceinline__ int shift_down(int a0, int lane) { return dpp_i32<0x130, 0xF>(0, a0); }

__global__ __launch_bounds__(BT) void box_dense_kernel(DevSetup sd, DevIn in, mpcb_outputs out, DenseWs ws) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = ufirst(tid >> 6);
    const int N = sd.N, n = N, m = 2 * N, ldj = n | 1;
    // ---- LDS -------------------------------------------------------------------
    // fixed-size arrays first, at compile-time offsets (fewer live address registers), then the
    // two matrices whose size depends on n
    double *xs = smem;                                // x by index (identical copies written by every wave)
    double *dsv = xs + PSTR;                          // d by index
    double *nus = dsv + PSTR;                         // suffix norms of d
    double *ca = nus + PSTR, *cb = ca + PSTR;         // update coefficients per column
    double *gs = cb + PSTR;                           // row qq of T
    double *ins = gs + PSTR;                          // reciprocal norms
    double *d1s = ins + PSTR;                         // d with the free part (positions >= iq) zeroed
    double *zc = d1s + 2 * PSTR;                      // Toeplitz column of Zu; zc[-PSTR .. -1] are zeros, so
                                                      // zc[k - j] needs no test for j > k
    double *dpart = zc + PSTR;                        // [BW][PSTR] partial sums of d = J' n+
    double *zpart = dpart + BW * PSTR;                // [BW][PSTR] partial sums of z = J2 d2
    double *rpart = zpart + BW * PSTR;                // [BW][PSTR] partial sums of r = T d1
    double *mpart = rpart + BW * PSTR;                // [BW][PSTR] partial sums of the ZMP
    double *J = mpart + BW * PSTR;                    // n x ldj, column-major
    double *Tp = J + n * ldj;                         // T = R^-1, packed: T(i,k) at k(k+1)/2 + i, i <= k

    // a lane owns the indices i0 = lane and i1 = lane + 64 of every length-n object
    const int i0 = lane, i1 = lane + HALF;
    const bool v0 = i0 < n, v1 = i1 < n;
    const int c0 = v0 ? i0 : n - 1, c1 = v1 ? i1 : n - 1;

    for (int k = tid; k < PSTR; k += BT) zc[k - PSTR] = 0.0;
    for (int k = tid; k < PSTR; k += BT) xs[k] = 0.0; // entries >= n stay zero
    const int count = *ws.count;
    for (int entry = blockIdx.x; entry < count; entry += gridDim.x) {
        const int inst = ws.list[entry];
#ifdef MPCB_STAMPS
        long long stamp_acc[16] = {0};
        long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
        // ---- inputs --------------------------------------------------------------
        // Cz = Cx - c Ca (MPCPlanner.cpp:58-63): with a height per instance the Toeplitz column of
        // Zu and the third column of Zx are shifted by c = height / gravity
        const bool hv = in.height != nullptr;
        const double ch = hv ? in.height[inst] * sd.inv_gravity : 0.0;
        for (int k = tid; k < PSTR; k += BT) zc[k] = k < N ? (hv ? ws.xp_col[k] - ch * sd.Ts : ws.zu_col[k]) : 0.0;
        double w4[4];
        {
            const double *px = in.x0 + (size_t)inst * 3;
            w4[0] = px[0];
            w4[1] = px[1];
            w4[2] = px[2];
            w4[3] = 0.0;
        }
        const double lo0 = v0 ? in.zmin[(size_t)inst * N + i0] : -DINF, hi0 = v0 ? in.zmax[(size_t)inst * N + i0] : DINF;
        const double lo1 = v1 ? in.zmin[(size_t)inst * N + i1] : -DINF, hi1 = v1 ? in.zmax[(size_t)inst * N + i1] : DINF;
        double zf0, zf1; // Zx x0 at the lane's knots
        {
            const double *a = (hv ? ws.xpx : ws.zx) + c0 * 3, *b = (hv ? ws.xpx : ws.zx) + c1 * 3;
            zf0 = a[0] * w4[0] + a[1] * w4[1] + (a[2] - ch) * w4[2];
            zf1 = b[0] * w4[0] + b[1] * w4[1] + (b[2] - ch) * w4[2];
        }
        // The unrolled sweeps run a few rows / columns past the end of their range with zero
        // coefficients; what they read there (J's padding row, the start of T) must be finite.
        for (int idx = tid; idx < n * (n + 1) / 2 + PSTR; idx += BT) Tp[idx] = 0.0;
        for (int col = tid; col < n; col += BT)
            for (int row = n; row < ldj; row++) J[col * ldj + row] = 0.0;
        // J = L^-T (eiquadprog.hpp:174-176), x = -G^-1 g0, f = 1/2 g0'x (:188-191)
        for (int idx = tid; idx < n * n; idx += BT) {
            const int col = idx / n, row = idx - col * n;
            J[col * ldj + row] = ws.J1[col * N + row];
        }
        double x0r, x1r;
        {
            const double *a = sd.Kx + c0 * 4, *b = sd.Kx + c1 * 4;
            x0r = a[0] * w4[0] + a[1] * w4[1] + a[2] * w4[2];
            x1r = b[0] * w4[0] + b[1] * w4[1] + b[2] * w4[2];
        }
        double f = 0.0;
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) f += sd.F[a * 4 + b] * w4[a] * w4[b];
        if (v0) xs[i0] = x0r;
        if (v1) xs[i1] = x1r;

        double u0 = 0.0, u1 = 0.0, uo0 = 0.0, uo1 = 0.0, xo0 = 0.0, xo1 = 0.0;
        int A0 = 0, A1 = 0, Ao0 = 0, Ao1 = 0;
        // bit 0 / 1: lower / upper bound row of knot i0, bit 2 / 3: of knot i1
        unsigned actm = 0, exclm = 0xf;
        const unsigned validm = (v0 ? 3u : 0u) | (v1 ? 12u : 0u);
        int iq = 0, iters = 0, passes = 0, status = MPCB_OK;
        double R_norm = 1.0;
        const double nn = (double)n, mm = (double)m;
        const double inv_weight_R = nn / sd.traceG1; // the Hessian is weight_R I (MPCPlanner.cpp:262)
        double falg = 2.0 * nn * nn;
        double zm0 = 0.0, zm1 = 0.0; // ZMP at the lane's knots
        bool zmp_current = false;

        // ZMP at the current x: zfree + Zu x, the triangular product cut into four bands of
        // columns (one per wave); ends with every wave holding the sums.  One barrier.
        auto eval_zmp = [&]() {
            // bands of a multiple of four columns; x is zero beyond n and zc beyond the diagonal
            const int band = 4 * ((N + 4 * BW - 1) / (4 * BW)), j0 = wave * band;
            double a0 = wave == 0 ? zf0 : 0.0, a1 = wave == 0 ? zf1 : 0.0;
            const int jend = min(j0 + band, (N + 3) & ~3);
            const double *z0p = zc + i0, *z1p = zc + i1;
            for (int j = j0; j < jend; j += 4) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const double xj = xs[j + e];
                    a0 = fma(z0p[-(j + e)], xj, a0);
                    a1 = fma(z1p[-(j + e)], xj, a1);
                }
            }
            mpart[wave * PSTR + i0] = a0;
            mpart[wave * PSTR + i1] = a1;
            __syncthreads();
            zm0 = ((mpart[i0] + mpart[PSTR + i0]) + mpart[2 * PSTR + i0]) + mpart[3 * PSTR + i0];
            zm1 = ((mpart[i1] + mpart[PSTR + i1]) + mpart[2 * PSTR + i1]) + mpart[3 * PSTR + i1];
        };
        // row values of the lane's four rows from the ZMP it holds (rows that do not exist: +inf)
        auto row_vals = [&](double &s00, double &s01, double &s10, double &s11) {
            s00 = zm0 - lo0; // row i0          (lower bound)
            s01 = hi0 - zm0; // row N + i0      (upper bound)
            s10 = zm1 - lo1;
            s11 = hi1 - zm1;
        };
        // Drop working-set position qq (delete_constraint, eiquadprog.hpp:516-592).
        auto drop_constraint = [&](int qq, double &d0, double &d1) {
            const int last = iq - 1; // columns qq .. last take part
            {
                const double c = (double)(iq - 1 - qq);
                falg += 6.0 * nn * c + 3.0 * c * (c - 1.0);
            }
            // the row of the dropped constraint leaves the active flags
            {
                const int src = qq & 63;
                const int l = __builtin_amdgcn_readlane(qq >= HALF ? A1 : A0, src);
                const int kk = l < N ? l : l - N, up = l < N ? 0 : 1;
                if (lane == (kk & 63)) actm &= ~(1u << (2 * (kk >> 6) + up));
            }
            if (qq < last) {
                // g = row qq of T, prefix norms rho_j, coefficients of the closed form
                const double g0 = (i0 >= qq && i0 <= last) ? Tp[i0 * (i0 + 1) / 2 + qq] : 0.0;
                const double g1 = (i1 >= qq && i1 <= last) ? Tp[c1 * (c1 + 1) / 2 + qq] : 0.0;
                const double p0 = wave_scan(g0 * g0);
                const double p1 = wave_scan(g1 * g1) + lane63(p0);
                double rho0 = 0.0, ir0 = 0.0, rho1 = 0.0, ir1 = 0.0;
                if (p0 > 0.0) sqrt_rsqrt(p0, rho0, ir0);
                if (p1 > 0.0) sqrt_rsqrt(p1, rho1, ir1);
                gs[i0] = g0;
                gs[i1] = g1;
                ins[i0] = ir0;
                ins[i1] = ir1;
                wsync();
                // column j (qq <= j < last): a_j = g_{j+1} / (rho_{j+1} rho_j), b_j = -rho_j / rho_{j+1};
                // column last (J only): a = 1 / rho_last, b = 0
                {
                    const double gn0 = gs[min(i0 + 1, PSTR - 1)], in0 = ins[min(i0 + 1, PSTR - 1)];
                    const double gn1 = gs[min(i1 + 1, PSTR - 1)], in1 = ins[min(i1 + 1, PSTR - 1)];
                    double a0c = 0.0, b0c = 0.0, a1c = 0.0, b1c = 0.0;
                    if (i0 >= qq && i0 < last) {
                        a0c = gn0 * (in0 * ir0);
                        b0c = -rho0 * in0;
                    } else if (i0 == last) {
                        a0c = ir0;
                    }
                    if (i1 >= qq && i1 < last) {
                        a1c = gn1 * (in1 * ir1);
                        b1c = -rho1 * in1;
                    } else if (i1 == last) {
                        a1c = ir1;
                    }
                    ca[i0] = a0c;
                    cb[i0] = b0c;
                    ca[i1] = a1c;
                    cb[i1] = b1c;
                    // d = J' n+ follows the columns of J: the same combination, applied to the
                    // entries the lanes hold (the next pass then needs no product with J)
                    const double q0 = wave_scan(g0 * d0);
                    const double q1 = wave_scan(g1 * d1) + lane63(q0);
                    const double dn0 = lane == 63 ? bcast(d1, 0) : shift_down(d0, lane);
                    const double dn1 = shift_down(d1, lane);
                    if (i0 >= qq && i0 <= last) d0 = fma(a0c, q0, b0c * dn0);
                    if (i1 >= qq && i1 <= last) d1 = fma(a1c, q1, b1c * dn1);
                }
                wsync();
                if (wave < 2) {
                    // rows of J: wave 0 rows 0..63, wave 1 rows 64..n-1
                    const int r = wave == 0 ? c0 : c1;
                    const bool rv = wave == 0 ? v0 : v1;
                    double *Jr = J + r;
                    double cj = Jr[qq * ldj], q = 0.0;
                    int j = qq;
                    for (; j + 4 <= last; j += 4) {
                        double cn[4], o[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) cn[e] = Jr[(j + e + 1) * ldj];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            q = fma(gs[j + e], cj, q);
                            o[e] = fma(ca[j + e], q, cb[j + e] * cn[e]);
                            cj = cn[e];
                        }
                        if (rv) {
#pragma unroll
                            for (int e = 0; e < 4; e++) Jr[(j + e) * ldj] = o[e];
                        }
                    }
                    for (; j < last; j++) {
                        const double cn = Jr[(j + 1) * ldj];
                        q = fma(gs[j], cj, q);
                        const double o = fma(ca[j], q, cb[j] * cn);
                        if (rv) Jr[j * ldj] = o;
                        cj = cn;
                    }
                    q = fma(gs[last], cj, q);
                    if (rv) Jr[last * ldj] = ca[last] * q;
                } else if (wave == 2) {
                    // rows of T (two per lane); row i > qq moves up to i - 1, row qq goes, and
                    // so does the last column
                    double cja = (i0 <= qq) ? Tp[qq * (qq + 1) / 2 + min(i0, qq)] : 0.0; // T(i0, qq)
                    double cjb = 0.0;                                                            // T(i1, qq), i1 > qq always or unused
                    if (i1 <= qq) cjb = Tp[qq * (qq + 1) / 2 + i1];
                    double qa = 0.0, qb = 0.0;
                    const int na = i0 > qq ? i0 - 1 : i0, nb = i1 > qq ? i1 - 1 : i1;
                    const bool wa = i0 != qq && i0 <= last, wb = i1 != qq && i1 <= last;
                    // (the lanes' second rows exist only once the working set is beyond 64 rows)
                    auto chain = [&](auto two) {
                        constexpr bool TWO = decltype(two)::value;
                        for (int jb = qq; jb < last; jb += 4) {
                            double cna[4], cnb[4], gv[4], av[4], bv4[4];
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int j = min(jb + e, last - 1);
                                const int cbase = (j + 1) * (j + 2) / 2;
                                const double ta = Tp[cbase + min(i0, j + 1)];
                                cna[e] = (i0 <= j + 1) ? ta : 0.0;
                                if constexpr (TWO) {
                                    const double tb = Tp[cbase + min(i1, j + 1)];
                                    cnb[e] = (i1 <= j + 1) ? tb : 0.0;
                                }
                                gv[e] = gs[j];
                                av[e] = ca[j];
                                bv4[e] = cb[j];
                            }
                            wsync(); // every lane has read columns jb + 1 .. jb + 4 before columns jb .. are overwritten
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int j = jb + e;
                                if (j < last) {
                                    const int obase = j * (j + 1) / 2;
                                    qa = fma(gv[e], cja, qa);
                                    const double oa = fma(av[e], qa, bv4[e] * cna[e]);
                                    cja = cna[e];
                                    if (wa && na <= j) Tp[obase + na] = oa;
                                    if constexpr (TWO) {
                                        qb = fma(gv[e], cjb, qb);
                                        const double ob = fma(av[e], qb, bv4[e] * cnb[e]);
                                        cjb = cnb[e];
                                        if (wb && nb <= j) Tp[obase + nb] = ob;
                                    }
                                }
                            }
                            wsync();
                        }
                    };
                    if (last >= HALF) chain(std::true_type{});
                    else chain(std::false_type{});
                }
            }
            // working-set arrays: positions above qq move down by one (registers, every wave)
            {
                const double un0 = shift_down(u0, lane), un1 = shift_down(u1, lane);
                const int an0 = shift_down(A0, lane), an1 = shift_down(A1, lane);
                const double ux = bcast(u1, 0);
                const int ax = __builtin_amdgcn_readlane(A1, 0);
                if (i0 >= qq) {
                    u0 = lane == 63 ? ux : un0;
                    A0 = lane == 63 ? ax : an0;
                }
                if (i1 >= qq) {
                    u1 = lane == 63 ? 0.0 : un1;
                    A1 = lane == 63 ? 0 : an1;
                }
            }
            iq--;
            __syncthreads(); // J and T as the next pass reads them
        };

        __syncthreads();
        BSTAMP(0);
        // Taking over from the first stage: its record holds the working set (in order) and the
        // multipliers at the top of the outer iteration it could not finish.  The rows are put
        // back one by one through the ordinary pass -- d, z, r and the update of J and T for an
        // added constraint -- with the step length forced to zero and no decisions taken, then x
        // follows from the multipliers (x = x_unc + G^-1 N u) and the method carries on with the
        // record's objective and counters.
        const int *wi = ws.warm_i ? ws.warm_i + (size_t)inst * WARM_STRIDE : nullptr;
        const double *wd = ws.warm_d ? ws.warm_d + (size_t)inst * WARM_STRIDE : nullptr;
        const int n_warm = wi ? ufirst(wi[WARM_N]) : 0;
        int replay = n_warm; // rows still to be put back
        if (n_warm > 0) R_norm = fmax(R_norm, wd[WARM_RNORM]);
        bool done = false;
        int ip = 0, kip = 0;
        double sgn = 1.0, ucand = 0.0, s_ip = 0.0;
        while (!done) { // step 1 (l1)
            double s00 = 0.0, s01 = 0.0, s10 = 0.0, s11 = 0.0;
            if (replay == 0) {
                iters++;
                if (iters > sd.iter_cap) {
                    status = MPCB_ITER_CAP;
                    break;
                }
                eval_zmp();
                BSTAMP(1);
                falg += 2.0 * mm * nn;
                row_vals(s00, s01, s10, s11);
                exclm = 0xf;
                // sum of infeasibilities (:255-261) and the most violated row among the rows that
                // are neither active nor excluded (:279-287), lowest row on ties
                double psi = 0.0;
                if (v0) psi += fmin(0.0, s00) + fmin(0.0, s01);
                if (v1) psi += fmin(0.0, s10) + fmin(0.0, s11);
                psi = wave_sum(psi);
                if (fabs(psi) <= sd.psi_thr) {
                    zmp_current = true;
                    break;
                }
                uo0 = u0;
                uo1 = u1;
                Ao0 = A0;
                Ao1 = A1;
                xo0 = x0r;
                xo1 = x1r;
            }
            double ssv = 0.0;
            bool restart;
            do { // step 2 (l2)
                restart = false;
                double bv = DINF;
                int brow = 0x7fffffff, aux = 0;
                if (replay > 0) {
                    ip = ufirst(wi[n_warm - replay]);
                    ssv = -1.0;
                    s_ip = 0.0;
                } else {
                    const unsigned ok = validm & ~actm & exclm;
                    auto consider = [&](double v, int row, unsigned bit) {
                        if ((ok & bit) && v < ssv && (v < bv || (v == bv && row < brow))) {
                            bv = v;
                            brow = row;
                        }
                    };
                    consider(s00, i0, 1u);
                    consider(s10, i1, 4u);
                    consider(s01, N + i0, 2u);
                    consider(s11, N + i1, 8u);
                    wave_argmin(bv, brow, aux);
                    if (bv < ssv) {
                        ssv = bv;
                        ip = brow;
                        s_ip = bv;
                    }
                }
                if (ssv >= 0.0) {
                    zmp_current = true; // x has not moved since the evaluation (or was restored)
                    done = true;
                    break;
                }
                kip = ip < N ? ip : ip - N;
                sgn = ip < N ? 1.0 : -1.0;
                ucand = replay > 0 ? wd[n_warm - replay] : 0.0; // (a row put back brings its multiplier)
                BSTAMP(2);
                double d0 = 0.0, d1 = 0.0;
                bool d_valid = false; // d = J' n+ is current (a drop rotates it along with J)
                for (;;) { // step 2a (l2a)
                    if (replay == 0) {
                        passes++;
                        if (passes > sd.iter_cap) {
                            status = MPCB_ITER_CAP;
                            done = true;
                            break;
                        }
                        falg += 2.0 * nn * nn + 2.0 * nn * (nn - iq) + (double)iq * iq;
                    }
                    if (!d_valid) {
                        // ---- d = J' n+ : rows 0 .. kip in four bands, a lane's two columns ------
                        {
                            // bands of a multiple of four rows; rows past kip meet zc's zero padding
                            const int band = 4 * ((kip + 4 * BW) / (4 * BW)), r0 = wave * band;
                            const int r1 = min(r0 + band, (kip + 4) & ~3);
                            const double *ja = J + c0 * ldj, *jb = J + c1 * ldj;
                            const double *zk = zc + kip;
                            double a0 = 0.0, a1 = 0.0;
                            for (int r = r0; r < r1; r += 4) {
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const double nr = zk[-(r + e)];
                                    a0 = fma(ja[r + e], nr, a0);
                                    a1 = fma(jb[r + e], nr, a1);
                                }
                            }
                            dpart[wave * PSTR + i0] = sgn * a0;
                            dpart[wave * PSTR + i1] = sgn * a1;
                        }
                        BSTAMP(3);
                        __syncthreads();
                        BSTAMP(4);
                        d0 = ((dpart[i0] + dpart[PSTR + i0]) + dpart[2 * PSTR + i0]) + dpart[3 * PSTR + i0];
                        d1 = ((dpart[i1] + dpart[PSTR + i1]) + dpart[2 * PSTR + i1]) + dpart[3 * PSTR + i1];
                        d0 = v0 ? d0 : 0.0;
                        d1 = v1 ? d1 : 0.0;
                        d_valid = true;
                    }
                    dsv[i0] = d0;
                    dsv[i1] = d1;
                    d1s[i0] = i0 < iq ? d0 : 0.0;
                    d1s[i1] = i1 < iq ? d1 : 0.0;
                    // last column of the free part with a nonzero d (beyond it z gets nothing)
                    int jt;
                    {
                        const unsigned long long nz0 = __ballot(i0 >= iq && d0 != 0.0);
                        const unsigned long long nz1 = __ballot(i1 >= iq && d1 != 0.0);
                        jt = nz1 ? 127 - __builtin_clzll(nz1) : nz0 ? 63 - __builtin_clzll(nz0) : iq - 1;
                    }
                    wsync();
                    // ---- z = J2 d2 over columns iq .. jt in four segments; r = T d1 likewise ----
                    // segments of a multiple of four columns; d is zero past jt
                    const int seg = 4 * ((jt - iq + 1 + 4 * BW - 1) / (4 * BW));
                    const int slo = iq + wave * seg, shi = min(slo + seg, iq + ((jt - iq + 4) & ~3));
                    double jb0 = 0.0, jb1 = 0.0; // column slo - 1, kept for the update of column slo
                    {
                        double a0 = 0.0, a1 = 0.0;
                        const double *j0p = J + c0, *j1p = J + c1;
                        for (int j = slo; j < shi; j += 4) {
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const double dj = dsv[j + e];
                                a0 = fma(j0p[(j + e) * ldj], dj, a0);
                                a1 = fma(j1p[(j + e) * ldj], dj, a1);
                            }
                        }
                        if (slo > iq && slo < shi) {
                            jb0 = j0p[(slo - 1) * ldj];
                            jb1 = j1p[(slo - 1) * ldj];
                        }
                        zpart[wave * PSTR + i0] = a0;
                        zpart[wave * PSTR + i1] = a1;
                        // r_i = sum_k T(i, k) d_k, k in this wave's quarter of [0, iq)
                        const int tseg = 4 * ((iq + 4 * BW - 1) / (4 * BW)), k0 = wave * tseg;
                        const int k1 = min(k0 + tseg, (iq + 3) & ~3);
                        double b0 = 0.0, b1 = 0.0;
                        if (iq > HALF) {
                            for (int k = k0; k < k1; k += 4) {
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int kk = k + e, base = kk * (kk + 1) / 2;
                                    const double dk = d1s[kk];
                                    const double t0 = Tp[base + min(i0, kk)], t1 = Tp[base + min(i1, kk)];
                                    b0 = fma(i0 <= kk ? t0 : 0.0, dk, b0);
                                    b1 = fma(i1 <= kk ? t1 : 0.0, dk, b1);
                                }
                            }
                        } else { // no working-set position in the lanes' second half yet
                            for (int k = k0; k < k1; k += 4) {
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int kk = k + e, base = kk * (kk + 1) / 2;
                                    const double t0 = Tp[base + min(i0, kk)];
                                    b0 = fma(i0 <= kk ? t0 : 0.0, d1s[kk], b0);
                                }
                            }
                        }
                        rpart[wave * PSTR + i0] = b0;
                        rpart[wave * PSTR + i1] = b1;
                    }
                    BSTAMP(5);
                    __syncthreads();
                    BSTAMP(6);
                    // ---- step lengths, in every wave ------------------------------------------
                    const double z0 = ((zpart[i0] + zpart[PSTR + i0]) + zpart[2 * PSTR + i0]) + zpart[3 * PSTR + i0];
                    const double z1 = ((zpart[i1] + zpart[PSTR + i1]) + zpart[2 * PSTR + i1]) + zpart[3 * PSTR + i1];
                    double r0v = ((rpart[i0] + rpart[PSTR + i0]) + rpart[2 * PSTR + i0]) + rpart[3 * PSTR + i0];
                    double r1v = ((rpart[i1] + rpart[PSTR + i1]) + rpart[2 * PSTR + i1]) + rpart[3 * PSTR + i1];
                    r0v = i0 < iq ? r0v : 0.0;
                    r1v = i1 < iq ? r1v : 0.0;
                    // z . n+ = ||d2||^2 (z = J2 d2 and J2' n+ = d2), and for the box problem, whose
                    // Hessian is weight_R I, z . z = ||d2||^2 / weight_R: both come from the suffix
                    // sums of d^2, which an addition needs anyway (accumulated from the far end like
                    // the reference's chain of hypots; a total minus a prefix would cancel)
                    const double e0 = (i0 >= iq && v0) ? d0 * d0 : 0.0, e1 = (i1 >= iq && v1) ? d1 * d1 : 0.0;
                    const double S1 = wave_suffix_scan(e1, lane);
                    const double S0 = wave_suffix_scan(e0, lane) + bcast(S1, 0);
                    const double zn = bcast(iq >= HALF ? S1 : S0, iq & 63);
                    const double zz = zn * inv_weight_R;
                    // step 2b: t1 = min u_k / r_k over r_k > 0, lowest position on ties (:322-334)
                    double t1 = (i0 < iq && r0v > 0.0) ? u0 / r0v : DINF;
                    int kpos = i0, aux2 = 0;
                    {
                        const double tb = (i1 < iq && r1v > 0.0) ? u1 / r1v : DINF;
                        if (tb < t1) {
                            t1 = tb;
                            kpos = i1;
                        }
                    }
                    wave_argmin(t1, kpos, aux2);
                    const double t2 = fabs(zz) > EPS ? -s_ip / zn : DINF; // :336-339
                    const double t = replay > 0 ? 0.0 : fmin(t1, t2);
                    BSTAMP(7);
                    if (replay > 0) {
                        // nothing moves: straight to the update for an added constraint
                    } else if (t >= DINF) { // :350-356
                        status = MPCB_INFEASIBLE;
                        f = DINF;
                        done = true;
                        break;
                    }
                    if (replay == 0 && t2 >= DINF) { // dual step only, then drop (:358-373)
                        u0 -= t * r0v;
                        u1 -= t * r1v;
                        ucand += t;
                        drop_constraint(kpos, d0, d1);
                        BSTAMP(10);
                        continue;
                    }
                    // step in primal and dual space (:377-382)
                    if (replay == 0) {
                        x0r = fma(t, z0, x0r);
                        x1r = fma(t, z1, x1r);
                        if (v0) xs[i0] = x0r;
                        if (v1) xs[i1] = x1r;
                        f += t * zn * (0.5 * t + ucand);
                        u0 -= t * r0v;
                        u1 -= t * r1v;
                        ucand += t;
                    }
                    if (t == t2 || replay > 0) { // full step: add the constraint (:392-426, add_constraint :450-513)
                        if (replay == 0) falg += 6.0 * nn * (nn - iq - 1);
                        // suffix norms nu_j = ||d[j .. jt]|| and their reciprocals from one rsqrt chain:
                        // no division in the coefficients
                        double nu0 = 0.0, inu0 = 0.0, nu1 = 0.0, inu1 = 0.0;
                        if (S0 > 0.0) sqrt_rsqrt(S0, nu0, inu0);
                        if (S1 > 0.0) sqrt_rsqrt(S1, nu1, inu1);
                        nus[i0] = nu0;
                        nus[i1] = nu1;
                        ins[i0] = inu0;
                        ins[i1] = inu1;
                        wsync();
                        const double rho = ufirst(nus[iq]); // new diagonal entry of R
                        const double irho = ufirst(ins[iq]);
                        {
                            const double dm0 = dsv[max(i0 - 1, 0)], im0 = ins[max(i0 - 1, 0)];
                            const double dm1 = dsv[i1 - 1], im1 = ins[i1 - 1];
                            double a0c = 0.0, b0c = 0.0, a1c = 0.0, b1c = 0.0;
                            if (i0 == iq) a0c = inu0;
                            else if (i0 > iq && i0 <= jt) {
                                a0c = -dm0 * (im0 * inu0);
                                b0c = nu0 * im0;
                            }
                            if (i1 == iq) a1c = inu1;
                            else if (i1 > iq && i1 <= jt) {
                                a1c = -dm1 * (im1 * inu1);
                                b1c = nu1 * im1;
                            }
                            ca[i0] = a0c;
                            cb[i0] = b0c;
                            ca[i1] = a1c;
                            cb[i1] = b1c;
                        }
                        wsync();
                        BSTAMP(8);
                        // the update: this wave's column segment, from its top down; P starts at
                        // the partial sums of the segments above
                        {
                            double P0 = 0.0, P1 = 0.0;
                            for (int w2 = BW - 1; w2 > wave; w2--) {
                                P0 += zpart[w2 * PSTR + i0];
                                P1 += zpart[w2 * PSTR + i1];
                            }
                            double *j0p = J + c0, *j1p = J + c1;
                            if (slo < shi) {
                                double cur0 = j0p[(shi - 1) * ldj], cur1 = j1p[(shi - 1) * ldj];
                                for (int jtop = shi - 1; jtop >= slo; jtop -= 4) {
                                    // columns jtop .. jtop - 3 (the segment is a multiple of four long);
                                    // everything the group reads is loaded before its first store, or each
                                    // column would wait for the store of the one before
                                    double p0v[4], p1v[4], dv[4], av[4], bv4[4];
#pragma unroll
                                    for (int e = 0; e < 4; e++) {
                                        const int j = jtop - e;
                                        p0v[e] = j0p[(j - 1) * ldj];
                                        p1v[e] = j1p[(j - 1) * ldj];
                                        dv[e] = dsv[j];
                                        av[e] = ca[j];
                                        bv4[e] = cb[j];
                                    }
                                    if (jtop - 3 == slo) { // the column below the segment was read before the barrier
                                        p0v[3] = jb0;
                                        p1v[3] = jb1;
                                    }
#pragma unroll
                                    for (int e = 0; e < 4; e++) {
                                        const int j = jtop - e;
                                        P0 = fma(cur0, dv[e], P0);
                                        P1 = fma(cur1, dv[e], P1);
                                        const double o0 = fma(av[e], P0, bv4[e] * p0v[e]), o1 = fma(av[e], P1, bv4[e] * p1v[e]);
                                        if (j <= jt) { // columns past jt keep what they hold
                                            if (v0) j0p[j * ldj] = o0;
                                            if (v1) j1p[j * ldj] = o1;
                                        }
                                        cur0 = p0v[e];
                                        cur1 = p1v[e];
                                    }
                                }
                            }
                        }
                        if (replay == 0 && rho <= EPS * R_norm) {
                            // degenerate (:400-418): exclude ip, restore u, A and x of step 1,
                            // choose again
                            if (lane == (kip & 63)) exclm &= ~(1u << (2 * (kip >> 6) + (ip < N ? 0 : 1)));
                            if (i0 < iq) {
                                u0 = uo0;
                                A0 = Ao0;
                            }
                            if (i1 < iq) {
                                u1 = uo1;
                                A1 = Ao1;
                            }
                            x0r = xo0;
                            x1r = xo1;
                            if (v0) xs[i0] = x0r;
                            if (v1) xs[i1] = x1r;
                            actm = 0;
                            for (int a = 0; a < iq; a++) {
                                const int l = __builtin_amdgcn_readlane(a >= HALF ? A1 : A0, a & 63);
                                const int kk = l < N ? l : l - N;
                                if (lane == (kk & 63)) actm |= 1u << (2 * (kk >> 6) + (l < N ? 0 : 1));
                            }
                            __syncthreads(); // J as the next pass reads it
                            restart = true;
                            break;
                        }
                        R_norm = fmax(R_norm, rho);
                        // new column of T = [-r; 1] / rho  (wave 0 stores it)
                        if (wave == 0) {
                            const int base = iq * (iq + 1) / 2;
                            if (i0 < iq) Tp[base + i0] = -r0v * irho;
                            if (i1 < iq) Tp[base + i1] = -r1v * irho;
                            if (i0 == iq) Tp[base + i0] = irho;
                            if (i1 == iq) Tp[base + i1] = irho;
                        }
                        if (i0 == iq) {
                            u0 = ucand;
                            A0 = ip;
                        }
                        if (i1 == iq) {
                            u1 = ucand;
                            A1 = ip;
                        }
                        if (lane == (kip & 63)) actm |= 1u << (2 * (kip >> 6) + (ip < N ? 0 : 1));
                        iq++;
                        if (replay > 0) {
                            __syncthreads(); // J and T as the next row's pass reads them
                            if (--replay == 0) {
                                // x = x_unc + G^-1 N u with G = weight_R I: row a contributes
                                // +-u_a Zu[k_a, :] / weight_R (zc is zero left of the diagonal)
                                for (int a = 0; a < n_warm; a++) {
                                    const int l = __builtin_amdgcn_readlane(a >= HALF ? A1 : A0, a & 63);
                                    const double ua = bcast(a >= HALF ? u1 : u0, a & 63);
                                    const int ka = l < N ? l : l - N;
                                    const double ca_ = (l < N ? ua : -ua) * inv_weight_R;
                                    x0r = fma(ca_, zc[ka - i0], x0r);
                                    x1r = fma(ca_, zc[ka - i1], x1r);
                                }
                                if (v0) xs[i0] = x0r;
                                if (v1) xs[i1] = x1r;
                                f = wd[WARM_F];
                                falg = wd[WARM_FALG];
                                iters = wi[WARM_ITERS];
                                passes = wi[WARM_PASSES];
                            }
                        }
                        BSTAMP(9);
                        break; // goto l1 (its barrier publishes J and T)
                    }
                    // partial step (:434-446): drop the blocking constraint, re-evaluate row ip
                    drop_constraint(kpos, d0, d1);
                    BSTAMP(10);
                    {
                        const double t0 = zc[kip - min(i0, kip)], t1b = zc[kip - min(i1, kip)];
                        const double acc = wave_sum((i0 <= kip ? t0 * x0r : 0.0) + ((v1 && i1 <= kip) ? t1b * x1r : 0.0));
                        const double zk = bcast((kip >= HALF) ? zf1 : zf0, kip & 63) + acc;
                        const double lok = bcast((kip >= HALF) ? lo1 : lo0, kip & 63);
                        const double hik = bcast((kip >= HALF) ? hi1 : hi0, kip & 63);
                        s_ip = ip < N ? zk - lok : hik - zk;
                    }
                    BSTAMP(11);
                }
            } while (restart);
        }

        // ---- outputs ---------------------------------------------------------------
#ifdef MPCB_STAMPS
        BSTAMP(12);
        stamp_acc[13] = passes;
        stamp_acc[14] = iters;
        if (tid == 0 && sd.dbg)
            for (int i = 0; i < 16; i++) sd.dbg[(size_t)inst * 16 + i] = stamp_acc[i];
#endif
        const bool solved = status == MPCB_OK || status == MPCB_ITER_CAP;
        if (tid == 0) {
            if (out.cost) out.cost[inst] = f;
            if (out.status) out.status[inst] = status;
            if (out.iters) out.iters[inst] = iters;
            if (out.n_active) out.n_active[inst] = iq;
            if (out.alg_flops) out.alg_flops[inst] = falg;
        }
        if (wave == 0 && (out.active_idx || out.multipliers)) {
            for (int a = lane; a < out.active_stride; a += HALF) {
                int av = -1;
                double uv = 0.0;
                if (a < 2 * HALF) {
                    const int as = a >= HALF ? A1 : A0;
                    const double us = a >= HALF ? u1 : u0;
                    if (a < iq) {
                        av = as;
                        uv = us;
                    }
                }
                if (out.active_idx) out.active_idx[(size_t)inst * out.active_stride + a] = av;
                if (out.multipliers) out.multipliers[(size_t)inst * out.active_stride + a] = uv;
            }
        }
        if (!solved) {
            x0r = x1r = 0.0;
            if (v0) xs[i0] = 0.0;
            if (v1) xs[i1] = 0.0;
        }
        __syncthreads();
        if (!zmp_current || !solved) eval_zmp();
        else __syncthreads();
        if (wave == 0) {
            auto put = [&](int k, double xk, double zk, double lok, double hik) {
                const size_t o = (size_t)inst * N + k;
                if (out.jerk_x) out.jerk_x[o] = xk;
                if (out.zmp_x) out.zmp_x[o] = solved ? zk : 0.0;
                if (out.zmp_y) out.zmp_y[o] = 0.0;
                if (out.min_margin) out.min_margin[o] = solved ? fmin(zk - lok, hik - zk) : 0.0;
            };
            if (v0) put(i0, x0r, zm0, lo0, hi0);
            if (v1) put(i1, x1r, zm1, lo1, hi1);
        }
        if (out.com_x || out.com_y || out.com_xd || out.com_yd) {
            // computeCOMtrajectory (MPCPlanner.cpp:196-209): Sx x0 + Su jerk with the Toeplitz columns
            for (int k = tid; k < N; k += BT) {
                const size_t o = (size_t)inst * N + k;
                const double *pr = ws.xpx + k * 3, *vr = ws.xvx + k * 3;
                double cx = pr[0] * w4[0] + pr[1] * w4[1] + pr[2] * w4[2];
                double vx = vr[0] * w4[0] + vr[1] * w4[1] + vr[2] * w4[2];
                for (int j = 0; j <= k; j++) {
                    cx += ws.xp_col[k - j] * xs[j];
                    vx += ws.xv_col[k - j] * xs[j];
                }
                if (out.com_x) out.com_x[o] = solved ? cx : 0.0;
                if (out.com_y) out.com_y[o] = 0.0;
                if (out.com_xd) out.com_xd[o] = solved ? vx : 0.0;
                if (out.com_yd) out.com_yd[o] = 0.0;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) retire_counters(ws.count, gridDim.x, ws.host_count);
}

} // namespace

size_t box_dense_lds_bytes(const DevSetup &sd) {
    const size_t n = sd.N;
    const size_t doubles = n * (n | 1) + n * (n + 1) / 2 + PSTR + 10 * PSTR + 4 * BW * PSTR;
    return doubles * 8;
}

bool box_dense_applies(const DevSetup &sd) {
    return sd.form == MPCB_BOX_1D && sd.N <= 2 * HALF && box_dense_lds_bytes(sd) <= 160 * 1024 - 512;
}

hipError_t launch_box_dense(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, const DenseWs &ws,
                            int blocks, hipStream_t stream) {
    if (blocks <= 0) return hipSuccess;
    const size_t shmem = box_dense_lds_bytes(sd);
    const void *fn = reinterpret_cast<const void *>(box_dense_kernel);
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    void *args[] = {(void *)&sd, (void *)&in, (void *)&out, (void *)&ws};
    return hipLaunchKernel(fn, dim3(blocks), dim3(BT), args, shmem, stream);
}

} // namespace mpcb

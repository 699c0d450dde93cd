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
//
// With one wave per SIMD nothing hides a wait, so the sweeps over J and T are written for the
// LDS pipeline: reads are issued by hand (below) as plain 8- and 16-byte reads, two groups of
// four columns in flight, and each group is consumed when it has arrived.  In the sweeps along
// the rows of J a lane holds two adjacent rows (one 16-byte read per column), in the sweeps
// along its columns four consecutive rows are two 16-byte reads; J's leading dimension is
// chosen so that both are aligned and free of bank conflicts.  None of this touches the order
// of a single floating-point operation: a build is checked by the digest of every output of
// the decoupled benchmark batch (scripts/box_stage_ab.py), whose numerically infeasible draws
// follow 200-iteration paths that any change of rounding would re-route.
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

// Leading dimension of J: the smallest one >= n that is 2 modulo 4.  Even, so that two
// consecutive rows of a column are one aligned 16-byte read; and with 8 * ldj = 16 * odd the
// 16-byte reads of sixteen lanes that hold sixteen consecutive columns fall on distinct banks.
__host__ __device__ constexpr int box_ldj(int n) { return (((n + 1) >> 2) << 2) + 2; }

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

// LDS reads issued by hand.  Left to the compiler, two 8-byte reads whose addresses differ by a
// constant become one two-address read, which moves half the bytes per LDS cycle of the plain
// 8-byte read, and every group of reads is waited for in full before the next one is issued.
// The value of a read may be used only after lds_arrived() has been given it: the compiler
// takes the register for written when the read is issued, so nothing may touch it before --
// scripts/lds_hazard_check.py verifies exactly that on the generated code (it runs with the CPU
// tests), and values that live on into the next group are copied by arrived_copy().
__device__ __forceinline__ unsigned lds_off(const double *p) { return (unsigned)(unsigned long long)p; }
template <int OFF> __device__ __forceinline__ double lds8(unsigned a) {
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
// 16 bytes from a 16-byte aligned address.  (The pair stays one register tuple until it has
// arrived: taking it apart earlier would copy registers that are still being written.)
typedef double pair_t __attribute__((ext_vector_type(2)));
typedef double four_t[4];
struct quad_t {
    pair_t lo, hi;
    __device__ __forceinline__ double operator[](int e) const { return e == 0 ? lo.x : e == 1 ? lo.y : e == 2 ? hi.x : hi.y; }
};
template <int OFF> __device__ __forceinline__ pair_t lds16(unsigned a) {
    pair_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
// A value that is carried from one group of a sweep into the next is copied by hand once it has
// arrived: a copy the compiler places on its own may land between the read and lds_arrived().
__device__ __forceinline__ double arrived_copy(double v) {
    double r;
    asm volatile("v_mov_b64 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ pair_t arrived_copy(pair_t v) {
    pair_t r;
    asm volatile("v_mov_b64 %0, %2\n\tv_mov_b64 %1, %3" : "=&v"(r.x), "=&v"(r.y) : "v"(v.x), "v"(v.y));
    return r;
}
typedef pair_t pair2_t[2];
typedef pair_t pair4_t[4];
// LATER = the number of reads issued after the ones waited for (LDS reads complete in order)
template <int LATER = 0> __device__ __forceinline__ void lds_arrived(quad_t &a, four_t &b, four_t &c) {
    asm volatile("s_waitcnt lgkmcnt(%10)"
                 : "+v"(a.lo), "+v"(a.hi), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]),
                   "+v"(c[2]), "+v"(c[3])
                 : "n"(LATER)
                 : "memory");
}
template <int LATER = 0> __device__ __forceinline__ void lds_arrived(quad_t &a, four_t &b) {
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(a.lo), "+v"(a.hi), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "n"(LATER)
                 : "memory");
}
template <int LATER = 0> __device__ __forceinline__ void lds_arrived(four_t &a, pair2_t &b, pair2_t &c) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(c[0]), "+v"(c[1])
                 : "n"(LATER)
                 : "memory");
}
template <int LATER = 0> __device__ __forceinline__ void lds_arrived(four_t &a, pair4_t &b) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "n"(LATER)
                 : "memory");
}
template <int LATER = 0> __device__ __forceinline__ void lds_arrived(quad_t &a, four_t &b, pair4_t &c) {
    asm volatile("s_waitcnt lgkmcnt(%10)"
                 : "+v"(a.lo), "+v"(a.hi), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]),
                   "+v"(c[2]), "+v"(c[3])
                 : "n"(LATER)
                 : "memory");
}
template <int LATER = 0>
__device__ __forceinline__ void lds_arrived(quad_t &a, four_t &b, four_t &c, pair4_t &d) {
    asm volatile("s_waitcnt lgkmcnt(%14)"
                 : "+v"(a.lo), "+v"(a.hi), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]),
                   "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
                 : "n"(LATER)
                 : "memory");
}
template <int LATER = 0> __device__ __forceinline__ void lds_arrived(four_t &a, pair4_t &b, pair4_t &c) {
    asm volatile("s_waitcnt lgkmcnt(%12)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                 : "n"(LATER)
                 : "memory");
}
// A sweep in groups of four columns, two groups in flight: `issue` starts the reads of the next
// group into the arrays it is given, `use` consumes a group.  READS = reads per group.
template <int READS, class A, class B, class C, class Issue, class Use>
__device__ __forceinline__ void sweep3(int groups, Issue issue, Use use) {
    A a0, b0;
    B a1, b1;
    C a2, b2;
    for (; groups >= 2; groups -= 2) {
        issue(a0, a1, a2);
        issue(b0, b1, b2);
        lds_arrived<READS>(a0, a1, a2);
        use(a0, a1, a2);
        lds_arrived<0>(b0, b1, b2);
        use(b0, b1, b2);
    }
    if (groups > 0) {
        issue(a0, a1, a2);
        lds_arrived<0>(a0, a1, a2);
        use(a0, a1, a2);
    }
}
template <int READS, class A, class B, class C, class D, class Issue, class Use>
__device__ __forceinline__ void sweep4(int groups, Issue issue, Use use) {
    A a0, b0;
    B a1, b1;
    C a2, b2;
    D a3, b3;
    for (; groups >= 2; groups -= 2) {
        issue(a0, a1, a2, a3);
        issue(b0, b1, b2, b3);
        lds_arrived<READS>(a0, a1, a2, a3);
        use(a0, a1, a2, a3);
        lds_arrived<0>(b0, b1, b2, b3);
        use(b0, b1, b2, b3);
    }
    if (groups > 0) {
        issue(a0, a1, a2, a3);
        lds_arrived<0>(a0, a1, a2, a3);
        use(a0, a1, a2, a3);
    }
}
template <int READS, class A, class B, class Issue, class Use>
__device__ __forceinline__ void sweep2(int groups, Issue issue, Use use) {
    A a0, b0;
    B a1, b1;
    for (; groups >= 2; groups -= 2) {
        issue(a0, a1);
        issue(b0, b1);
        lds_arrived<READS>(a0, a1);
        use(a0, a1);
        lds_arrived<0>(b0, b1);
        use(b0, b1);
    }
    if (groups > 0) {
        issue(a0, a1);
        lds_arrived<0>(a0, a1);
        use(a0, a1);
    }
}

__device__ __forceinline__ double shift_down(double a0, int lane) { // value of lane + 1
    return dpp_f64<0x130, 0xF>(0.0, a0);
}
__device__ __forceinline__ int shift_down(int a0, int lane) { return dpp_i32<0x130, 0xF>(0, a0); }

__global__ __launch_bounds__(BT) void box_dense_kernel(DevSetup sd, DevIn in, mpcb_outputs out, DenseWs ws) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = ufirst(tid >> 6);
    const int N = sd.N, n = N, m = 2 * N, ldj = box_ldj(n);
    // ---- LDS -------------------------------------------------------------------
    // fixed-size arrays first, at compile-time offsets (fewer live address registers), then the
    // two matrices whose size depends on n
    double *xs = smem;                                // x by index (identical copies written by every wave)
    double *dsv = xs + PSTR;                          // d by index
    double *nus = dsv + PSTR;                         // suffix norms of d
    double *ab = nus + PSTR;                          // update coefficients per column: (a_j, b_j) at ab[2j], ab[2j+1]
    double *gs = ab + 2 * PSTR;                       // row qq of T
    double *ins = gs + PSTR;                          // reciprocal norms
    double *d1s = ins + PSTR;                         // d with the free part (positions >= iq) zeroed
    double *zc = d1s + 2 * PSTR;                      // Toeplitz column of Zu; zc[-PSTR .. -1] are zeros, so
                                                      // zc[k - j] needs no test for j > k
    double *side = zc + PSTR;                         // new row 63 of T while a drop is under way
    double *dpart = side + PSTR;                      // [BW][PSTR] partial sums of d = J' n+
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
        for (int idx = tid; idx < n * (n + 1) / 2 + 2 * PSTR; idx += BT) Tp[idx] = 0.0;
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
            // x[j .. j+3] from one address for the whole wave; the Toeplitz entries zc[i - j - e] of the
            // lane's knots i0 and i0 + 64 descend from zc[i - j]
            unsigned xa = lds_off(xs + j0), za = lds_off(zc + i0 - j0 - 3);
            sweep3<10, quad_t, four_t, four_t>(
                (jend - j0) >> 2,
                [&](quad_t &xj, double(&t0)[4], double(&t1)[4]) {
                    xj.lo = lds16<0>(xa);
                    xj.hi = lds16<16>(xa);
                    t0[0] = lds8<24>(za);
                    t1[0] = lds8<512 + 24>(za);
                    t0[1] = lds8<16>(za);
                    t1[1] = lds8<512 + 16>(za);
                    t0[2] = lds8<8>(za);
                    t1[2] = lds8<512 + 8>(za);
                    t0[3] = lds8<0>(za);
                    t1[3] = lds8<512>(za);
                    xa += 32;
                    za -= 32;
                },
                [&](const quad_t &xj, double(&t0)[4], double(&t1)[4]) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        a0 = fma(t0[e], xj[e], a0);
                        a1 = fma(t1[e], xj[e], a1);
                    }
                });
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
                    ab[2 * i0] = a0c;
                    ab[2 * i0 + 1] = b0c;
                    ab[2 * i1] = a1c;
                    ab[2 * i1 + 1] = b1c;
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
                // Both sweeps below run over groups of four columns that start at a multiple of four
                // (so that g and the coefficients of a group are aligned 16-byte reads): the columns in
                // front of qq have g = a = b = 0, which leaves the running sum at zero, and so have
                // the ones behind `last`; neither is stored.  A wave's LDS operations are carried out
                // in program order, so within one wave reads need no fence against the stores that
                // follow them; the reads of the next group go out ahead of the stores of this one.
                BSTAMP(10);
                const int jb0 = qq & ~3, dgroups = ((last - jb0) >> 2) + 1;
                if (wave < 2) {
                    // rows of J: wave 0 rows 0..63, wave 1 rows 64..n-1
                    const int r = wave == 0 ? c0 : c1;
                    const bool rv = wave == 0 ? v0 : v1;
                    double *Jr = J + r;
                    const unsigned cs = (unsigned)ldj * 8u;
                    double cj = Jr[jb0 * ldj], q = 0.0;
                    unsigned pa = lds_off(Jr + (jb0 + 1) * ldj), ga = lds_off(gs + jb0), aba = lds_off(ab + 2 * jb0);
                    int j = jb0;
                    sweep3<10, quad_t, four_t, pair4_t>(
                        dgroups,
                        [&](quad_t &gv, four_t &cn, pair4_t &cf) {
                            const unsigned p1 = pa + cs, p2 = p1 + cs, p3 = p2 + cs;
                            gv.lo = lds16<0>(ga);
                            gv.hi = lds16<16>(ga);
                            cn[0] = lds8<0>(pa);
                            cn[1] = lds8<0>(p1);
                            cn[2] = lds8<0>(p2);
                            cn[3] = lds8<0>(p3);
                            cf[0] = lds16<0>(aba);
                            cf[1] = lds16<16>(aba);
                            cf[2] = lds16<32>(aba);
                            cf[3] = lds16<48>(aba);
                            pa = p3 + cs;
                            ga += 32;
                            aba += 64;
                        },
                        [&](const quad_t &gv, four_t &cn, pair4_t &cf) {
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                q = fma(gv[e], cj, q);
                                const double o = fma(cf[e].x, q, cf[e].y * cn[e]);
                                if (rv && j + e >= qq && j + e < last) Jr[(j + e) * ldj] = o;
                                cj = e < 3 ? cn[e] : arrived_copy(cn[3]);
                            }
                            j += 4;
                        });
                    if (rv) Jr[last * ldj] = ab[2 * last] * q; // (q has taken in column `last`; behind it g = 0)
                } else if (wave == 2 || last >= HALF) {
                    // rows of T, one per lane: wave 2 rows 0 .. 63, wave 3 rows 64 .. last once the
                    // working set is beyond 64 rows.  Row i > qq moves up to i - 1, row qq goes, and so
                    // does the last column; T(i, j) = 0 below the diagonal (i > j), where the packed
                    // storage holds other columns.  With qq < 64 the new row 63 comes from row 64: wave
                    // 3 works it out, but its place is still being read by wave 2 -- it goes to `side`
                    // and is put in place after a barrier (below).
                    const int ri = wave == 2 ? i0 : i1;
                    const int nr = ri > qq ? ri - 1 : ri;
                    const bool wr = ri != qq && ri <= last;
                    const bool defer = wave == 3 && lane == 0 && qq < HALF;
                    const unsigned ta = lds_off(Tp + ri);
                    double cjr; // T(ri, j) of the running column j
                    {
                        const double t = Tp[jb0 * (jb0 + 1) / 2 + min(ri, jb0)];
                        cjr = ri <= jb0 ? t : 0.0;
                    }
                    double qr = 0.0;
                    unsigned ga = lds_off(gs + jb0), aba = lds_off(ab + 2 * jb0);
                    int ji = jb0, ju = jb0; // first column of the next group to read / to use
                    sweep3<10, quad_t, four_t, pair4_t>(
                        dgroups,
                        [&](quad_t &gv, four_t &cn, pair4_t &cf) {
                            gv.lo = lds16<0>(ga);
                            gv.hi = lds16<16>(ga);
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int jn = min(ji + e + 1, last); // column j + 1
                                cn[e] = lds8<0>(ta + (unsigned)(jn * (jn + 1) / 2) * 8u);
                            }
                            cf[0] = lds16<0>(aba);
                            cf[1] = lds16<16>(aba);
                            cf[2] = lds16<32>(aba);
                            cf[3] = lds16<48>(aba);
                            ji += 4;
                            ga += 32;
                            aba += 64;
                        },
                        [&](const quad_t &gv, four_t &cn, pair4_t &cf) {
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int j = ju + e;
                                const double nx = ri <= j + 1 ? cn[e] : 0.0;
                                qr = fma(gv[e], cjr, qr);
                                const double o = fma(cf[e].x, qr, cf[e].y * nx);
                                cjr = nx;
                                if (j >= qq && j < last && wr && nr <= j) *(defer ? side + j : Tp + j * (j + 1) / 2 + nr) = o;
                            }
                            ju += 4;
                        });
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
            BSTAMP(15);
            if (qq < last && last >= HALF && qq < HALF) {
                __syncthreads(); // wave 2 has read the old row 63
                const int j = HALF - 1 + lane;
                if (wave == 3 && j >= max(qq, HALF - 1) && j < last) Tp[j * (j + 1) / 2 + HALF - 1] = side[j];
            }
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
                            unsigned ja = lds_off(J + c0 * ldj + r0), jb = lds_off(J + c1 * ldj + r0);
                            unsigned za = lds_off(zc + kip - r0 - 3); // n+ descends from zc[kip - r]
                            double a0 = 0.0, a1 = 0.0;
                            // (rows r .. r+3 of a column are two aligned 16-byte reads)
                            sweep3<8, four_t, pair2_t, pair2_t>(
                                (r1 - r0) >> 2,
                                [&](four_t &nr, pair2_t &ca_, pair2_t &cb_) {
                                    nr[0] = lds8<24>(za);
                                    nr[1] = lds8<16>(za);
                                    nr[2] = lds8<8>(za);
                                    nr[3] = lds8<0>(za);
                                    ca_[0] = lds16<0>(ja);
                                    cb_[0] = lds16<0>(jb);
                                    ca_[1] = lds16<16>(ja);
                                    cb_[1] = lds16<16>(jb);
                                    ja += 32;
                                    jb += 32;
                                    za -= 32;
                                },
                                [&](four_t &nr, pair2_t &ca_, pair2_t &cb_) {
                                    a0 = fma(ca_[0].x, nr[0], a0);
                                    a1 = fma(cb_[0].x, nr[0], a1);
                                    a0 = fma(ca_[0].y, nr[1], a0);
                                    a1 = fma(cb_[0].y, nr[1], a1);
                                    a0 = fma(ca_[1].x, nr[2], a0);
                                    a1 = fma(cb_[1].x, nr[2], a1);
                                    a0 = fma(ca_[1].y, nr[3], a0);
                                    a1 = fma(cb_[1].y, nr[3], a1);
                                });
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
                    pair_t jbp = {0.0, 0.0}; // column slo - 1 (the lane's two rows), kept for the update of column slo
                    {
                        double a0 = 0.0, a1 = 0.0;
                        {
                            // In the sweeps along the rows of J (here and in the update below) a lane
                            // works on rows 2 lane and 2 lane + 1, one 16-byte read per column; a lane
                            // without rows reads past the end of its column, and nothing of it is kept.
                            const unsigned cs = (unsigned)ldj * 8u;
                            unsigned da = lds_off(dsv + slo), pa = lds_off(J + 2 * lane + slo * ldj);
                            sweep2<8, four_t, pair4_t>(
                                max(shi - slo, 0) >> 2,
                                [&](four_t &dj, pair4_t &rr) {
                                    const unsigned p1 = pa + cs, p2 = p1 + cs, p3 = p2 + cs;
                                    dj[0] = lds8<0>(da);
                                    dj[1] = lds8<8>(da);
                                    dj[2] = lds8<16>(da);
                                    dj[3] = lds8<24>(da);
                                    rr[0] = lds16<0>(pa);
                                    rr[1] = lds16<0>(p1);
                                    rr[2] = lds16<0>(p2);
                                    rr[3] = lds16<0>(p3);
                                    da += 32;
                                    pa = p3 + cs;
                                },
                                [&](four_t &dj, pair4_t &rr) {
#pragma unroll
                                    for (int e = 0; e < 4; e++) {
                                        a0 = fma(rr[e].x, dj[e], a0);
                                        a1 = fma(rr[e].y, dj[e], a1);
                                    }
                                });
                        }
                        if (slo > iq && slo < shi) jbp = *reinterpret_cast<const pair_t *>(J + 2 * lane + (slo - 1) * ldj);
                        {
                            pair_t zp;
                            zp.x = a0;
                            zp.y = a1;
                            *reinterpret_cast<pair_t *>(zpart + wave * PSTR + 2 * lane) = zp;
                        }
                        // r_i = sum_k T(i, k) d_k, k in this wave's quarter of [0, iq)
                        const int tseg = 4 * ((iq + 4 * BW - 1) / (4 * BW)), k0 = wave * tseg;
                        const int k1 = min(k0 + tseg, (iq + 3) & ~3);
                        double b0 = 0.0, b1 = 0.0;
                        // (a lane below the diagonal of column kk reads into the columns behind it --
                        // T is followed by padding -- and the value is discarded)
                        const unsigned ta = lds_off(Tp + i0);
                        unsigned dka = lds_off(d1s + k0);
                        int ki = k0, ku = k0; // next column to issue / to use
                        // Columns below 64 have no entry in the lanes' second rows (i1 >= 64 > kk), and
                        // from column 64 on the first rows lie above the diagonal throughout: the
                        // wave's quarter is swept in those two parts.
                        const int ksplit = min(k1, max(k0, HALF));
                        sweep2<6, quad_t, four_t>(
                            (ksplit - k0) >> 2,
                            [&](quad_t &dk, double(&t0)[4]) {
                                dk.lo = lds16<0>(dka);
                                dk.hi = lds16<16>(dka);
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int kk = ki + e;
                                    t0[e] = lds8<0>(ta + (unsigned)(kk * (kk + 1) / 2) * 8u);
                                }
                                dka += 32;
                                ki += 4;
                            },
                            [&](const quad_t &dk, double(&t0)[4]) {
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int kk = ku + e;
                                    b0 = fma(i0 <= kk ? t0[e] : 0.0, dk[e], b0);
                                }
                                ku += 4;
                            });
                        sweep3<10, quad_t, four_t, four_t>(
                            (k1 - ksplit) >> 2,
                            [&](quad_t &dk, double(&t0)[4], double(&t1)[4]) {
                                dk.lo = lds16<0>(dka);
                                dk.hi = lds16<16>(dka);
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int kk = ki + e;
                                    const unsigned a = ta + (unsigned)(kk * (kk + 1) / 2) * 8u;
                                    t0[e] = lds8<0>(a);
                                    t1[e] = lds8<512>(a);
                                }
                                dka += 32;
                                ki += 4;
                            },
                            [&](const quad_t &dk, double(&t0)[4], double(&t1)[4]) {
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int kk = ku + e;
                                    b0 = fma(t0[e], dk[e], b0);
                                    b1 = fma(i1 <= kk ? t1[e] : 0.0, dk[e], b1);
                                }
                                ku += 4;
                            });
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
                        BSTAMP(12);
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
                            ab[2 * i0] = a0c;
                            ab[2 * i0 + 1] = b0c;
                            ab[2 * i1] = a1c;
                            ab[2 * i1 + 1] = b1c;
                        }
                        wsync();
                        BSTAMP(8);
                        // the update: this wave's column segment, from its top down, the lane's rows
                        // 2 lane and 2 lane + 1; P starts at the partial sums of the segments above
                        {
                            const int ra = 2 * lane;
                            const bool both = ra + 1 < n, one = ra < n;
                            pair_t P = {0.0, 0.0};
                            for (int w2 = BW - 1; w2 > wave; w2--) P += *reinterpret_cast<const pair_t *>(zpart + w2 * PSTR + ra);
                            if (slo < shi) {
                                double *Jr = J + ra;
                                pair_t cur = *reinterpret_cast<const pair_t *>(Jr + (shi - 1) * ldj);
                                const unsigned cs = (unsigned)ldj * 8u;
                                // columns jtop .. jtop - 3 per group (the segment is a multiple of four
                                // long): column j needs column j - 1, d_j and (a_j, b_j); two groups are in
                                // flight, the reads of the second go out before the stores of the first
                                unsigned pa = lds_off(Jr + (shi - 2) * ldj), ua = lds_off(dsv + shi - 4), aba = lds_off(ab + 2 * (shi - 4));
                                int jtop = shi - 1;
                                sweep3<12, four_t, pair4_t, pair4_t>(
                                    (shi - slo) >> 2,
                                    [&](four_t &dv, pair4_t &pv, pair4_t &cf) {
                                        const unsigned p1 = pa - cs, p2 = p1 - cs, p3 = p2 - cs;
                                        dv[0] = lds8<24>(ua);
                                        dv[1] = lds8<16>(ua);
                                        dv[2] = lds8<8>(ua);
                                        dv[3] = lds8<0>(ua);
                                        pv[0] = lds16<0>(pa);
                                        pv[1] = lds16<0>(p1);
                                        pv[2] = lds16<0>(p2);
                                        pv[3] = lds16<0>(p3);
                                        cf[0] = lds16<48>(aba);
                                        cf[1] = lds16<32>(aba);
                                        cf[2] = lds16<16>(aba);
                                        cf[3] = lds16<0>(aba);
                                        pa = p3 - cs;
                                        ua -= 32;
                                        aba -= 64;
                                    },
                                    [&](four_t &dv, pair4_t &pv, pair4_t &cf) {
                                        if (jtop - 3 == slo) pv[3] = jbp; // the column below the segment was read before the barrier
#pragma unroll
                                        for (int e = 0; e < 4; e++) {
                                            const int j = jtop - e;
                                            P.x = fma(cur.x, dv[e], P.x);
                                            P.y = fma(cur.y, dv[e], P.y);
                                            pair_t o;
                                            o.x = fma(cf[e].x, P.x, cf[e].y * pv[e].x);
                                            o.y = fma(cf[e].x, P.y, cf[e].y * pv[e].y);
                                            if (j <= jt) { // columns past jt keep what they hold
                                                if (both) *reinterpret_cast<pair_t *>(Jr + j * ldj) = o;
                                                else if (one) Jr[j * ldj] = o.x;
                                            }
                                            cur = e < 3 ? pv[e] : arrived_copy(pv[3]);
                                        }
                                        jtop -= 4;
                                    });
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
                    BSTAMP(12);
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
        BSTAMP(0);
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
    const size_t doubles = n * (size_t)box_ldj((int)n) + n * (n + 1) / 2 + 2 * PSTR + 11 * PSTR + 4 * BW * PSTR;
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

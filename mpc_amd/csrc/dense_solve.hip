// Dense dual active-set kernel: ONE QP PER WORKGROUP, the reference's method in full.
//
// Second stage behind the wave-per-QP kernel.  That kernel keeps only the q x q
// factor of the working set, which loses accuracy when many nearly parallel
// constraints are active at once (in practice: infeasible or almost infeasible
// instances, whose working set grows towards n) and has a fixed capacity.  The
// instances it hands over are solved here exactly as eiquadprog.hpp:130-592 does it:
// the full n x n matrix J = L^-T Q and the triangular R, updated by Givens rotations,
// both in the workgroup's slot of an HBM workspace (n = 400: 1.28 MB each, L2/MALL
// resident while the workgroup runs); vectors of length n and m live in LDS.
// The m x n constraint matrix is still never formed: row i is
// (p_i, q_i) * Zu[k_i, :], generated from the Toeplitz column of Zu.
//
// J is column-major so that all three sweeps over it are coalesced:
//   d = J' n+   : one wave per column, lanes along the column, shuffle reduction
//   z = J2 d2   : one thread per row, axpy over columns
//   rotations   : one thread per row carries the running column in a register
// The rotation angles of add_constraint (a sequential hypot chain in the reference)
// are computed in parallel from suffix norms of d.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

#include "device_common.h"
#include "kernels.h"
#include "zspace_block.h"
#include <type_traits>

namespace mpcb {
namespace {

constexpr int T = 256;

// Diagnostic build only (-DMPCB_STAMPS): cycles per section into sd.dbg[inst*10 + section].
#ifdef MPCB_STAMPS
#define DSTAMP(i)                                          \
    do {                                                   \
        const long long t_ = __builtin_amdgcn_s_memtime(); \
        stamp_acc[i] += t_ - stamp_t;                      \
        stamp_t = t_;                                      \
    } while (0)
#else
#define DSTAMP(i) \
    do {          \
    } while (0)
#endif

// The reductions exchange per-wave partials through one of two small LDS buffers, taken in
// turn (flip): a wave can still be reading the buffer of the previous reduction while another
// already writes this one's, and the barrier of this reduction orders it against the one after
// next -- so each reduction needs a single barrier.
constexpr int NW = T / 64, RED_D = 3 * NW, RED_I = NW;

__device__ __forceinline__ double block_sum(double v, double *red, int tid, int &flip) {
    v = wave_sum(v);
    double *b = red + flip * RED_D;
    flip ^= 1;
    if ((tid & 63) == 0) b[tid >> 6] = v;
    __syncthreads();
    double tot = b[0];
#pragma unroll
    for (int w = 1; w < NW; w++) tot += b[w];
    return tot;
}

// minimum of (val, key), smaller key wins ties; result identical in every thread
// and, in the same exchange, the sums of a and b
__device__ __forceinline__ void block_argmin_sums(double &val, int &key, double &a, double &b, double *red,
                                                  int *redk, int tid, int &flip) {
    int aux = 0;
    wave_argmin(val, key, aux);
    a = wave_sum(a);
    b = wave_sum(b);
    double *bd = red + flip * RED_D;
    int *bk = redk + flip * RED_I;
    flip ^= 1;
    if ((tid & 63) == 0) {
        bd[tid >> 6] = val;
        bk[tid >> 6] = key;
        bd[NW + (tid >> 6)] = a;
        bd[2 * NW + (tid >> 6)] = b;
    }
    __syncthreads();
    val = bd[0];
    key = bk[0];
    a = bd[NW];
    b = bd[2 * NW];
#pragma unroll
    for (int w = 1; w < NW; w++) {
        if (bd[w] < val || (bd[w] == val && bk[w] < key)) {
            val = bd[w];
            key = bk[w];
        }
        a += bd[NW + w];
        b += bd[2 * NW + w];
    }
}

__device__ __forceinline__ void block_argmin(double &val, int &key, double *red, int *redk, int tid, int &flip) {
    double a = 0.0, b = 0.0;
    block_argmin_sums(val, key, a, b, red, redk, tid, flip);
}

// LDSJ: J (n x n) and R (packed, one sub-diagonal) live in the workgroup's LDS instead of
// its HBM workspace slot -- possible up to n ~ 105, e.g. the N = 100 box problems.
// FORM: 0 = box rows on one axis (solveQPconstraint), 1 = polygon rows on both axes
// (solveQPConstraintCoupled*), 2 = box rows with one slack variable per bound
// (solveQPconstraintSlack, MPCPlanner.cpp:351-394: x = [jerk; slacks], G = diag(R I, Qs I),
// g0 = [0; 1000 Qs], rows i < 2N: -/+ (zmp_k - bound) + slack_i >= 0, rows 2N + i: -slack_i >= 0).
template <int FORM, bool LDSJ>
__global__ __launch_bounds__(T) void dense_kernel(DevSetup sd, DevIn in, mpcb_outputs out, DenseWs ws) {
    constexpr bool POLY = FORM == 1, SLACK = FORM == 2;
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = sd.N, n = POLY ? 2 * N : SLACK ? 3 * N : N, m = sd.m;
    // ---- LDS carve-up -------------------------------------------------------
    double *x = smem, *z = x + n, *d = z + n, *np = d + n, *x_old = np + n;
    double *u = x_old + n, *u_old = u + (n + 1), *r = u_old + (n + 1);
    // rotation cos / sin / sin/(1+cos) per column, suffix norms; rb = -cos, or +1 where a column
    // pair is left unrotated: (new, carried) = (c t1 + s t2, s t1 + rb t2) then needs no test
    // Without LDS-resident matrices rb shares the storage of z: the step direction is dead from
    // the moment x has been updated until the next pass recomputes it, which is when rotations
    // happen.  The LDS variant computes the coefficients of a possible addition while z is still
    // live (see the pass loop) and gives rb an array of its own behind R.
    double *rc = r + (n + 1), *rs = rc + n, *rx = rs + n, *rb = z, *suf = rx + n;
    double *s = suf + (n + 1);
    double *zx = s + m, *zy = zx + N;
    double *lo = zy + N, *hi = lo + N;                                   // BOX limits
    double *zc = hi + N;                                                 // Toeplitz column of Zu
    double *tab = zc + N;                                                // POLY: [phase][edge][p,q,r]
    double *cref = tab + (POLY ? sd.nphase * 12 : 0);                    // POLY: [phase][x,y]
    double *red = cref + (POLY ? 2 * max(sd.nphase, sd.nref) : 0);
    int *redk = reinterpret_cast<int *>(red + 2 * RED_D);
    int *A = redk + 2 * RED_I, *A_old = A + (n + 1), *iai = A_old + (n + 1);
    int flip = 0;
    int *iaexcl = iai + m;
    // 8-byte aligned start of the LDS-resident matrices.  Plain pointer arithmetic on smem: a
    // round trip through an integer would turn every access to J and R into a flat load.
    const int ints_used = 2 * RED_I + 2 * (n + 1) + 2 * m;
    double *lds_mats = (red + 2 * RED_D) + (ints_used + 1) / 2;
    __shared__ int sh_i[4];
    // leading dimension of J: odd in LDS so that one thread per column walks conflict-free
    const int ldj = LDSJ ? (n | 1) : n;
    // R(i, k), i <= k + 1: full column-major in HBM; in LDS column k keeps k + 2 entries
    auto ridx = [&](int i, int k) -> size_t {
        if constexpr (LDSJ) return (size_t)k * (k + 3) / 2 + i;
        else return (size_t)k * n + i;
    };

    // Coupled forms with HBM-resident factors: the instances that outgrew the wide edition of the
    // wave-per-QP kernel (ws.block_list) come first and are tried with the block edition of that
    // method (zspace_block.h), which needs nothing n x n; the full method below takes what that
    // cannot finish, and then its own list.
    const int count = ws.count[CNT_HANDOVER];
    int nblock = 0;
    if constexpr (POLY && !LDSJ) nblock = ws.block_list ? ws.count[CNT_BLOCK] : 0;
    for (int entry = blockIdx.x; entry < nblock + count; entry += gridDim.x) {
        int inst;
        if constexpr (POLY && !LDSJ) {
            if (entry < nblock) {
                inst = ws.block_list[entry];
                if (block_solve(sd, in, out, ws, inst, smem)) continue;
                if (tid == 0) atomicAdd(ws.count + CNT_FELL, 1);
            } else {
                inst = ws.list[entry - nblock];
            }
        } else {
            inst = ws.list[entry];
        }
        // Cz = Cx - c Ca (MPCPlanner.cpp:58-63): with a height per instance the Toeplitz column of
        // Zu and the third column of Zx are shifted by c = height / gravity
        const bool hv = in.height != nullptr;
        const double ch = hv ? in.height[inst] * sd.inv_gravity : 0.0;
        __syncthreads();
        for (int k = tid; k < N; k += T) zc[k] = hv ? ws.xp_col[k] - ch * sd.Ts : ws.zu_col[k];
        auto zx_row = [&](int k, double &a, double &b, double &c) {
            const double *zr = (hv ? ws.xpx : ws.zx) + (size_t)k * 3;
            a = zr[0];
            b = zr[1];
            c = hv ? zr[2] - ch : zr[2];
        };
#ifdef MPCB_STAMPS
        long long stamp_acc[16] = {0};
        long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
        double *J, *R; // J column-major n x ldj; R upper triangular plus one sub-diagonal
        if constexpr (LDSJ) {
            J = lds_mats;
            R = lds_mats + (size_t)n * ldj;
            rb = R + (size_t)n * (n + 3) / 2 + 2;
        } else {
            J = ws.J + (size_t)blockIdx.x * n * n;
            R = ws.R + (size_t)blockIdx.x * n * n;
        }
        __syncthreads();

        // ---- inputs, constraint rows ---------------------------------------
        double wx[4] = {0, 0, 0, 0}, wy[4] = {0, 0, 0, 0};
        {
            const double *px = in.x0 + (size_t)inst * 3;
            wx[0] = px[0];
            wx[1] = px[1];
            wx[2] = px[2];
            if constexpr (POLY) {
                const double *py = in.y0 + (size_t)inst * 3;
                wy[0] = py[0];
                wy[1] = py[1];
                wy[2] = py[2];
                wx[3] = in.speed[(size_t)inst * 2];
                wy[3] = in.speed[(size_t)inst * 2 + 1];
            }
        }
        const double *rows = (POLY && in.rows_pqr) ? in.rows_pqr + (size_t)inst * m * 3 : nullptr;
        if constexpr (POLY) {
            if (!rows)
                for (int phb = 0; phb < sd.nphase; phb += T)
                    polygon_phase(sd, in, inst, wx[3], wy[3], phb + tid, tab, cref);
            if (in.zref_phase)
                for (int i = tid; i < 2 * sd.nref; i += T) cref[i] = in.zref_phase[(size_t)inst * 2 * sd.nref + i];
        } else {
            for (int k = tid; k < N; k += T) {
                lo[k] = in.zmin[(size_t)inst * N + k];
                hi[k] = in.zmax[(size_t)inst * N + k];
            }
        }
        __syncthreads();
        auto rowK = [&](int i) -> int {
            if constexpr (POLY) return ws.row_knot[i];
            else return i < N ? i : i - N;
        };
        auto rowP = [&](int i, int k) -> double {
            if constexpr (POLY) return rows ? rows[i * 3] : tab[(sd.knot_phase[k] * 4 + ws.row_slot[i]) * 3];
            else return i < N ? 1.0 : -1.0;
        };
        auto rowQ = [&](int i, int k) -> double {
            if constexpr (POLY) return rows ? rows[i * 3 + 1] : tab[(sd.knot_phase[k] * 4 + ws.row_slot[i]) * 3 + 1];
            else return 0.0;
        };
        auto rowR = [&](int i, int k) -> double {
            if constexpr (POLY) return rows ? rows[i * 3 + 2] : tab[(sd.knot_phase[k] * 4 + ws.row_slot[i]) * 3 + 2];
            else return i < N ? -lo[k] : hi[k];
        };
        // s_i = n_i' x + ci0_i through the ZMP: zx = Zx x0 + Zu x  (never forms CI).
        // value of row i given the ZMP (zxk, zyk) of its knot
        auto row_value = [&](int i, int k, double zxk, double zyk) -> double {
            if constexpr (SLACK) {
                if (i >= 2 * N) return -x[N + (i - 2 * N)];
                return (i < N ? zxk - lo[k] : hi[k] - zxk) + x[N + i];
            } else {
                return rowP(i, k) * zxk + rowQ(i, k) * zyk + rowR(i, k);
            }
        };
        auto row_knot = [&](int i) -> int {
            if constexpr (SLACK) return i >= 2 * N ? 0 : (i < N ? i : i - N);
            else return rowK(i);
        };
        // all rows: the triangular product Zu x is cut into NW column bands, one per wave, lanes
        // over the knots; the bands' partial sums (kept in the rotation coefficient arrays, which
        // are dead between updates of the factor) are added in a fixed order per row.  Ends with
        // s written but NOT yet visible to other threads: the caller's next barrier does that.
        auto eval_all = [&]() {
            double *part = wave == 0 ? rc : wave == 1 ? rs : wave == 2 ? rx : suf;
            const int band = (N + NW - 1) / NW, j0 = wave * band;
            for (int k = lane; k < N; k += 64) {
                double ax = 0.0, ay = 0.0;
                if (wave == 0) {
                    double z0, z1, z2;
                    zx_row(k, z0, z1, z2);
                    ax = z0 * wx[0] + z1 * wx[1] + z2 * wx[2];
                    ay = z0 * wy[0] + z1 * wy[1] + z2 * wy[2];
                }
                const int j1 = min(k, j0 + band - 1);
#pragma unroll 8
                for (int j = j0; j <= j1; j++) {
                    const double c = zc[k - j];
                    ax += c * x[j];
                    if constexpr (POLY) ay += c * x[N + j];
                }
                part[k] = ax;
                if constexpr (POLY) part[N + k] = ay;
            }
            __syncthreads();
            for (int i = tid; i < m; i += T) {
                const int k = row_knot(i);
                const double zxk = ((rc[k] + rs[k]) + rx[k]) + suf[k];
                double zyk = 0.0;
                if constexpr (POLY) zyk = ((rc[N + k] + rs[N + k]) + rx[N + k]) + suf[N + k];
                s[i] = row_value(i, k, zxk, zyk);
            }
        };
        // one row: a single dot product, lanes of wave 0 over the jerks
        auto eval_one = [&](int i) {
            if (wave == 0) {
                const int k = row_knot(i);
                double ax = 0.0, ay = 0.0;
                for (int j = lane; j <= k; j += 64) {
                    const double c = zc[k - j];
                    ax += c * x[j];
                    if constexpr (POLY) ay += c * x[N + j];
                }
                ax = wave_sum(ax);
                if constexpr (POLY) ay = wave_sum(ay);
                double z0, z1, z2;
                zx_row(k, z0, z1, z2);
                ax += z0 * wx[0] + z1 * wx[1] + z2 * wx[2];
                ay += z0 * wy[0] + z1 * wy[1] + z2 * wy[2];
                if (lane == 0) s[i] = row_value(i, k, ax, ay);
            }
            __syncthreads();
        };

        // ---- preprocessing (eiquadprog.hpp:162-191) -------------------------
        // J = L^-T = blkdiag(J1, J1); x = -G^-1 g0 ; f = 1/2 g0'x
        for (int col = 0; col < n; col++)
            for (int row = tid; row < n; row += T) {
                double v = 0.0;
                if (col < N && row < N) v = ws.J1[(size_t)col * N + row];
                else if (SLACK && col >= N && row == col) v = 1.0 / sd.slack_ls;
                else if (POLY && col >= N && row >= N) v = ws.J1[(size_t)(col - N) * N + (row - N)];
                J[(size_t)col * ldj + row] = v;
            }
        // slack part of x = -G^-1 g0 through the factor, as chol.solve does (eiquadprog.hpp:188)
        const double slack_x0 = -((sd.slack_g0 / sd.slack_ls) / sd.slack_ls);
        for (int i = tid; i < n; i += T) {
            if (SLACK && i >= N) {
                x[i] = slack_x0;
                continue;
            }
            const int k = i < N ? i : i - N;
            const double *kx = sd.Kx + (size_t)k * 4;
            const double *w = i < N ? wx : wy;
            double v = kx[0] * w[0] + kx[1] * w[1] + kx[2] * w[2] + kx[3] * w[3];
            if constexpr (POLY) {
                if (sd.Wp)
                    for (int p2 = 0; p2 < sd.nref; p2++)
                        v += cref[2 * p2 + (i < N ? 0 : 1)] * sd.Wp[(size_t)p2 * N + k];
            }
            x[i] = v;
        }
        double f;
        {
            double fp = 0.0;
            for (int pb = 0; pb < (sd.F2 ? sd.nref : 1); pb += T)
                fp += unconstrained_cost_part(sd, wx, cref, 0, pb + tid, pb + tid == 0) +
                      unconstrained_cost_part(sd, wy, cref, 1, pb + tid, pb + tid == 0);
            f = block_sum(fp, red, tid, flip);
            if constexpr (SLACK) f += 0.5 * (2.0 * N) * sd.slack_g0 * slack_x0; // 1/2 g0'x (:191)
        }
        for (int i = tid; i < m; i += T) iai[i] = i;
        // stopping threshold (eiquadprog.hpp:267) on the rows the caller really has: empty slots
        // of caller-built rows ((0, 0, +inf)) do not count
        double psi_thr = sd.psi_thr;
        if constexpr (POLY) {
            if (rows) {
                double mine = 0.0;
                for (int i = tid; i < m; i += T) mine += rows[i * 3 + 2] < DINF ? 1.0 : 0.0;
                psi_thr = sd.psi_thr * (block_sum(mine, red, tid, flip) / (double)m);
            }
        }
        double R_norm = 1.0;
        int iq = 0, iters = 0, passes = 0, status = MPCB_OK;
        int ip = 0;
        const double nn = (double)n, mm = (double)m;
        double falg = 2.0 * nn * nn;
        __syncthreads();

        // delete_constraint (eiquadprog.hpp:516-592); u has iq+1 live entries
        auto delete_constraint = [&](int l) {
            // position of l in the working set (its entries are distinct: one thread finds it)
            for (int i = tid; i < iq; i += T)
                if (A[i] == l) sh_i[0] = i;
            __syncthreads();
            const int qq = ufirst(sh_i[0]);
            {
                const double c = (double)(iq - 1 - qq);
                falg += 6.0 * nn * c + 3.0 * c * (c - 1.0);
            }
            // shift the columns of R: one thread per row, eight entries read before any is written
            for (int row = tid; row < iq; row += T)
                for (int i0 = (qq > row - 1 ? qq : row - 1); i0 < iq - 1; i0 += 8) {
                    double v[8];
#pragma unroll
                    for (int q8 = 0; q8 < 8; q8++) v[q8] = R[ridx(row, min(i0 + q8, iq - 2) + 1)];
#pragma unroll
                    for (int q8 = 0; q8 < 8; q8++)
                        if (i0 + q8 < iq - 1) R[ridx(row, i0 + q8)] = v[q8];
                }
            // shift A and u: chunks of T entries, each read, then (after a barrier) written
            for (int c0 = qq; c0 < iq; c0 += T) {
                const int i = c0 + tid;
                int av = 0;
                double uv = 0.0;
                if (i < iq) {
                    av = A[i + 1];
                    uv = u[i + 1];
                }
                __syncthreads();
                if (i < iq) {
                    A[i] = av;
                    u[i] = uv;
                }
            }
            if (tid == 0) {
                A[iq] = 0;
                u[iq] = 0.0;
            }
            __syncthreads();
            for (int j = tid; j < iq; j += T) R[ridx(j, iq - 1)] = 0.0;
            iq = ufirst(iq - 1); // wave-uniform by construction; saying so keeps loop control scalar
            __syncthreads();
            if (iq == 0) return;
            // rotations of rows (j, j+1) of R: a true sequential chain
            if constexpr (LDSJ) {
                // n <= 128: wave 0, lane k (and k+64) owns column k and carries the current
                // row-j entry in a register; the pivot pair travels by readlane
                if (wave == 0) {
                    const int k0 = lane, k1 = lane + 64;
                    double top0 = (k0 >= qq && k0 < iq) ? R[ridx(qq, k0)] : 0.0;
                    double top1 = (k1 >= qq && k1 < iq) ? R[ridx(qq, k1)] : 0.0;
                    for (int j = qq; j < iq; j++) {
                        const double bot0 = (k0 >= j && k0 < iq) ? R[ridx(j + 1, k0)] : 0.0;
                        const double bot1 = (k1 >= j && k1 < iq) ? R[ridx(j + 1, k1)] : 0.0;
                        const double tsel = j >= 64 ? top1 : top0, bsel = j >= 64 ? bot1 : bot0;
                        const double tj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tsel), j & 63),
                                                           __builtin_amdgcn_readlane(__double2loint(tsel), j & 63));
                        const double bj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bsel), j & 63),
                                                           __builtin_amdgcn_readlane(__double2loint(bsel), j & 63));
                        // h = hypot(tj, bj) (distance(), eiquadprog.hpp:90-108) and 1/h from one
                        // chain; cos and sin by multiplication
                        const double hh = tj * tj + bj * bj;
                        double h = 0.0, ih = 0.0;
                        if (hh > 0.0) sqrt_rsqrt(hh, h, ih);
                        if (h == 0.0) {
                            if (lane == 0) {
                                rc[j] = 1.0;
                                rs[j] = 0.0;
                                rx[j] = DINF;
                                rb[j] = 1.0;
                            }
                            if (k0 >= j && k0 < iq) R[ridx(j, k0)] = top0;
                            if (k1 >= j && k1 < iq) R[ridx(j, k1)] = top1;
                            top0 = bot0;
                            top1 = bot1;
                            continue;
                        }
                        double cc = tj * ih, ss = bj * ih, hd = h;
                        if (cc < 0.0) {
                            cc = -cc;
                            ss = -ss;
                            hd = -h;
                        }
                        if (lane == 0) {
                            rc[j] = cc;
                            rs[j] = ss;
                            rx[j] = 0.0;
                            rb[j] = -cc;
                        }
                        if (k0 == j || k1 == j) {
                            R[ridx(j, j)] = hd;
                            R[ridx(j + 1, j)] = 0.0;
                        }
                        if (k0 > j && k0 < iq) {
                            R[ridx(j, k0)] = top0 * cc + bot0 * ss;
                            top0 = top0 * ss - bot0 * cc;
                        }
                        if (k1 > j && k1 < iq) {
                            R[ridx(j, k1)] = top1 * cc + bot1 * ss;
                            top1 = top1 * ss - bot1 * cc;
                        }
                    }
                }
                __syncthreads();
            } else {
            for (int j = qq; j < iq; j++) {
                if (tid == 0) {
                    double cc = R[ridx(j, j)], ss = R[ridx(j + 1, j)];
                    const double h = distance(cc, ss);
                    if (h == 0.0) {
                        rc[j] = 1.0;
                        rs[j] = 0.0;
                        rx[j] = DINF; // marks "no rotation"
                        rb[j] = 1.0;
                    } else {
                        cc /= h;
                        ss /= h;
                        R[ridx(j + 1, j)] = 0.0;
                        if (cc < 0.0) {
                            R[ridx(j, j)] = -h;
                            cc = -cc;
                            ss = -ss;
                        } else
                            R[ridx(j, j)] = h;
                        rc[j] = cc;
                        rs[j] = ss;
                        rx[j] = ss / (1.0 + cc);
                        rb[j] = -cc;
                    }
                }
                __syncthreads();
                if (rx[j] != DINF) {
                    const double cc = rc[j], ss = rs[j], xny = rx[j];
                    for (int k = j + 1 + tid; k < iq; k += T) {
                        const double t1 = R[ridx(j, k)], t2 = R[ridx(j + 1, k)];
                        const double nv = t1 * cc + t2 * ss;
                        R[ridx(j, k)] = nv;
                        R[ridx(j + 1, k)] = xny * (t1 + nv) - t2;
                    }
                }
                __syncthreads();
            }
            }
            // the same rotations on columns (j, j+1) of J: one thread per row.  Eight columns
            // are read before any is written back so the loads overlap (a store to J between
            // two loads would otherwise serialise them).
            for (int row = tid; row < n; row += T) {
                double t1 = J[(size_t)qq * ldj + row];
                // FULL: all eight columns exist, nothing is tested.  Otherwise the loads go to a
                // clamped (always valid) column and the values are replaced by the padding
                // (c, s, b, t2) = (0, 1, 0, 0), which leaves the carried value alone.
                auto group = [&](int j0, auto full) {
                    constexpr bool FULL = decltype(full)::value;
                    double t2v[8], outv[8], cv[8], sv[8], xv[8];
#pragma unroll
                    for (int q8 = 0; q8 < 8; q8++) {
                        const bool in = FULL || j0 + q8 < iq;
                        const int jc = FULL ? j0 + q8 : min(j0 + q8, iq - 1);
                        const double a = J[(size_t)(jc + 1) * ldj + row], b = rc[jc], c = rs[jc], e = rb[jc];
                        t2v[q8] = in ? a : 0.0;
                        cv[q8] = in ? b : 0.0;
                        sv[q8] = in ? c : 1.0;
                        xv[q8] = in ? e : 0.0;
                    }
#pragma unroll
                    for (int q8 = 0; q8 < 8; q8++) {
                        outv[q8] = fma(t1, cv[q8], t2v[q8] * sv[q8]);
                        t1 = fma(t1, sv[q8], t2v[q8] * xv[q8]);
                    }
#pragma unroll
                    for (int q8 = 0; q8 < 8; q8++)
                        if (FULL || j0 + q8 < iq) J[(size_t)(j0 + q8) * ldj + row] = outv[q8];
                };
                int j0 = qq;
                for (; j0 + 8 <= iq; j0 += 8) group(j0, std::true_type{});
                if (j0 < iq) group(j0, std::false_type{});
                J[(size_t)iq * ldj + row] = t1;
            }
            __syncthreads();
        };

        // Rotation coefficients of add_constraint (eiquadprog.hpp:450-513) for the current d:
        // the reference's sequential hypot chain over d[n-1], d[n-2], ... is the suffix norm
        // nu_j = ||d[j..n-1]||, so all rotations come out in parallel.  Threads t0, t0 + nt, ...
        // of a group that `sync` synchronises.
        auto add_coefs = [&](int t0, int nt, auto sync) {
            for (int j = iq + t0; j < n; j += nt) {
                double acc = 0.0;
#pragma unroll 8
                for (int i = n - 1; i >= j; i--) acc += d[i] * d[i];
                suf[j] = sqrt(acc);
            }
            sync();
            for (int j = iq + 1 + t0; j < n; j += nt) {
                // rotation of (j-1, j): cc = d[j-1], ss = value carried at j
                const double h = suf[j - 1];
                if (h == 0.0) {
                    rc[j] = 1.0;
                    rs[j] = 0.0;
                    rb[j] = 1.0;
                } else {
                    double cur = (j == n - 1) ? d[j] : (d[j] < 0.0 ? -suf[j] : suf[j]);
                    double cc = d[j - 1] / h, sv = cur / h;
                    if (cc < 0.0) {
                        cc = -cc;
                        sv = -sv;
                    }
                    rc[j] = cc;
                    rs[j] = sv;
                    rb[j] = -cc;
                }
            }
        };

        bool done = false;
        DSTAMP(0);
        while (!done) { // step 1 (l1)
            iters++;
            if (iters > sd.iter_cap) {
                status = MPCB_ITER_CAP;
                break;
            }
            for (int i = tid; i < iq; i += T) iai[A[i]] = -1;
            for (int i = tid; i < m; i += T) iaexcl[i] = 1;
            __syncthreads(); // iai, iaexcl
            eval_all();
            DSTAMP(1);
            falg += 2.0 * mm * nn;
            // the sum of infeasibilities and the most violated row in one exchange; every
            // thread looks at the rows it has just written itself
            double psi = 0.0, unused = 0.0, bv0 = DINF;
            int bi0 = 0x7fffffff;
            for (int i = tid; i < m; i += T) {
                const double si = s[i];
                psi += fmin(0.0, si);
                if (si < 0.0 && iai[i] != -1 && (si < bv0 || (si == bv0 && i < bi0))) {
                    bv0 = si;
                    bi0 = i;
                }
            }
            block_argmin_sums(bv0, bi0, psi, unused, red, redk, tid, flip);
            if (fabs(psi) <= psi_thr) break;
            for (int i = tid; i < iq; i += T) {
                u_old[i] = u[i];
                A_old[i] = A[i];
            }
            for (int i = tid; i < n; i += T) x_old[i] = x[i];
            double ss = 0.0;
            bool restart, first = true;
            do { // step 2 (l2)
                restart = false;
                double bv = bv0;
                int bi = bi0;
                if (!first) { // after a degenerate addition: iaexcl and iai have changed
                    __syncthreads();
                    bv = DINF;
                    bi = 0x7fffffff;
                    for (int i = tid; i < m; i += T)
                        if (s[i] < ss && iai[i] != -1 && iaexcl[i] && (s[i] < bv || (s[i] == bv && i < bi))) {
                            bv = s[i];
                            bi = i;
                        }
                    block_argmin(bv, bi, red, redk, tid, flip);
                }
                first = false;
                DSTAMP(2);
                if (bv < ss) {
                    ss = bv;
                    ip = ufirst(bi);
                }
                if (ss >= 0.0) {
                    done = true;
                    break;
                }
                // np = CI.col(ip)
                if constexpr (SLACK) {
                    const int k = ip < N ? ip : ip - N;
                    const double p = ip < N ? 1.0 : -1.0;
                    for (int j = tid; j < n; j += T) {
                        double v = 0.0;
                        if (ip < 2 * N) {
                            if (j < N) v = j <= k ? p * zc[k - j] : 0.0;
                            else v = j == N + ip ? 1.0 : 0.0;
                        } else {
                            v = j == N + (ip - 2 * N) ? -1.0 : 0.0;
                        }
                        np[j] = v;
                    }
                } else {
                    const int k = rowK(ip);
                    const double p = rowP(ip, k), q = rowQ(ip, k);
                    for (int j = tid; j < N; j += T) {
                        const double c = j <= k ? zc[k - j] : 0.0;
                        np[j] = p * c;
                        if constexpr (POLY) np[N + j] = q * c;
                    }
                }
                if (tid == 0) {
                    u[iq] = 0.0;
                    A[iq] = ip;
                }
                __syncthreads();
                for (;;) { // step 2a (l2a)
                    iq = ufirst(iq);
                    passes++;
                    if (passes > sd.iter_cap) {
                        status = MPCB_ITER_CAP;
                        done = true;
                        break;
                    }
                    falg += 2.0 * nn * nn + 2.0 * nn * (nn - iq) + (double)iq * iq;
                    DSTAMP(2);
                    // d = J' np : one wave per column
                    if constexpr (LDSJ) {
                        // one thread per column; the odd leading dimension keeps the lanes on
                        // different banks
                        // np is zero beyond the knot of row ip on each axis, and adding those
                        // exact zeros changes nothing: the sums stop there
                        const int kz = SLACK ? n - 1 : rowK(ip);
                        for (int j = tid; j < n; j += T) {
                            const double *col = J + (size_t)j * ldj;
                            double acc = 0.0;
#pragma unroll 8
                            for (int k = 0; k <= kz; k++) acc += col[k] * np[k];
                            if constexpr (POLY) {
#pragma unroll 8
                                for (int k = N; k <= N + kz; k++) acc += col[k] * np[k];
                            }
                            d[j] = acc;
                        }
                    } else {
                        for (int j = wave; j < n; j += T / 64) {
                            const double *col = J + (size_t)j * ldj;
                            double acc = 0.0;
                            for (int k = lane; k < n; k += 64) acc += col[k] * np[k];
                            acc = wave_sum(acc);
                            if (lane == 0) d[j] = acc;
                        }
                    }
                    __syncthreads();
                    DSTAMP(3);
                    // z = J2 d2 : one thread per row
                    // (LDS variant, n <= 128: wave 0 goes straight to the back substitution, which
                    // is the longest chain of the pass; waves 1 and 2 share the rows of z; wave 3
                    // prepares the rotation coefficients of an addition, which is how two passes
                    // in three end and depends on d alone)
                    // The LDS variant also folds the reductions of the step lengths into this
                    // stretch: waves 1 and 2 leave their parts of z.z and z.n+, wave 0 the minimum
                    // ratio u_k / r_k (from r still in registers), in the exchange buffer of this
                    // pass; the barrier after the back substitution publishes all of it.
                    double *xd = red + flip * RED_D;
                    int *xk = redk + flip * RED_I;
                    if constexpr (LDSJ) flip ^= 1;
                    // (thread 0 rewrites u[iq] after the step: every thread reads it here, ahead of
                    // the barrier that follows the back substitution)
                    const double uq_early = u[iq];
                    if (!LDSJ || wave == 1 || wave == 2) {
                        double pz = 0.0, pn = 0.0;
                        for (int row = LDSJ ? tid - 64 : tid; row < n; row += LDSJ ? 128 : T) {
                            double acc = 0.0;
#pragma unroll 8
                            for (int j = iq; j < n; j++) acc += J[(size_t)j * ldj + row] * d[j];
                            z[row] = acc;
                            pz += acc * acc;
                            pn += acc * np[row];
                        }
                        if constexpr (LDSJ) {
                            pz = wave_sum(pz);
                            pn = wave_sum(pn);
                            if (lane == 0) {
                                xd[NW + wave] = pz;
                                xd[2 * NW + wave] = pn;
                            }
                        }
                    }
                    if constexpr (LDSJ) {
                        if (wave == 3) add_coefs(lane, 64, [] { wsync(); });
                    }
                    DSTAMP(4);
                    // r = R^-1 d1 (back substitution, column oriented)
                    if constexpr (LDSJ) {
                        // n <= 128: wave 0 holds r in two registers per lane and broadcasts
                        // r_k by readlane -- no barrier inside the chain; the column of R and
                        // the reciprocal pivot do not depend on the chain and are read ahead
                        if (wave == 0) {
                            double r0 = lane < iq ? d[lane] : 0.0, r1 = lane + 64 < iq ? d[lane + 64] : 0.0;
                            // reciprocal pivots: two per lane, computed side by side ahead of the chain
                            const double iv0 = lane < iq ? 1.0 / R[ridx(lane, lane)] : 0.0;
                            const double iv1 = lane + 64 < iq ? 1.0 / R[ridx(lane + 64, lane + 64)] : 0.0;
                            // eight columns of R per group: their loads are issued together, the
                            // chain then runs from registers
                            // HI: the group touches columns >= 64, whose rows need the second
                            // register of each lane; below that only r0 takes part
                            auto group = [&](int k0, auto hi) {
                                constexpr bool HI = decltype(hi)::value;
                                double c0v[8], c1v[8];
#pragma unroll
                                for (int q8 = 0; q8 < 8; q8++) {
                                    const int k = k0 - q8;
                                    // always a valid address (row clamped into the column), masked
                                    // afterwards: no branch around the load
                                    const int kk = k >= 0 ? k : 0;
                                    const double a0 = R[ridx(lane < kk ? lane : kk, kk)];
                                    // the reciprocal pivot goes into the column here, off the chain
                                    const double ivsel = HI && kk >= 64 ? iv1 : iv0;
                                    const double inv = __hiloint2double(
                                        __builtin_amdgcn_readlane(__double2hiint(ivsel), kk & 63),
                                        __builtin_amdgcn_readlane(__double2loint(ivsel), kk & 63));
                                    c0v[q8] = (k >= 0 && lane < k) ? a0 * inv : 0.0;
                                    if constexpr (HI) {
                                        const double a1 = R[ridx(lane + 64 < kk ? lane + 64 : kk, kk)];
                                        c1v[q8] = (k >= 0 && lane + 64 < k) ? a1 * inv : 0.0;
                                    }
                                }
                                // keep the loads of the group ahead of the chain (the scheduler would
                                // otherwise sink each load next to its use and serialise on LDS latency)
                                __builtin_amdgcn_sched_barrier(0);
                                // chain: acc_i -= (R_ik / R_kk) * acc_k -- one broadcast and one FMA
                                // per column; acc_k is final once the columns above k are done, and
                                // r_k = acc_k / R_kk is formed for all k at the end
#pragma unroll
                                for (int q8 = 0; q8 < 8; q8++) {
                                    const int k = k0 - q8;
                                    if (k >= 0) {
                                        const double asel = HI && k >= 64 ? r1 : r0;
                                        const double ak = __hiloint2double(
                                            __builtin_amdgcn_readlane(__double2hiint(asel), k & 63),
                                            __builtin_amdgcn_readlane(__double2loint(asel), k & 63));
                                        r0 = fma(-c0v[q8], ak, r0);
                                        if constexpr (HI) r1 = fma(-c1v[q8], ak, r1);
                                    }
                                }
                            };
                            int k0 = iq - 1;
                            for (; k0 >= 64; k0 -= 8) group(k0, std::true_type{});
                            for (; k0 >= 0; k0 -= 8) group(k0, std::false_type{});
                            r0 *= iv0;
                            r1 *= iv1;
                            if (lane < iq) r[lane] = r0;
                            if (lane + 64 < iq) r[lane + 64] = r1;
                            // step 2b: minimum of u_k / r_k over r_k > 0, lowest k on ties
                            double ta = (lane < iq && r0 > 0.0) ? u[lane] / r0 : DINF;
                            const double tb = (lane + 64 < iq && r1 > 0.0) ? u[lane + 64] / r1 : DINF;
                            int ka = lane, aux = 0;
                            if (tb < ta) {
                                ta = tb;
                                ka = lane + 64;
                            }
                            wave_argmin(ta, ka, aux);
                            if (lane == 0) {
                                xd[0] = ta;
                                xk[0] = ka;
                            }
                        }
                        __syncthreads();
                    } else {
                        for (int i = tid; i < iq; i += T) r[i] = d[i];
                        __syncthreads();
                        for (int k = iq - 1; k >= 0; k--) {
                            if (tid == 0) r[k] = r[k] / R[ridx(k, k)];
                            __syncthreads();
                            const double rk = r[k];
                            for (int i = tid; i < k; i += T) r[i] -= R[ridx(i, k)] * rk;
                            __syncthreads();
                        }
                    }
                    DSTAMP(5);
                    // step lengths
                    double t1 = DINF, zz = 0.0, zn = 0.0, uq;
                    int kmin = 0x7fffffff;
                    if constexpr (LDSJ) {
                        t1 = xd[0];
                        kmin = xk[0];
                        zz = xd[NW + 1] + xd[NW + 2];
                        zn = xd[2 * NW + 1] + xd[2 * NW + 2];
                        uq = uq_early;
                    } else {
                        for (int k = tid; k < iq; k += T)
                            if (r[k] > 0.0) {
                                const double tmp = u[k] / r[k];
                                if (tmp < t1 || (tmp == t1 && k < kmin)) {
                                    t1 = tmp;
                                    kmin = k;
                                }
                            }
                        for (int i = tid; i < n; i += T) {
                            zz += z[i] * z[i];
                            zn += z[i] * np[i];
                        }
                        // read ahead of the exchange below, whose barrier then lets thread 0 update
                        // it without another one
                        uq = u[iq];
                        block_argmin_sums(t1, kmin, zz, zn, red, redk, tid, flip);
                    }
                    const int l = t1 < DINF ? A[kmin] : 0;
                    DSTAMP(6);
                    const double t2 = fabs(zz) > EPS ? -s[ip] / zn : DINF;
                    const double t = fmin(t1, t2);
                    if (t >= DINF) {
                        status = MPCB_INFEASIBLE;
                        f = DINF;
                        done = true;
                        break;
                    }
                    // Every thread updates only entries of x and u that it alone touches in this
                    // stretch, so the step needs no barrier of its own; the branches below start
                    // with one where they go on to shared bookkeeping.
                    if (t2 >= DINF) { // dual step
                        for (int k = tid; k < iq; k += T) u[k] -= t * r[k];
                        if (tid == 0) {
                            u[iq] = uq + t;
                            iai[l] = l;
                        }
                        __syncthreads();
                        delete_constraint(l);
                        continue;
                    }
                    for (int i = tid; i < n; i += T) x[i] += t * z[i];
                    f += t * zn * (0.5 * t + uq);
                    for (int k = tid; k < iq; k += T) u[k] -= t * r[k];
                    if (tid == 0) u[iq] = uq + t;
                    if (t == t2) {
                        // add_constraint (eiquadprog.hpp:450-513)
                        falg += 6.0 * nn * (nn - iq - 1);
                        DSTAMP(10);
                        if constexpr (!LDSJ) {
                            __syncthreads(); // rb shares the storage of z, which the step above read
                            add_coefs(tid, T, [] { __syncthreads(); });
                            __syncthreads();
                        }
                        // new diagonal entry of R: the hypot chain ends in +-||d[iq..n-1]||
                        const double dq = (n - 1 >= iq + 1 && suf[iq] != 0.0) ? (d[iq] < 0.0 ? -suf[iq] : suf[iq]) : d[iq];
                        DSTAMP(11);
                        for (int row = tid; row < n; row += T) {
                            if (n - 1 >= iq + 1) {
                                double t2c = J[(size_t)(n - 1) * ldj + row];
                                // as in delete_constraint: untested full groups, then one group
                                // with clamped loads and the padding (c, s, b, t1) = (0, 1, 0, 0)
                                auto group = [&](int j0, auto full) {
                                    constexpr bool FULL = decltype(full)::value;
                                    double t1v[8], outv[8], cv[8], sv[8], xv[8];
#pragma unroll
                                    for (int q8 = 0; q8 < 8; q8++) {
                                        const bool in = FULL || j0 - q8 >= iq + 1;
                                        const int jc = FULL ? j0 - q8 : max(j0 - q8, iq + 1);
                                        const double a = J[(size_t)(jc - 1) * ldj + row], b = rc[jc], c = rs[jc],
                                                     e = rb[jc];
                                        t1v[q8] = in ? a : 0.0;
                                        cv[q8] = in ? b : 0.0;
                                        sv[q8] = in ? c : 1.0;
                                        xv[q8] = in ? e : 0.0;
                                    }
#pragma unroll
                                    for (int q8 = 0; q8 < 8; q8++) {
                                        outv[q8] = fma(t2c, xv[q8], t1v[q8] * sv[q8]);
                                        t2c = fma(t2c, sv[q8], t1v[q8] * cv[q8]);
                                    }
#pragma unroll
                                    for (int q8 = 0; q8 < 8; q8++)
                                        if (FULL || j0 - q8 >= iq + 1) J[(size_t)(j0 - q8) * ldj + row] = outv[q8];
                                };
                                int j0 = n - 1;
                                for (; j0 - 7 >= iq + 1; j0 -= 8) group(j0, std::true_type{});
                                if (j0 >= iq + 1) group(j0, std::false_type{});
                                J[(size_t)iq * ldj + row] = t2c;
                            }
                        }
                        DSTAMP(14);
                        // the new column of R is d1 with dq below it (d itself is not kept: the
                        // next pass recomputes it).  No barrier: nothing here is read again before
                        // the one that opens the next outer iteration.
                        for (int i = tid; i <= iq; i += T) R[ridx(i, iq)] = i < iq ? d[i] : dq;
                        iq = ufirst(iq + 1);
                        DSTAMP(15);
                        if (fabs(dq) <= EPS * R_norm) { // degenerate (:400-418)
                            __syncthreads();
                            if (tid == 0) iaexcl[ip] = 0;
                            __syncthreads();
                            delete_constraint(ip);
                            for (int i = tid; i < m; i += T) iai[i] = i;
                            __syncthreads();
                            for (int i = tid; i < iq; i += T) {
                                A[i] = A_old[i];
                                u[i] = u_old[i];
                            }
                            __syncthreads();
                            for (int i = tid; i < iq; i += T) iai[A[i]] = -1;
                            for (int i = tid; i < n; i += T) x[i] = x_old[i];
                            restart = true;
                            break;
                        }
                        DSTAMP(7);
                        R_norm = fmax(R_norm, fabs(dq));
                        if (tid == 0) iai[ip] = -1;
                        break; // goto l1
                    }
                    // partial step (:434-446)
                    if (tid == 0) iai[l] = l;
                    __syncthreads();
                    DSTAMP(6);
                    delete_constraint(l);
                    DSTAMP(8);
                    eval_one(ip);
                    DSTAMP(1);
                }
            } while (restart);
        }
        __syncthreads();

        DSTAMP(9);
#ifdef MPCB_STAMPS
        stamp_acc[12] = passes;
        stamp_acc[13] = iters;
        if (tid == 0 && sd.dbg)
            for (int i = 0; i < 16; i++) sd.dbg[(size_t)inst * 16 + i] = stamp_acc[i];
#endif
        // ---- outputs ---------------------------------------------------------
        const bool solved = status == MPCB_OK || status == MPCB_ITER_CAP;
        if (tid == 0) {
            if (out.cost) out.cost[inst] = f;
            if (out.status) out.status[inst] = status;
            if (out.iters) out.iters[inst] = iters;
            if (out.n_active) out.n_active[inst] = iq;
            if (out.alg_flops) out.alg_flops[inst] = falg + (POLY ? 2.0 * mm * nn : 0.0);
        }
        if (out.active_idx || out.multipliers)
            for (int a = tid; a < out.active_stride; a += T) {
                const bool in_set = a < iq;
                if (out.active_idx) out.active_idx[(size_t)inst * out.active_stride + a] = in_set ? A[a] : -1;
                if (out.multipliers) out.multipliers[(size_t)inst * out.active_stride + a] = in_set ? u[a] : 0.0;
            }
        if (!solved)
            for (int i = tid; i < n; i += T) x[i] = 0.0;
        __syncthreads();
        eval_all();
        for (int k = tid; k < N; k += T) {
            zx[k] = ((rc[k] + rs[k]) + rx[k]) + suf[k];
            zy[k] = POLY ? ((rc[N + k] + rs[N + k]) + rx[N + k]) + suf[N + k] : 0.0;
        }
        __syncthreads();
        if constexpr (SLACK) {
            if (out.slacks)
                for (int i = tid; i < 2 * N; i += T) out.slacks[(size_t)inst * 2 * N + i] = x[N + i];
        }
        for (int k = tid; k < N; k += T) {
            const size_t o = (size_t)inst * N + k;
            if (out.jerk_x) out.jerk_x[o] = x[k];
            if (out.jerk_y && POLY) out.jerk_y[o] = x[N + k];
            if (out.zmp_x) out.zmp_x[o] = solved ? zx[k] : 0.0;
            if (out.zmp_y) out.zmp_y[o] = solved ? zy[k] : 0.0;
            if (out.com_x || out.com_y || out.com_xd || out.com_yd) {
                // computeCOMtrajectory: Sx x0 + Su jerk with the Toeplitz first columns
                const double *pr = ws.xpx + (size_t)k * 3, *vr = ws.xvx + (size_t)k * 3;
                double cx = pr[0] * wx[0] + pr[1] * wx[1] + pr[2] * wx[2];
                double cy = pr[0] * wy[0] + pr[1] * wy[1] + pr[2] * wy[2];
                double vx = vr[0] * wx[0] + vr[1] * wx[1] + vr[2] * wx[2];
                double vy = vr[0] * wy[0] + vr[1] * wy[1] + vr[2] * wy[2];
                for (int j = 0; j <= k; j++) {
                    const double cp = ws.xp_col[k - j], cv = ws.xv_col[k - j];
                    cx += cp * x[j];
                    vx += cv * x[j];
                    if constexpr (POLY) {
                        cy += cp * x[N + j];
                        vy += cv * x[N + j];
                    }
                }
                if (out.com_x) out.com_x[o] = solved ? cx : 0.0;
                if (out.com_y) out.com_y[o] = solved ? cy : 0.0;
                if (out.com_xd) out.com_xd[o] = solved ? vx : 0.0;
                if (out.com_yd) out.com_yd[o] = solved ? vy : 0.0;
            }
        }
        if (out.min_margin) {
            // getConstraintViolation (MPCPlanner.cpp:1132-1153): rows of a knot are contiguous
            for (int k = tid; k < N; k += T) {
                double mn = DINF;
                if constexpr (POLY) {
                    const int r0 = sd.knot_row0[k];
                    const int ph = sd.knot_phase[k];
                    const int E = ((ph & 1) && ph < 2 * sd.steps) ? 3 : 4;
                    for (int e = 0; e < E; e++) mn = fmin(mn, s[r0 + e]);
                } else if constexpr (SLACK) {
                    mn = fmin(zx[k] - lo[k], hi[k] - zx[k]);
                } else {
                    mn = fmin(s[k], s[N + k]);
                }
                out.min_margin[(size_t)inst * N + k] = solved ? mn : 0.0;
            }
        }
        __syncthreads();
    }
    if (tid == 0) retire_counters(ws.count, gridDim.x, ws.host_count);
}

__global__ void list_all_kernel(int *list, int *count, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) list[i] = i;
    if (i == 0) *count = B;
}

size_t dense_lds_bytes(const DevSetup &sd, bool ldsj) {
    const size_t N = sd.N, n = sd.nvar, m = sd.m;
    size_t doubles = 5 * n + 3 * (n + 1) + 3 * n + (n + 1) + m + 5 * N + (size_t)sd.nphase * 12 +
                     2 * (size_t)(sd.form == MPCB_COUPLED_REF ? std::max(sd.nphase, sd.N) : sd.nphase) + 2 * RED_D;
    size_t ints = 2 * RED_I + 2 * (n + 1) + 2 * m;
    size_t bytes = doubles * 8 + ints * 4 + 64;
    if (ldsj) bytes += (n * (n | 1) + n * (n + 3) / 2 + 2 + n) * 8;
    return bytes;
}

} // namespace

size_t dense_lds_bytes_needed(const DevSetup &sd, bool *in_lds, bool with_block) {
    if (box_dense_applies(sd)) {
        if (in_lds) *in_lds = true;
        return box_dense_lds_bytes(sd) + 256;
    }
    // 160 KB of LDS per workgroup at most; keep a margin for the static part
    const bool ldsj = dense_lds_bytes(sd, true) <= 160 * 1024 - 256;
    if (in_lds) *in_lds = ldsj;
    size_t bytes = dense_lds_bytes(sd, ldsj);
    if (with_block && !ldsj) bytes = std::max(bytes, block_lds_bytes(sd));
    return bytes + 256;
}

bool block_edition_applies(const DevSetup &sd) {
    if (box_dense_applies(sd) || !block_applies(sd)) return false;
    return !(dense_lds_bytes(sd, true) <= 160 * 1024 - 256); // (LDS-resident dense factors: small problems)
}

hipError_t launch_list_all(const DenseWs &ws, int B, hipStream_t stream) {
    hipLaunchKernelGGL(list_all_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, ws.list, ws.count, B);
    return hipGetLastError();
}

hipError_t launch_dense(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, const DenseWs &ws,
                        int blocks, hipStream_t stream) {
    if (blocks <= 0) return hipSuccess;
    if (box_dense_applies(sd)) return launch_box_dense(sd, in, out, ws, blocks, stream);
    const bool poly = sd.form != MPCB_BOX_1D;
    // 160 KB of LDS per workgroup at most; keep a margin for the static part
    const bool ldsj = dense_lds_bytes(sd, true) <= 160 * 1024 - 256;
    size_t shmem = dense_lds_bytes(sd, ldsj);
    if (ws.block_list && poly && !ldsj) shmem = std::max(shmem, block_lds_bytes(sd)); // its workgroups run the block edition first
    const void *fn;
    if (sd.form == MPCB_BOX_SLACK)
        fn = ldsj ? reinterpret_cast<const void *>(dense_kernel<2, true>)
                  : reinterpret_cast<const void *>(dense_kernel<2, false>);
    else if (poly) fn = ldsj ? reinterpret_cast<const void *>(dense_kernel<1, true>)
                             : reinterpret_cast<const void *>(dense_kernel<1, false>);
    else fn = ldsj ? reinterpret_cast<const void *>(dense_kernel<0, true>)
                   : reinterpret_cast<const void *>(dense_kernel<0, false>);
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    void *args[] = {(void *)&sd, (void *)&in, (void *)&out, (void *)&ws};
    return hipLaunchKernel(fn, dim3(blocks), dim3(T), args, shmem, stream);
}

} // namespace mpcb

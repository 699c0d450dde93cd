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

#include "device_common.h"
#include "kernels.h"

namespace mpcb {
namespace {

constexpr int T = 256;

__device__ __forceinline__ double block_sum(double v, double *red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// minimum of (val, key), smaller key wins ties; result identical in every thread
__device__ __forceinline__ void block_argmin(double &val, int &key, double *red, int *redk, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(val, o, 64);
        const int ok = __shfl_xor(key, o, 64);
        if (ov < val || (ov == val && ok < key)) {
            val = ov;
            key = ok;
        }
    }
    __syncthreads();
    if ((tid & 63) == 0) {
        red[tid >> 6] = val;
        redk[tid >> 6] = key;
    }
    __syncthreads();
    val = red[0];
    key = redk[0];
#pragma unroll
    for (int w = 1; w < 4; w++)
        if (red[w] < val || (red[w] == val && redk[w] < key)) {
            val = red[w];
            key = redk[w];
        }
}

template <bool POLY>
__global__ __launch_bounds__(T) void dense_kernel(DevSetup sd, DevIn in, mpcb_outputs out, DenseWs ws) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = sd.N, n = POLY ? 2 * N : N, m = sd.m;
    // ---- LDS carve-up -------------------------------------------------------
    double *x = smem, *z = x + n, *d = z + n, *np = d + n, *x_old = np + n;
    double *u = x_old + n, *u_old = u + (n + 1), *r = u_old + (n + 1);
    double *rc = r + (n + 1), *rs = rc + n, *rx = rs + n, *suf = rx + n; // rotation cos/sin/xny, suffix norms
    double *s = suf + (n + 1);
    double *zx = s + m, *zy = zx + N;
    double *lo = zy + N, *hi = lo + N;                                   // BOX limits
    double *tab = hi + N;                                                // POLY: [phase][edge][p,q,r]
    double *cref = tab + (POLY ? sd.nphase * 12 : 0);                    // POLY: [phase][x,y]
    double *red = cref + (POLY ? sd.nphase * 2 : 0);
    int *redk = reinterpret_cast<int *>(red + 4);
    int *A = redk + 4, *A_old = A + (n + 1), *iai = A_old + (n + 1);
    int *iaexcl = iai + m;
    __shared__ int sh_i[4];
    __shared__ double sh_d[4];

    const int count = *ws.count;
    for (int entry = blockIdx.x; entry < count; entry += gridDim.x) {
        const int inst = ws.list[entry];
        double *J = ws.J + (size_t)blockIdx.x * n * n;   // column-major n x n
        double *R = ws.R + (size_t)blockIdx.x * n * n;   // column-major, upper triangle used
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
                for (int i = tid; i < 2 * sd.nphase; i += T) cref[i] = in.zref_phase[(size_t)inst * 2 * sd.nphase + i];
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
        // s_i = n_i' x + ci0_i through the ZMP: zx = Zx x0 + Zu x  (never forms CI)
        auto eval_s = [&](int only) {
            for (int k = tid; k < N; k += T) {
                const double *zr = ws.zx + (size_t)k * 3;
                double ax = zr[0] * wx[0] + zr[1] * wx[1] + zr[2] * wx[2];
                double ay = zr[0] * wy[0] + zr[1] * wy[1] + zr[2] * wy[2];
                for (int j = 0; j <= k; j++) {
                    const double c = ws.zu_col[k - j];
                    ax += c * x[j];
                    if constexpr (POLY) ay += c * x[N + j];
                }
                zx[k] = ax;
                zy[k] = ay;
            }
            __syncthreads();
            if (only >= 0) {
                if (tid == 0) {
                    const int k = rowK(only);
                    s[only] = rowP(only, k) * zx[k] + rowQ(only, k) * zy[k] + rowR(only, k);
                }
            } else {
                for (int i = tid; i < m; i += T) {
                    const int k = rowK(i);
                    s[i] = rowP(i, k) * zx[k] + rowQ(i, k) * zy[k] + rowR(i, k);
                }
            }
            __syncthreads();
        };

        // ---- preprocessing (eiquadprog.hpp:162-191) -------------------------
        // J = L^-T = blkdiag(J1, J1); x = -G^-1 g0 ; f = 1/2 g0'x
        for (size_t e = tid; e < (size_t)n * n; e += T) {
            const int col = (int)(e / n), row = (int)(e % n);
            double v = 0.0;
            if (col < N && row < N) v = ws.J1[(size_t)col * N + row];
            else if (col >= N && row >= N) v = ws.J1[(size_t)(col - N) * N + (row - N)];
            J[e] = v;
        }
        for (int i = tid; i < n; i += T) {
            const int k = i < N ? i : i - N;
            const double *kx = sd.Kx + (size_t)k * 4;
            const double *w = i < N ? wx : wy;
            double v = kx[0] * w[0] + kx[1] * w[1] + kx[2] * w[2] + kx[3] * w[3];
            if constexpr (POLY) {
                if (sd.Wp)
                    for (int p2 = 0; p2 < sd.nphase; p2++)
                        v += cref[2 * p2 + (i < N ? 0 : 1)] * sd.Wp[(size_t)p2 * N + k];
            }
            x[i] = v;
        }
        double f;
        {
            double fp = 0.0;
            for (int pb = 0; pb < (sd.F2 ? sd.nphase : 1); pb += T)
                fp += unconstrained_cost_part(sd, wx, cref, 0, pb + tid, pb + tid == 0) +
                      unconstrained_cost_part(sd, wy, cref, 1, pb + tid, pb + tid == 0);
            f = block_sum(fp, red, tid);
        }
        for (int i = tid; i < m; i += T) iai[i] = i;
        double R_norm = 1.0;
        int iq = 0, iters = 0, passes = 0, status = MPCB_OK;
        int ip = 0;
        const double nn = (double)n, mm = (double)m;
        double falg = 2.0 * nn * nn;
        __syncthreads();

        // delete_constraint (eiquadprog.hpp:516-592); u has iq+1 live entries
        auto delete_constraint = [&](int l) {
            if (tid == 0) {
                int qq = 0;
                for (int i = 0; i < iq; i++)
                    if (A[i] == l) {
                        qq = i;
                        break;
                    }
                sh_i[0] = qq;
            }
            __syncthreads();
            const int qq = sh_i[0];
            {
                const double c = (double)(iq - 1 - qq);
                falg += 6.0 * nn * c + 3.0 * c * (c - 1.0);
            }
            // shift A, u and the columns of R
            for (int row = tid; row < iq; row += T)
                for (int i = qq; i < iq - 1; i++) R[(size_t)i * n + row] = R[(size_t)(i + 1) * n + row];
            if (tid == 0) {
                for (int i = qq; i < iq - 1; i++) {
                    A[i] = A[i + 1];
                    u[i] = u[i + 1];
                }
                A[iq - 1] = A[iq];
                u[iq - 1] = u[iq];
                A[iq] = 0;
                u[iq] = 0.0;
            }
            __syncthreads();
            for (int j = tid; j < iq; j += T) R[(size_t)(iq - 1) * n + j] = 0.0;
            iq--;
            __syncthreads();
            if (iq == 0) return;
            // rotations of rows (j, j+1) of R: a true sequential chain
            for (int j = qq; j < iq; j++) {
                if (tid == 0) {
                    double cc = R[(size_t)j * n + j], ss = R[(size_t)j * n + j + 1];
                    const double h = distance(cc, ss);
                    if (h == 0.0) {
                        rc[j] = 1.0;
                        rs[j] = 0.0;
                        rx[j] = DINF; // marks "no rotation"
                    } else {
                        cc /= h;
                        ss /= h;
                        R[(size_t)j * n + j + 1] = 0.0;
                        if (cc < 0.0) {
                            R[(size_t)j * n + j] = -h;
                            cc = -cc;
                            ss = -ss;
                        } else
                            R[(size_t)j * n + j] = h;
                        rc[j] = cc;
                        rs[j] = ss;
                        rx[j] = ss / (1.0 + cc);
                    }
                }
                __syncthreads();
                if (rx[j] != DINF) {
                    const double cc = rc[j], ss = rs[j], xny = rx[j];
                    for (int k = j + 1 + tid; k < iq; k += T) {
                        const double t1 = R[(size_t)k * n + j], t2 = R[(size_t)k * n + j + 1];
                        const double nv = t1 * cc + t2 * ss;
                        R[(size_t)k * n + j] = nv;
                        R[(size_t)k * n + j + 1] = xny * (t1 + nv) - t2;
                    }
                }
                __syncthreads();
            }
            // the same rotations on columns (j, j+1) of J: one thread per row
            for (int row = tid; row < n; row += T) {
                double t1 = J[(size_t)qq * n + row];
                for (int j = qq; j < iq; j++) {
                    const double t2 = J[(size_t)(j + 1) * n + row];
                    if (rx[j] == DINF) {
                        J[(size_t)j * n + row] = t1;
                        t1 = t2;
                    } else {
                        const double nv = t1 * rc[j] + t2 * rs[j];
                        J[(size_t)j * n + row] = nv;
                        t1 = rx[j] * (nv + t1) - t2;
                    }
                }
                J[(size_t)iq * n + row] = t1;
            }
            __syncthreads();
        };

        bool done = false;
        while (!done) { // step 1 (l1)
            iters++;
            if (iters > sd.iter_cap) {
                status = MPCB_ITER_CAP;
                break;
            }
            for (int i = tid; i < iq; i += T) iai[A[i]] = -1;
            for (int i = tid; i < m; i += T) iaexcl[i] = 1;
            eval_s(-1);
            falg += 2.0 * mm * nn;
            double part = 0.0;
            for (int i = tid; i < m; i += T) part += fmin(0.0, s[i]);
            const double psi = block_sum(part, red, tid);
            if (fabs(psi) <= sd.psi_thr) break;
            for (int i = tid; i < iq; i += T) {
                u_old[i] = u[i];
                A_old[i] = A[i];
            }
            for (int i = tid; i < n; i += T) x_old[i] = x[i];
            double ss = 0.0;
            bool restart;
            do { // step 2 (l2)
                restart = false;
                __syncthreads();
                double bv = DINF;
                int bi = 0x7fffffff;
                for (int i = tid; i < m; i += T)
                    if (s[i] < ss && iai[i] != -1 && iaexcl[i] && (s[i] < bv || (s[i] == bv && i < bi))) {
                        bv = s[i];
                        bi = i;
                    }
                block_argmin(bv, bi, red, redk, tid);
                if (bv < ss) {
                    ss = bv;
                    ip = bi;
                }
                if (ss >= 0.0) {
                    done = true;
                    break;
                }
                // np = CI.col(ip)
                {
                    const int k = rowK(ip);
                    const double p = rowP(ip, k), q = rowQ(ip, k);
                    for (int j = tid; j < N; j += T) {
                        const double c = j <= k ? ws.zu_col[k - j] : 0.0;
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
                    passes++;
                    if (passes > sd.iter_cap) {
                        status = MPCB_ITER_CAP;
                        done = true;
                        break;
                    }
                    falg += 2.0 * nn * nn + 2.0 * nn * (nn - iq) + (double)iq * iq;
                    // d = J' np : one wave per column
                    for (int j = wave; j < n; j += 4) {
                        const double *col = J + (size_t)j * n;
                        double acc = 0.0;
                        for (int k = lane; k < n; k += 64) acc += col[k] * np[k];
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
                        if (lane == 0) d[j] = acc;
                    }
                    __syncthreads();
                    // z = J2 d2 : one thread per row
                    for (int row = tid; row < n; row += T) {
                        double acc = 0.0;
                        for (int j = iq; j < n; j++) acc += J[(size_t)j * n + row] * d[j];
                        z[row] = acc;
                    }
                    // r = R^-1 d1 (back substitution, column oriented)
                    for (int i = tid; i < iq; i += T) r[i] = d[i];
                    __syncthreads();
                    for (int k = iq - 1; k >= 0; k--) {
                        if (tid == 0) r[k] = r[k] / R[(size_t)k * n + k];
                        __syncthreads();
                        const double rk = r[k];
                        for (int i = tid; i < k; i += T) r[i] -= R[(size_t)k * n + i] * rk;
                        __syncthreads();
                    }
                    // step lengths
                    double t1 = DINF;
                    int kmin = 0x7fffffff;
                    for (int k = tid; k < iq; k += T)
                        if (r[k] > 0.0) {
                            const double tmp = u[k] / r[k];
                            if (tmp < t1 || (tmp == t1 && k < kmin)) {
                                t1 = tmp;
                                kmin = k;
                            }
                        }
                    block_argmin(t1, kmin, red, redk, tid);
                    const int l = t1 < DINF ? A[kmin] : 0;
                    double pz = 0.0, pn = 0.0;
                    for (int i = tid; i < n; i += T) {
                        pz += z[i] * z[i];
                        pn += z[i] * np[i];
                    }
                    const double zz = block_sum(pz, red, tid);
                    const double zn = block_sum(pn, red, tid);
                    const double t2 = fabs(zz) > EPS ? -s[ip] / zn : DINF;
                    const double t = fmin(t1, t2);
                    if (t >= DINF) {
                        status = MPCB_INFEASIBLE;
                        f = DINF;
                        done = true;
                        break;
                    }
                    __syncthreads();
                    if (t2 >= DINF) { // dual step
                        for (int k = tid; k < iq; k += T) u[k] -= t * r[k];
                        if (tid == 0) {
                            u[iq] += t;
                            iai[l] = l;
                        }
                        __syncthreads();
                        delete_constraint(l);
                        continue;
                    }
                    for (int i = tid; i < n; i += T) x[i] += t * z[i];
                    f += t * zn * (0.5 * t + u[iq]);
                    __syncthreads();
                    for (int k = tid; k < iq; k += T) u[k] -= t * r[k];
                    if (tid == 0) u[iq] += t;
                    __syncthreads();
                    if (t == t2) {
                        // add_constraint (eiquadprog.hpp:450-513)
                        falg += 6.0 * nn * (nn - iq - 1);
                        // suffix norms nu_j = ||d[j..n-1]||
                        if (tid == 0) {
                            double acc = 0.0;
                            for (int j = n - 1; j >= iq; j--) {
                                acc += d[j] * d[j];
                                suf[j] = sqrt(acc);
                            }
                        }
                        __syncthreads();
                        for (int j = iq + 1 + tid; j < n; j += T) {
                            // rotation of (j-1, j): cc = d[j-1], ss = value carried at j
                            const double h = suf[j - 1];
                            if (h == 0.0) {
                                rx[j] = DINF;
                                rc[j] = 1.0;
                                rs[j] = 0.0;
                            } else {
                                double cur = (j == n - 1) ? d[j] : (d[j] < 0.0 ? -suf[j] : suf[j]);
                                double cc = d[j - 1] / h, sv = cur / h;
                                if (cc < 0.0) {
                                    cc = -cc;
                                    sv = -sv;
                                }
                                rc[j] = cc;
                                rs[j] = sv;
                                rx[j] = sv / (1.0 + cc);
                            }
                        }
                        __syncthreads();
                        for (int row = tid; row < n; row += T) {
                            if (n - 1 >= iq + 1) {
                                double t2c = J[(size_t)(n - 1) * n + row];
                                for (int j = n - 1; j >= iq + 1; j--) {
                                    const double t1c = J[(size_t)(j - 1) * n + row];
                                    if (rx[j] == DINF) {
                                        J[(size_t)j * n + row] = t2c;
                                        t2c = t1c;
                                    } else {
                                        const double nv = t1c * rc[j] + t2c * rs[j];
                                        J[(size_t)j * n + row] = rx[j] * (t1c + nv) - t2c;
                                        t2c = nv;
                                    }
                                }
                                J[(size_t)iq * n + row] = t2c;
                            }
                        }
                        __syncthreads();
                        if (tid == 0) {
                            if (n - 1 >= iq + 1 && suf[iq] != 0.0) d[iq] = d[iq] < 0.0 ? -suf[iq] : suf[iq];
                        }
                        __syncthreads();
                        for (int j = iq + 1 + tid; j < n; j += T) d[j] = 0.0;
                        for (int i = tid; i <= iq; i += T) R[(size_t)iq * n + i] = d[i];
                        iq++;
                        __syncthreads();
                        const double dq = d[iq - 1];
                        if (fabs(dq) <= EPS * R_norm) { // degenerate (:400-418)
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
                        R_norm = fmax(R_norm, fabs(dq));
                        if (tid == 0) iai[ip] = -1;
                        break; // goto l1
                    }
                    // partial step (:434-446)
                    if (tid == 0) iai[l] = l;
                    __syncthreads();
                    delete_constraint(l);
                    eval_s(ip);
                }
            } while (restart);
        }
        __syncthreads();

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
        eval_s(-1);
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
                } else {
                    mn = fmin(s[k], s[N + k]);
                }
                out.min_margin[(size_t)inst * N + k] = solved ? mn : 0.0;
            }
        }
        __syncthreads();
    }
}

__global__ void list_all_kernel(int *list, int *count, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) list[i] = i;
    if (i == 0) *count = B;
}

size_t dense_lds_bytes(const DevSetup &sd) {
    const size_t N = sd.N, n = sd.axes * N, m = sd.m;
    size_t doubles = 5 * n + 3 * (n + 1) + 3 * n + (n + 1) + m + 4 * N + (size_t)sd.nphase * 14 + 4;
    size_t ints = 4 + 2 * (n + 1) + 2 * m;
    return doubles * 8 + ints * 4 + 64;
}

} // namespace

hipError_t launch_list_all(const DenseWs &ws, int B, hipStream_t stream) {
    hipLaunchKernelGGL(list_all_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, ws.list, ws.count, B);
    return hipGetLastError();
}

hipError_t launch_dense(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, const DenseWs &ws,
                        int blocks, hipStream_t stream) {
    if (blocks <= 0) return hipSuccess;
    const size_t shmem = dense_lds_bytes(sd);
    const bool poly = sd.form != MPCB_BOX_1D;
    const void *fn = poly ? reinterpret_cast<const void *>(dense_kernel<true>)
                          : reinterpret_cast<const void *>(dense_kernel<false>);
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    if (poly) hipLaunchKernelGGL(dense_kernel<true>, dim3(blocks), dim3(T), shmem, stream, sd, in, out, ws);
    else hipLaunchKernelGGL(dense_kernel<false>, dim3(blocks), dim3(T), shmem, stream, sd, in, out, ws);
    return hipGetLastError();
}

} // namespace mpcb

// Block edition of the multiplier-space dual active-set method: ONE QP PER WORKGROUP of 256
// threads, working sets of up to 160 rows.
//
// The wave-per-QP kernel (zspace_solve.hip) keeps one working-set position per lane, so its two
// editions end at 32 and 60 rows.  A coupled instance that needs more is not ill-conditioned, only
// bigger, and the full method behind them (dense_solve.hip: J = L^-T Q, n x n = 1.28 MB per
// instance in HBM, rotated on every pass) pays ~0.1 ms per pass for it.  This routine runs the
// same multiplier-space method -- same pivot rules, tolerances and step formulas as
// eiquadprog.hpp:130-447, on the multipliers of the working set only, see the head of
// zspace_solve.hip -- with a thread per knot and a thread per working-set position, the factor
// T = R^-1 packed in LDS (160 x 161 / 2 doubles) and the response tables read through L2.  It is
// called by the dense kernel's workgroups for the instances the wide edition could not hold
// (DenseWs::block_list) before they fall back to the full method, which still takes what exceeds
// 160 rows or loses the reduced system's accuracy.
#pragma once
#include <type_traits>

#include "device_common.h"
#include "kernels.h"

namespace mpcb {
namespace {

constexpr int BQ = 160;                // rows the packed factor holds
constexpr int BT_ = 256;               // threads of the workgroup
constexpr int BW_ = BT_ / 64;          // its waves
constexpr int BTRI = BQ * (BQ + 1) / 2;

__device__ __forceinline__ int btri(int c) { return c * (c + 1) / 2; }

// Reductions over the workgroup.  Per-wave partials go through one of two small LDS buffers taken in
// turn, so that a wave still reading the previous reduction's buffer is not overtaken by a wave
// writing this one's: one barrier per reduction.
struct BlockRed {
    double *d; // [2][4 * BW_]
    int *k;    // [2][BW_]
    int flip;
};

__device__ __forceinline__ double blk_sum(double v, BlockRed &r, int tid) {
    v = wave_sum(v);
    double *b = r.d + r.flip * 4 * BW_;
    r.flip ^= 1;
    if ((tid & 63) == 0) b[tid >> 6] = v;
    __syncthreads();
    double tot = b[0];
#pragma unroll
    for (int w = 1; w < BW_; w++) tot += b[w];
    return tot;
}

// minimum of (val, key), the smaller key wins ties, and the sum of a, in one exchange; results
// identical in every thread
__device__ __forceinline__ void blk_argmin_sum(double &val, int &key, double &a, BlockRed &r, int tid) {
    int aux = 0;
    wave_argmin(val, key, aux);
    a = wave_sum(a);
    double *bd = r.d + r.flip * 4 * BW_;
    int *bk = r.k + r.flip * BW_;
    r.flip ^= 1;
    if ((tid & 63) == 0) {
        bd[tid >> 6] = val;
        bk[tid >> 6] = key;
        bd[BW_ + (tid >> 6)] = a;
    }
    __syncthreads();
    val = bd[0];
    key = bk[0];
    a = bd[BW_];
#pragma unroll
    for (int w = 1; w < BW_; w++) {
        if (bd[w] < val || (bd[w] == val && bk[w] < key)) {
            val = bd[w];
            key = bk[w];
        }
        a += bd[BW_ + w];
    }
}

// inclusive prefix sum over the threads of the workgroup
__device__ __forceinline__ double blk_scan(double v, BlockRed &r, int tid) {
    v = wave_scan(v);
    double *b = r.d + r.flip * 4 * BW_;
    r.flip ^= 1;
    if ((tid & 63) == 63) b[tid >> 6] = v;
    __syncthreads();
    double before = 0.0;
#pragma unroll
    for (int w = 0; w < BW_ - 1; w++)
        if (w < (tid >> 6)) before += b[w];
    return v + before;
}

// LDS bytes the routine needs for this shape.
__host__ __device__ inline size_t block_lds_bytes(const DevSetup &sd) {
    const size_t nseg = (size_t)(sd.nphase > sd.nref ? sd.nphase : sd.nref);
    const size_t doubles = BTRI + 12 * BQ + 2 * BT_ + 2 * 4 * BW_ + (size_t)(sd.nphase + 1) * 12 + 2 * nseg;
    const size_t ints = 4 * BQ + 2 * BW_ + 4;
    return doubles * 8 + ints * 4;
}

// Whether the routine covers this shape: a coupled form whose knots fit one workgroup.
__host__ __device__ inline bool block_applies(const DevSetup &sd) {
    return (sd.form == MPCB_COUPLED_JERK || sd.form == MPCB_COUPLED_VEL || sd.form == MPCB_COUPLED_REF) &&
           sd.N <= BT_ && block_lds_bytes(sd) <= 160 * 1024 - 1024;
}

// Solves instance `inst` from the start.  Returns true when it is finished (outputs written; the
// status may be OK, INFEASIBLE or ITER_CAP) and false when the instance has to go to the full
// method (more than BQ rows, or the reduced system lost its accuracy); nothing is written then.
// Every thread of the workgroup must call it; the result is the same in all of them.
__device__ __forceinline__ bool block_solve(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, const DenseWs &ws,
                                         const int inst, double *smem) {
    const int tid = threadIdx.x;
    const int N = sd.N, m = sd.m;
    const int k = tid, kc = k < N ? k : N - 1;
    const bool kv = k < N;
    // ---- LDS ---------------------------------------------------------------------------------
    double *Tp = smem;                 // T = R^-1, packed upper triangle: T(i, c) at btri(c) + i, i <= c
    double *U = Tp + BTRI;             // working set: multiplier, p, q
    double *Pp = U + BQ, *Qq = Pp + BQ;
    double *oldU = Qq + BQ;            // copy taken at step 1 (eiquadprog.hpp:275-277)
    double *cx = oldU + BQ, *cy = cx + BQ;   // row coefficients of the response sums
    double *gv = cy + BQ, *dv = gv + BQ;     // g = N_A' G^-1 n_ip, d = T' g
    double *dg = dv + BQ, *dir = dg + BQ;    // drop: row qq of T, reciprocal prefix norms
    double *ca = dir + BQ, *cb = ca + BQ;    // drop: coefficients of the closed form
    double *jxs = cb + BQ, *jys = jxs + BT_; // the jerk by knot (epilogue)
    double *redd = jys + BT_;
    double *tab = redd + 2 * 4 * BW_;  // [phase][edge][p, q, r]
    double *cref = tab + (size_t)(sd.nphase + 1) * 12;
    const int nseg = sd.nphase > sd.nref ? sd.nphase : sd.nref;
    int *Row = reinterpret_cast<int *>(cref + 2 * nseg);
    int *KS = Row + BQ, *oldRow = KS + BQ, *oldKS = oldRow + BQ; // constraint row; (knot << 2) | slot
    int *redk = oldKS + BQ;
    BlockRed red = {redd, redk, 0};

    __syncthreads(); // whoever used this LDS before is done with it
    // ---- inputs, constraint rows -------------------------------------------------------------
    double wx[4], wy[4];
    {
        const double *px = in.x0 + (size_t)inst * 3, *py = in.y0 + (size_t)inst * 3;
        wx[0] = px[0];
        wx[1] = px[1];
        wx[2] = px[2];
        wy[0] = py[0];
        wy[1] = py[1];
        wy[2] = py[2];
        wx[3] = in.speed[(size_t)inst * 2];
        wy[3] = in.speed[(size_t)inst * 2 + 1];
    }
    // A CoM height per instance (in.height): the response tables are combined from their height-free
    // parts as they are read, S = Spp - c Sps + c^2 Saa, W^T = Wtp - c Wta, Kz = Kzp - c Kza with
    // c = height / gravity, in the same arithmetic as the wave-per-QP kernel.
    const bool hv = in.height != nullptr;
    const double ch = hv ? in.height[inst] * sd.inv_gravity : 0.0, ch2 = ch * ch;
    auto S_at = [&](int r, int c) -> double {
        const size_t o = (size_t)r * N + c;
        return hv ? fma(ch2, sd.Saa[o], fma(-ch, sd.Sps[o], sd.Spp[o])) : sd.S[o];
    };
    auto W_at = [&](int r, int c) -> double {
        const size_t o = (size_t)r * N + c;
        return hv ? fma(-ch, sd.Wta[o], sd.Wtp[o]) : sd.Wt[o];
    };
    const double *rows = in.rows_pqr ? in.rows_pqr + (size_t)inst * m * 3 : nullptr;
    if (!rows)
        for (int pb = 0; pb < sd.nphase; pb += BT_) polygon_phase(sd, in, inst, wx[3], wy[3], pb + tid, tab, cref);
    __syncthreads();
    if (in.zref_phase)
        for (int i = tid; i < 2 * sd.nref; i += BT_) cref[i] = in.zref_phase[(size_t)inst * 2 * sd.nref + i];
    __syncthreads();
    // the (at most four) rows of this thread's knot; slots that do not exist evaluate to +inf
    const int ph = sd.knot_phase[kc], row0 = sd.knot_row0[kc];
    unsigned valid = 0;
    if (kv) valid = (((ph & 1) && ph < 2 * sd.steps) ? 7u : 15u);
    double sp[4], sq[4], sr[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (rows) {
            const int r = row0 + e < m ? row0 + e : m - 1;
            sp[e] = rows[r * 3];
            sq[e] = rows[r * 3 + 1];
            sr[e] = rows[r * 3 + 2];
        } else if (valid >> e & 1) {
            sp[e] = tab[(ph * 4 + e) * 3];
            sq[e] = tab[(ph * 4 + e) * 3 + 1];
            sr[e] = tab[(ph * 4 + e) * 3 + 2];
        } else {
            sp[e] = sq[e] = 0.0;
            sr[e] = DINF;
        }
    }
    auto row_p = [&](int ks) -> double { // (p, q) of row (knot, slot): wave-uniform arguments
        const int kk = ks >> 2, e = ks & 3;
        return rows ? rows[(sd.knot_row0[kk] + e) * 3] : tab[(sd.knot_phase[kk] * 4 + e) * 3];
    };
    auto row_q = [&](int ks) -> double {
        const int kk = ks >> 2, e = ks & 3;
        return rows ? rows[(sd.knot_row0[kk] + e) * 3 + 1] : tab[(sd.knot_phase[kk] * 4 + e) * 3 + 1];
    };

    // ---- unconstrained solution (x = -G^-1 g0, eiquadprog.hpp:188-191) ------------------------
    double zxu, zyu;
    {
        double kz[4];
#pragma unroll
        for (int a = 0; a < 4; a++)
            kz[a] = hv ? fma(-ch, sd.Kza[(size_t)kc * 4 + a], sd.Kzp[(size_t)kc * 4 + a]) : sd.Kz[(size_t)kc * 4 + a];
        zxu = kz[0] * wx[0] + kz[1] * wx[1] + kz[2] * wx[2] + kz[3] * wx[3];
        zyu = kz[0] * wy[0] + kz[1] * wy[1] + kz[2] * wy[2] + kz[3] * wy[3];
        if (sd.Sp)
            for (int p = 0; p < sd.nref; p++) {
                const double s = sd.Sp[(size_t)p * N + kc];
                zxu += cref[2 * p] * s;
                zyu += cref[2 * p + 1] * s;
            }
    }
    double f;
    {
        double fp = 0.0;
        for (int pb = 0; pb < (sd.F2 ? sd.nref : 1); pb += BT_)
            fp += unconstrained_cost_part(sd, wx, cref, 0, pb + tid, pb + tid == 0) +
                  unconstrained_cost_part(sd, wy, cref, 1, pb + tid, pb + tid == 0);
        f = blk_sum(fp, red, tid);
    }
    // stopping threshold m * eps * c1 * c2 * 100 (eiquadprog.hpp:267); empty slots of caller-built
    // rows ((0, 0, +inf)) are not rows of the caller's A
    double psi_thr = sd.psi_thr;
    if (rows) {
        int mine = 0;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if ((valid >> e & 1) && sr[e] < DINF) mine++;
        const int m_eff = (int)blk_sum((double)mine, red, tid);
        psi_thr = sd.psi_thr * ((double)m_eff / (double)sd.m);
    }

    // ---- dual active-set loop ----------------------------------------------------------------
    int iq = 0, iters = 0, passes = 0, status = MPCB_OK;
    const long long nn = 2LL * N, mm = m;
    long long falg = 2 * nn * nn; // operation count of the dense method along this pivot sequence
    unsigned actm = 0, exclm = 0xf;
    double R_norm = 1.0;
    double zx = 0.0, zy = 0.0;
    bool pending = false, zmp_current = false, to_dense = false;
    int ipRow = 0, ipKS = 0;
    double pip = 0.0, qip = 0.0, ucand = 0.0, s_ip = 0.0;

    // zmp (or, with the W table, jerk) response to the first n working-set rows with the
    // coefficients in cx, cy, added to (vx, vy)
    // (sixteen rows' loads in flight per thread: with one workgroup on the compute unit there is
    // nothing else to hide the trip to L2 behind)
    auto accum_rows = [&](auto wtag, int n, double &vx, double &vy) {
        constexpr bool WT = decltype(wtag)::value; // the jerk response W^T instead of the ZMP response S
        auto M_at = [&](int r) -> double { return WT ? W_at(r, kc) : S_at(r, kc); };
        int a = 0;
        for (; a + 16 <= n; a += 16) {
            double s[16];
#pragma unroll
            for (int j = 0; j < 16; j++) s[j] = M_at(KS[a + j] >> 2);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                vx += cx[a + j] * s[j];
                vy += cy[a + j] * s[j];
            }
        }
        for (; a + 4 <= n; a += 4) {
            double s[4];
#pragma unroll
            for (int j = 0; j < 4; j++) s[j] = M_at(KS[a + j] >> 2);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                vx += cx[a + j] * s[j];
                vy += cy[a + j] * s[j];
            }
        }
        for (; a < n; a++) {
            const double s = M_at(KS[a] >> 2);
            vx += cx[a] * s;
            vy += cy[a] * s;
        }
    };
    // Drop working-set position qq (delete_constraint, eiquadprog.hpp:516-592) in the closed form of
    // zspace_solve.hip's drop_constraint: with g_j = T(qq, j), the prefix norms rho_j = ||g[qq..j]||
    // and the prefix sums Q_j = sum_{c = qq..j} g_c col_c,
    //     new col_j = g_{j+1} / (rho_{j+1} rho_j) Q_j - (rho_j / rho_{j+1}) col_{j+1},  qq <= j < iq - 1;
    // row qq and the last column go.  A thread carries one row of T along the columns, four columns
    // per barrier (a column is read by every row before anything is stored over it).
    auto drop_constraint = [&](int qq) {
        const int last = iq - 1;
        {
            const int ksd = KS[qq];
            if (k == (ksd >> 2)) actm &= ~(1u << (ksd & 3));
        }
        const bool mv = tid > qq && tid < iq;
        double tu = 0, tp = 0, tq = 0;
        int trow = 0, tks = 0;
        if (mv) {
            tu = U[tid];
            tp = Pp[tid];
            tq = Qq[tid];
            trow = Row[tid];
            tks = KS[tid];
        }
        double g = 0.0;
        if (qq < last && tid >= qq && tid <= last) g = Tp[btri(tid) + qq];
        const double pfx = blk_scan(g * g, red, tid); // (its barrier also orders the reads above)
        if (mv) {
            U[tid - 1] = tu;
            Pp[tid - 1] = tp;
            Qq[tid - 1] = tq;
            Row[tid - 1] = trow;
            KS[tid - 1] = tks;
        }
        if (qq < last) {
            double rho = 0.0, irho = 0.0;
            sqrt_rsqrt(pfx > 0.0 ? pfx : 1.0, rho, irho);
            rho = pfx > 0.0 ? rho : 0.0;
            irho = pfx > 0.0 ? irho : 0.0;
            if (tid < BQ) {
                dg[tid] = g;
                dir[tid] = irho;
            }
            __syncthreads();
            if (tid < BQ) {
                const bool cj = tid >= qq && tid < last;
                const double gn = cj ? dg[tid + 1] : 0.0, irn = cj ? dir[tid + 1] : 0.0;
                ca[tid] = cj ? gn * (irn * irho) : 0.0;
                cb[tid] = cj ? -rho * irn : 0.0;
            }
            __syncthreads();
            const int i = tid;
            const bool ri = i < iq;
            const int inew = i < qq ? i : i - 1;
            const bool wr = ri && i != qq;
            double cur = (ri && i <= qq) ? Tp[btri(qq) + i] : 0.0, q = 0.0;
            for (int jb = qq; jb < last; jb += 4) {
                double nx[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int j = jb + e;
                    nx[e] = (j < last && ri && i <= j + 1) ? Tp[btri(j + 1) + i] : 0.0;
                }
                __syncthreads(); // columns jb + 1 .. jb + 4 are read by every row before columns jb .. are stored
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int j = jb + e;
                    if (j < last) {
                        q = fma(dg[j], cur, q);
                        const double o = fma(ca[j], q, cb[j] * nx[e]);
                        if (wr && inew <= j) Tp[btri(j) + inew] = o;
                        cur = nx[e];
                    }
                }
            }
        }
        iq--;
        __syncthreads();
    };

    for (;;) { // step 1 (l1)
        iters++;
        if (iters > sd.iter_cap) {
            status = MPCB_ITER_CAP;
            break;
        }
        // zmp at x = x_unc + G^-1 N_A u
        if (tid < iq) {
            cx[tid] = U[tid] * Pp[tid];
            cy[tid] = U[tid] * Qq[tid];
        }
        __syncthreads();
        zx = zxu;
        zy = zyu;
        accum_rows(std::false_type{}, iq, zx, zy);
        // s(x) for every row (:255-261) and the sum of the violations
        double sv[4], psi = 0.0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            sv[e] = (valid >> e & 1) ? fma(sp[e], zx, fma(sq[e], zy, sr[e])) : DINF;
            psi += fmin(0.0, sv[e]);
        }
        psi = blk_sum(psi, red, tid);
        falg += 2 * mm * nn;
        exclm = 0xf; // iaexcl[i] = true (:257)
        if (fabs(psi) <= psi_thr) { // :267-272
            zmp_current = true;
            break;
        }
        if (tid < iq) { // :275-277
            oldU[tid] = U[tid];
            oldRow[tid] = Row[tid];
            oldKS[tid] = KS[tid];
        }
        double ssv = 0.0;
        bool done = false, restart, restart_seen = false;
        do { // step 2 (l2), :279-299: the most violated row that is neither active nor excluded,
             // the lowest row on ties (rows ascend with (knot, edge))
            restart = false;
            double bv = DINF, dummy = 0.0;
            int bks = 0x7fffffff;
            {
                const unsigned elig = valid & ~actm & exclm;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if ((elig >> e & 1) && sv[e] < ssv && sv[e] < bv) {
                        bv = sv[e];
                        bks = (k << 2) | e;
                    }
            }
            blk_argmin_sum(bv, bks, dummy, red, tid);
            if (bv < ssv) {
                ssv = bv;
                ipKS = bks;
                ipRow = sd.knot_row0[bks >> 2] + (bks & 3);
                s_ip = bv;
            }
            if (ssv >= 0.0) {
                zmp_current = !restart_seen; // after a degenerate restart u was restored, zx is stale
                done = true;
                break;
            }
            const int kip = ipKS >> 2, eip = ipKS & 3;
            pip = row_p(ipKS);
            qip = row_q(ipKS);
            ucand = 0.0; // u(iq) = 0 (:297)
            pending = true;
            const double skk = S_at(kip, kip);
            const double mii = (pip * pip + qip * qip) * skk; // n_ip' G^-1 n_ip

            for (;;) { // step 2a (l2a), :306-446
                passes++;
                if (passes > sd.iter_cap) {
                    status = MPCB_ITER_CAP;
                    done = true;
                    break;
                }
                falg += 2 * nn * nn + 2 * nn * (nn - iq) + (long long)iq * iq;
                // g = N_A' G^-1 n_ip ; d = R^-T g (= J1' n_ip) ; r = R^-1 d
                if (tid < iq) gv[tid] = (Pp[tid] * pip + Qq[tid] * qip) * S_at(kip, KS[tid] >> 2);
                __syncthreads();
                double da = 0.0;
                // (eight terms' loads are issued before the first of them is used: taken one by one each
                // term would wait out the LDS latency)
                if (tid < iq) {
                    const double *col = Tp + btri(tid);
                    int j = 0;
                    for (; j + 8 <= tid + 1; j += 8) {
                        double t[8], g[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            t[e] = col[j + e];
                            g[e] = gv[j + e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; e++) da = fma(t[e], g[e], da);
                    }
                    for (; j <= tid; j++) da = fma(col[j], gv[j], da);
                    dv[tid] = da;
                }
                __syncthreads();
                double ra = 0.0, ua = 0.0;
                if (tid < iq) {
                    int c = tid;
                    for (; c + 8 <= iq; c += 8) {
                        double t[8], d8[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            t[e] = Tp[btri(c + e) + tid];
                            d8[e] = dv[c + e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; e++) ra = fma(t[e], d8[e], ra);
                    }
                    for (; c < iq; c++) ra = fma(Tp[btri(c) + tid], dv[c], ra);
                    ua = U[tid];
                }
                // z . n_ip and step 2b: t1 = min u_k / r_k over r_k > 0, first minimum wins (:322-334)
                double dd = da * da;
                double t1 = (tid < iq && ra > 0.0) ? ua / ra : DINF;
                int apos = tid < iq ? tid : 0x7fffffff;
                blk_argmin_sum(t1, apos, dd, red, tid);
                if (apos == 0x7fffffff) apos = 0;
                const double zn = mii - dd;
                double rho = 0.0, irho = 0.0;
                if (zn > 0.0) sqrt_rsqrt(zn, rho, irho);
                // z != 0 test (:336): ||z||^2 >= zn / lambda_max(G1) >= zn / trace(G1)
                bool znz;
                if (zn > 1e-5 * mii && zn > 64.0 * EPS * sd.traceG1) {
                    znz = true;
                } else {
                    // explicit z = G^-1 (n_ip - N_A r)
                    __syncthreads();
                    if (tid < iq) {
                        cx[tid] = -ra * Pp[tid];
                        cy[tid] = -ra * Qq[tid];
                    }
                    __syncthreads();
                    const double w = W_at(kip, kc);
                    double zvx = pip * w, zvy = qip * w;
                    accum_rows(std::true_type{}, iq, zvx, zvy);
                    znz = fabs(blk_sum(kv ? zvx * zvx + zvy * zvy : 0.0, red, tid)) > EPS;
                }
                if (znz && !(zn > 1e-7 * mii)) {
                    // n_ip is numerically dependent on the working set although z is not negligible:
                    // the reduced system has lost its accuracy
                    to_dense = true;
                    done = true;
                    break;
                }
                const double t2 = znz ? -s_ip * irho * irho : DINF;
                const double t = fmin(t1, t2);
                if (t >= DINF) { // :350-356
                    status = MPCB_INFEASIBLE;
                    f = DINF;
                    done = true;
                    break;
                }
                if (t2 >= DINF) { // dual step only, drop (:358-373)
                    if (tid < iq) U[tid] = ua - t * ra;
                    ucand += t;
                    {
                        const long long cdrop = iq - 1 - apos;
                        falg += 6 * nn * cdrop + 3 * cdrop * (cdrop - 1);
                    }
                    __syncthreads();
                    drop_constraint(apos);
                    continue;
                }
                // step in primal and dual space (:377-382)
                f += t * zn * (0.5 * t + ucand);
                if (tid < iq) U[tid] = ua - t * ra;
                ucand += t;
                if (t == t2) { // full step: add the constraint (:392-426, add_constraint :450-513)
                    if (iq >= (ws.block_rows < BQ ? ws.block_rows : BQ)) {
                        to_dense = true;
                        done = true;
                        break;
                    }
                    falg += 6 * nn * (nn - iq - 1);
                    if (rho <= EPS * R_norm) {
                        // degenerate: exclude ip, restore u and A of step 1, choose again
                        if (k == kip) exclm &= ~(1u << eip);
                        __syncthreads();
                        if (tid < iq) {
                            U[tid] = oldU[tid];
                            Row[tid] = oldRow[tid];
                            KS[tid] = oldKS[tid];
                            Pp[tid] = row_p(oldKS[tid]);
                            Qq[tid] = row_q(oldKS[tid]);
                        }
                        __syncthreads();
                        actm = 0;
                        for (int a = 0; a < iq; a++) {
                            const int ks = KS[a];
                            if (k == (ks >> 2)) actm |= 1u << (ks & 3);
                        }
                        pending = false;
                        restart = true;
                        restart_seen = true;
                        break;
                    }
                    R_norm = fmax(R_norm, rho);
                    // new column of T = [-r; 1] / rho
                    if (tid < iq) {
                        Tp[btri(iq) + tid] = -ra * irho;
                    } else if (tid == iq) {
                        Tp[btri(iq) + iq] = irho;
                        U[iq] = ucand;
                        Pp[iq] = pip;
                        Qq[iq] = qip;
                        Row[iq] = ipRow;
                        KS[iq] = ipKS;
                    }
                    if (k == kip) actm |= 1u << eip;
                    iq++;
                    pending = false;
                    __syncthreads();
                    break; // goto l1
                }
                // partial step: drop the blocking constraint and try again (:434-446)
                s_ip += t * zn;
                {
                    const long long cdrop = iq - 1 - apos;
                    falg += 6 * nn * cdrop + 3 * cdrop * (cdrop - 1);
                }
                __syncthreads();
                drop_constraint(apos);
            }
        } while (restart);
        if (done) break;
    }
    if (to_dense) return false;

    // ---- outputs -----------------------------------------------------------------------------
    const bool solved = status == MPCB_OK || status == MPCB_ITER_CAP;
    if (tid == 0) {
        if (out.cost) out.cost[inst] = f;
        if (out.status) out.status[inst] = status;
        if (out.iters) out.iters[inst] = iters;
        if (out.n_active) out.n_active[inst] = iq;
        if (out.alg_flops) out.alg_flops[inst] = (double)(falg + 2 * mm * nn);
    }
    __syncthreads();
    if (out.active_idx || out.multipliers) {
        for (int a = tid; a < out.active_stride; a += BT_) {
            const bool in_set = a < iq;
            if (out.active_idx) out.active_idx[(size_t)inst * out.active_stride + a] = in_set ? Row[a] : -1;
            if (out.multipliers) out.multipliers[(size_t)inst * out.active_stride + a] = in_set ? U[a] : 0.0;
        }
    }
    // multipliers -> trajectories through the response matrices
    int nresp = iq;
    if (tid < iq) {
        cx[tid] = U[tid] * Pp[tid];
        cy[tid] = U[tid] * Qq[tid];
    }
    if (pending && status == MPCB_ITER_CAP && iq < BQ) {
        // partial steps of an unfinished candidate are part of x
        if (tid == 0) {
            cx[iq] = ucand * pip;
            cy[iq] = ucand * qip;
            KS[iq] = ipKS;
        }
        nresp = iq + 1;
    }
    __syncthreads();
    const size_t o = (size_t)inst * N + kc;
    const bool want_com = out.com_x || out.com_y || out.com_xd || out.com_yd;
    if (out.jerk_x || out.jerk_y || want_com) {
        const double *k4 = sd.Kx + (size_t)kc * 4;
        double jx = k4[0] * wx[0] + k4[1] * wx[1] + k4[2] * wx[2] + k4[3] * wx[3];
        double jy = k4[0] * wy[0] + k4[1] * wy[1] + k4[2] * wy[2] + k4[3] * wy[3];
        if (sd.Wp)
            for (int p = 0; p < sd.nref; p++) {
                const double w = sd.Wp[(size_t)p * N + kc];
                jx += cref[2 * p] * w;
                jy += cref[2 * p + 1] * w;
            }
        accum_rows(std::true_type{}, nresp, jx, jy);
        if (kv) {
            if (out.jerk_x) out.jerk_x[o] = solved ? jx : 0.0;
            if (out.jerk_y) out.jerk_y[o] = solved ? jy : 0.0;
        }
        if (want_com) {
            // computeCOMtrajectory (MPCPlanner.cpp:196-209): Sx x0 + Su jerk with the Toeplitz columns
            jxs[tid] = kv ? jx : 0.0;
            jys[tid] = kv ? jy : 0.0;
            __syncthreads();
            if (kv) {
                const double *pr = ws.xpx + (size_t)k * 3, *vr = ws.xvx + (size_t)k * 3;
                double px = pr[0] * wx[0] + pr[1] * wx[1] + pr[2] * wx[2];
                double py = pr[0] * wy[0] + pr[1] * wy[1] + pr[2] * wy[2];
                double vx = vr[0] * wx[0] + vr[1] * wx[1] + vr[2] * wx[2];
                double vy = vr[0] * wy[0] + vr[1] * wy[1] + vr[2] * wy[2];
                for (int j = 0; j <= k; j++) {
                    const double cp = ws.xp_col[k - j], cv = ws.xv_col[k - j];
                    px = fma(cp, jxs[j], px);
                    py = fma(cp, jys[j], py);
                    vx = fma(cv, jxs[j], vx);
                    vy = fma(cv, jys[j], vy);
                }
                if (out.com_x) out.com_x[o] = solved ? px : 0.0;
                if (out.com_y) out.com_y[o] = solved ? py : 0.0;
                if (out.com_xd) out.com_xd[o] = solved ? vx : 0.0;
                if (out.com_yd) out.com_yd[o] = solved ? vy : 0.0;
            }
        }
    }
    if (out.zmp_x || out.zmp_y || out.min_margin) {
        // on a regular exit zx, zy were evaluated at the final multipliers in step 1
        if (!zmp_current) {
            zx = zxu;
            zy = zyu;
            accum_rows(std::false_type{}, nresp, zx, zy);
        }
        if (kv) {
            if (out.zmp_x) out.zmp_x[o] = solved ? zx : 0.0;
            if (out.zmp_y) out.zmp_y[o] = solved ? zy : 0.0;
            if (out.min_margin) {
                // getConstraintViolation (MPCPlanner.cpp:1132-1153)
                double mn = DINF;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (valid >> e & 1) mn = fmin(mn, sp[e] * zx + sq[e] * zy + sr[e]);
                out.min_margin[o] = solved ? mn : 0.0;
            }
        }
    }
    __syncthreads();
    return true;
}

} // namespace
} // namespace mpcb

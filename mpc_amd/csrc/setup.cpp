// Host side of the instance-independent tables: the layout of one QP (sizes, row bookkeeping),
// the inputs of the device-side table construction (tables.hip), and a long-double construction
// of the same tables on the host -- the checker of the device one (mpcb_params.table_build).
//
// What the reference rebuilds densely on every call -- Zx/Zu, Xvx/Xvu
// (MPCPlanner::buildMatrix, MPCPlanner.cpp:101-156), the Hessian blocks
// Gjerk + Gv (+ Gref) (:523-559, :723-739) and, inside the solver, the Cholesky
// factor and J = L^-T (eiquadprog.hpp:165-176) -- depends only on
// (N, Ts, g, height, weights, replanning window), so it is computed once per handle.
#include "setup.h"

#include <cmath>
#include <cstring>

namespace mpcb {
namespace {

typedef long double ld;
typedef std::vector<ld> LVec;

struct Pred {
    LVec sx;   // N x 3, row i = C A^(i+1)
    LVec col;  // first column of the lower-triangular Toeplitz input matrix
};

// Same iteration as buildMatrix (MPCPlanner.cpp:113-153): rows C*A^(i+1), and
// sub-diagonal k of the input matrix = C*A^k*B.
Pred prediction(const ld C[3], ld Ts, int N) {
    const ld A[3][3] = {{1, Ts, Ts * Ts / 2}, {0, 1, Ts}, {0, 0, 1}};
    const ld Bv[3] = {Ts * Ts * Ts / 6, Ts * Ts / 2, Ts};
    Pred p;
    p.sx.assign((size_t)N * 3, 0);
    p.col.assign(N, 0);
    ld row[3], t[3];
    for (int j = 0; j < 3; j++) row[j] = C[0] * A[0][j] + C[1] * A[1][j] + C[2] * A[2][j];
    p.col[0] = C[0] * Bv[0] + C[1] * Bv[1] + C[2] * Bv[2];
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < 3; j++) p.sx[(size_t)i * 3 + j] = row[j];
        if (i + 1 < N) p.col[i + 1] = row[0] * Bv[0] + row[1] * Bv[1] + row[2] * Bv[2];
        for (int j = 0; j < 3; j++) t[j] = row[0] * A[0][j] + row[1] * A[1][j] + row[2] * A[2][j];
        row[0] = t[0];
        row[1] = t[1];
        row[2] = t[2];
    }
    return p;
}

// Y = T X, T lower-triangular Toeplitz with first column col; X is N x c row-major.
LVec toep_mul(const LVec &col, const LVec &X, int N, int c) {
    LVec Y((size_t)N * c, 0);
    for (int i = 0; i < N; i++)
        for (int k = 0; k <= i; k++) {
            const ld t = col[i - k];
            const ld *x = &X[(size_t)k * c];
            ld *y = &Y[(size_t)i * c];
            for (int j = 0; j < c; j++) y[j] += t * x[j];
        }
    return Y;
}

// Y = T^T X
LVec toepT_mul(const LVec &col, const LVec &X, int N, int c) {
    LVec Y((size_t)N * c, 0);
    for (int i = 0; i < N; i++)
        for (int k = i; k < N; k++) {
            const ld t = col[k - i];
            const ld *x = &X[(size_t)k * c];
            ld *y = &Y[(size_t)i * c];
            for (int j = 0; j < c; j++) y[j] += t * x[j];
        }
    return Y;
}

// In-place lower Cholesky of a row-major N x N matrix; false if not positive definite.
bool cholesky(LVec &G, int N) {
    for (int j = 0; j < N; j++) {
        ld d = G[(size_t)j * N + j];
        for (int k = 0; k < j; k++) d -= G[(size_t)j * N + k] * G[(size_t)j * N + k];
        if (!(d > 0)) return false;
        d = sqrtl(d);
        G[(size_t)j * N + j] = d;
        for (int i = j + 1; i < N; i++) {
            ld s = G[(size_t)i * N + j];
            for (int k = 0; k < j; k++) s -= G[(size_t)i * N + k] * G[(size_t)j * N + k];
            G[(size_t)i * N + j] = s / d;
        }
    }
    return true;
}

// Solve (L L^T) X = B in place for an N x c row-major right-hand side.
void chol_solve(const LVec &L, LVec &B, int N, int c) {
    for (int i = 0; i < N; i++) {
        ld *bi = &B[(size_t)i * c];
        for (int k = 0; k < i; k++) {
            const ld l = L[(size_t)i * N + k];
            const ld *bk = &B[(size_t)k * c];
            for (int j = 0; j < c; j++) bi[j] -= l * bk[j];
        }
        const ld inv = 1 / L[(size_t)i * N + i];
        for (int j = 0; j < c; j++) bi[j] *= inv;
    }
    for (int i = N - 1; i >= 0; i--) {
        ld *bi = &B[(size_t)i * c];
        for (int k = i + 1; k < N; k++) {
            const ld l = L[(size_t)k * N + i];
            const ld *bk = &B[(size_t)k * c];
            for (int j = 0; j < c; j++) bi[j] -= l * bk[j];
        }
        const ld inv = 1 / L[(size_t)i * N + i];
        for (int j = 0; j < c; j++) bi[j] *= inv;
    }
}

std::vector<double> to_double(const LVec &v) {
    std::vector<double> d(v.size());
    for (size_t i = 0; i < v.size(); i++) d[i] = (double)v[i];
    return d;
}

std::vector<double> to_double_T(const LVec &v, int N) {
    std::vector<double> d(v.size());
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) d[(size_t)j * N + i] = (double)v[(size_t)i * N + j];
    return d;
}

// Velocity weights weightQv (MPCPlanner.cpp:535-547): constant Q, or makeGaussian(N, rw, rw) * Q * N
// (:1161-1171, 1-based knots, integer mean and deviation).
LVec velocity_weights(const mpcb_params &p) {
    const int N = p.horizon;
    LVec w(N);
    if (p.replanning_window == 1000) {
        for (int i = 0; i < N; i++) w[i] = p.weight_Q;
    } else {
        const ld mean = p.replanning_window, sd = p.replanning_window;
        for (int i = 0; i < N; i++) {
            const ld q = ((ld)(i + 1) - mean) / sd;
            w[i] = expl(-0.5L * q * q) / (sqrtl(2 * (ld)M_PIl) * sd) * (ld)p.weight_Q * (ld)N;
        }
    }
    return w;
}

} // namespace

int build_layout(const mpcb_params &p, HostSetup &o, const char **err) {
    const int N = p.horizon;
    if (N < 1 || N > 512) {
        *err = "horizon must be in [1, 512]";
        return MPCB_E_ARG;
    }
    if (!(p.Ts > 0) || !(p.gravity > 0) || !(p.weight_R > 0)) {
        *err = "Ts, gravity and weight_R must be positive";
        return MPCB_E_ARG;
    }
    const int form = p.formulation;
    if (form < MPCB_BOX_1D || form > MPCB_BOX_SLACK) {
        *err = "unknown formulation";
        return MPCB_E_ARG;
    }
    if (form == MPCB_UNCONSTRAINED && !(p.weight_Q > 0)) {
        *err = "weight_Q must be positive for UNCONSTRAINED";
        return MPCB_E_ARG;
    }
    const bool coupled = form == MPCB_COUPLED_JERK || form == MPCB_COUPLED_VEL || form == MPCB_COUPLED_REF;
    o = HostSetup();
    o.N = N;
    o.axes = coupled ? 2 : 1;
    o.nvar = form == MPCB_BOX_SLACK ? 3 * N : o.axes * N;
    if (form == MPCB_BOX_SLACK && !(p.weight_Qs > 0)) {
        *err = "weight_Qs must be positive for BOX_SLACK";
        return MPCB_E_ARG;
    }
    if (form == MPCB_BOX_SLACK) {
        o.slack_ls = sqrt(p.weight_Qs);
        o.slack_g0 = 1000 * p.weight_Qs; // MPCPlanner.cpp:373
    }
    // Row bookkeeping.
    o.knot_phase.assign(N, 0);
    o.knot_row0.assign(N, 0);
    if (coupled) {
        const int steps = p.number_of_steps;
        if (steps < 0 || steps > 255) {
            *err = "number_of_steps must be in [0, 255]";
            return MPCB_E_ARG;
        }
        // computeSteps (MPCPlanner.cpp:1200,1209): step_knots = floor(N/steps),
        // phase_duration = floor(step_knots/2); per step a 4-stance then a 3-stance
        // phase; the knots left over form a final 4-stance phase (:1309-1321).
        // steps == 0: no footstep schedule -- every knot belongs to the trailing phase and
        // owns four rows (4k .. 4k+3), which the caller fills (rows_pqr).
        const int step_knots = steps > 0 ? N / steps : 0, pd = step_knots / 2;
        const int covered = 2 * pd * steps;
        o.steps = steps;
        o.phase_knots = pd;
        o.nphase = 2 * steps + 1;
        o.m = 7 * pd * steps + 4 * (N - covered);
        if (steps > 0 && o.m > 4 * step_knots * steps) {
            *err = "horizon/number_of_steps leaves more trailing knots than computeSteps has rows for";
            return MPCB_E_ARG;
        }
        for (int k = 0; k < N; k++) {
            if (k < covered) {
                const int i = k / (2 * pd), off = k % (2 * pd);
                o.knot_phase[k] = 2 * i + (off >= pd ? 1 : 0);
                o.knot_row0[k] = i * 7 * pd + (off < pd ? 4 * off : 4 * pd + 3 * (off - pd));
            } else {
                o.knot_phase[k] = 2 * steps;
                o.knot_row0[k] = 7 * pd * steps + 4 * (k - covered);
            }
        }
        o.row_knot.assign(o.m, 0);
        o.row_slot.assign(o.m, 0);
        for (int k = 0; k < N; k++) {
            const int E = (o.knot_phase[k] & 1) ? 3 : 4;
            for (int e = 0; e < E; e++) {
                o.row_knot[o.knot_row0[k] + e] = k;
                o.row_slot[o.knot_row0[k] + e] = e;
            }
        }
    } else {
        o.m = (form == MPCB_BOX_1D) ? 2 * N : (form == MPCB_BOX_SLACK) ? 4 * N : 0;
        o.nphase = 0;
        for (int k = 0; k < N; k++) o.knot_row0[k] = k;
    }
    return 0;
}

bool height_tables_apply(int form) {
    return form == MPCB_BOX_1D || form == MPCB_COUPLED_JERK || form == MPCB_COUPLED_VEL;
}

void finish_scalars(const mpcb_params &p, HostSetup &o, double traceG1, double traceJ1) {
    // traces of the full Hessian and of L^-T (eiquadprog.hpp:162,176): `axes` identical blocks
    const ld tr = traceG1, trJ = traceJ1;
    const int N = o.N;
    o.traceG1 = traceG1;
    o.c1 = (double)(tr * o.axes);
    o.c2 = (double)(trJ * o.axes);
    if (p.formulation == MPCB_BOX_SLACK) {
        // G = diag(R I_N, Qs I_2N) (MPCPlanner.cpp:368-370)
        o.c1 = (double)(tr + 2 * (ld)N * (ld)p.weight_Qs);
        o.c2 = (double)(trJ + 2 * (ld)N / sqrtl((ld)p.weight_Qs));
    }
}

void table_spec(const mpcb_params &p, const HostSetup &o, TableSpec &t) {
    const int N = p.horizon, form = p.formulation;
    t = TableSpec();
    t.N = N;
    t.form = form;
    t.nphase = form == MPCB_COUPLED_REF ? o.nphase : 0;
    t.height_tables = height_tables_apply(form);
    t.Ts = p.Ts;
    t.Qs = p.weight_Qs;
    auto split = [](ld v, double &hi, double &lo) {
        hi = (double)v;
        lo = (double)(v - (ld)hi);
    };
    split(form == MPCB_UNCONSTRAINED ? (ld)p.weight_R / (ld)p.weight_Q : (ld)p.weight_R, t.diag_hi, t.diag_lo);
    split((ld)p.height / (ld)p.gravity, t.c_hi, t.c_lo);
    if (form == MPCB_COUPLED_VEL || form == MPCB_COUPLED_REF) {
        const LVec w = velocity_weights(p);
        t.w_hi.resize(N);
        t.w_lo.resize(N);
        for (int i = 0; i < N; i++) split(w[i], t.w_hi[i], t.w_lo[i]);
    }
    if (form == MPCB_COUPLED_REF) t.knot_phase = o.knot_phase;
}

int build_tables_host(const mpcb_params &p, HostSetup &o, const char **err) {
    const int N = p.horizon, form = p.formulation;

    const ld Ts = p.Ts;
    const ld Cz[3] = {1, 0, -(ld)p.height / (ld)p.gravity};
    const ld Cx[3] = {1, 0, 0}, Cv[3] = {0, 1, 0};
    Pred Z = prediction(Cz, Ts, N), Xp = prediction(Cx, Ts, N), Xv = prediction(Cv, Ts, N);

    // dense Zu^T (N x N row-major): row i, col k = Zu[k][i] = zcol[k-i] for k >= i
    LVec ZuT((size_t)N * N, 0);
    for (int i = 0; i < N; i++)
        for (int k = i; k < N; k++) ZuT[(size_t)i * N + k] = Z.col[k - i];

    // Hessian block G1 and the linear map Gm: g0 = Gm * (pos, vel, acc, target speed)
    LVec G1((size_t)N * N, 0), Gm((size_t)N * 4, 0);
    for (int i = 0; i < N; i++) G1[(size_t)i * N + i] = p.weight_R;
    if (form == MPCB_UNCONSTRAINED) {
        // G = Zu^T Zu + (R/Q) I (MPCPlanner.cpp:233)
        for (int i = 0; i < N; i++) G1[(size_t)i * N + i] = (ld)p.weight_R / (ld)p.weight_Q;
        LVec Zu((size_t)N * N, 0);
        for (int i = 0; i < N; i++)
            for (int k = 0; k <= i; k++) Zu[(size_t)i * N + k] = Z.col[i - k];
        LVec ZtZ = toepT_mul(Z.col, Zu, N, N);
        for (size_t i = 0; i < G1.size(); i++) G1[i] += ZtZ[i];
        // b = Zu^T (Zx x0 - zref) (:234): the x0 part is linear in x0
        LVec Zx4((size_t)N * 4, 0);
        for (int i = 0; i < N; i++)
            for (int j = 0; j < 3; j++) Zx4[(size_t)i * 4 + j] = Z.sx[(size_t)i * 3 + j];
        Gm = toepT_mul(Z.col, Zx4, N, 4);
    }
    if (form == MPCB_COUPLED_VEL || form == MPCB_COUPLED_REF) {
        const LVec w = velocity_weights(p);
        // Xvu^T W Xvu, Xvu^T W Xvx, -Xvu^T W 1
        LVec WX((size_t)N * N, 0), WXx((size_t)N * 4, 0);
        for (int i = 0; i < N; i++) {
            for (int k = 0; k <= i; k++) WX[(size_t)i * N + k] = w[i] * Xv.col[i - k];
            for (int j = 0; j < 3; j++) WXx[(size_t)i * 4 + j] = w[i] * Xv.sx[(size_t)i * 3 + j];
            WXx[(size_t)i * 4 + 3] = -w[i];
        }
        LVec Gv = toepT_mul(Xv.col, WX, N, N);
        for (size_t i = 0; i < G1.size(); i++) G1[i] += Gv[i];
        Gm = toepT_mul(Xv.col, WXx, N, 4);
    }
    if (form == MPCB_COUPLED_REF) {
        // Gref = Qs Zu^T Zu ; gref = Qs Zu^T (Zx x0 - zref) (MPCPlanner.cpp:728-735)
        LVec Zu((size_t)N * N, 0), Zx4((size_t)N * 4, 0);
        for (int i = 0; i < N; i++) {
            for (int k = 0; k <= i; k++) Zu[(size_t)i * N + k] = Z.col[i - k];
            for (int j = 0; j < 3; j++) Zx4[(size_t)i * 4 + j] = Z.sx[(size_t)i * 3 + j];
        }
        LVec ZtZ = toepT_mul(Z.col, Zu, N, N), ZtZx = toepT_mul(Z.col, Zx4, N, 4);
        for (size_t i = 0; i < G1.size(); i++) G1[i] += (ld)p.weight_Qs * ZtZ[i];
        for (size_t i = 0; i < Gm.size(); i++) Gm[i] += (ld)p.weight_Qs * ZtZx[i];
    }
    ld tr = 0;
    for (int i = 0; i < N; i++) tr += G1[(size_t)i * N + i];

    LVec L = G1;
    if (!cholesky(L, N)) {
        *err = "condensed Hessian is not positive definite";
        return MPCB_E_NUMERIC;
    }
    ld trJ = 0;
    for (int i = 0; i < N; i++) trJ += 1 / L[(size_t)i * N + i];
    finish_scalars(p, o, (double)tr, (double)trJ);

    // W = G1^-1 Zu^T  (N x N), then the response matrices
    LVec W = ZuT;
    chol_solve(L, W, N, N);
    LVec S = toep_mul(Z.col, W, N, N), Sc = toep_mul(Xp.col, W, N, N), Sv = toep_mul(Xv.col, W, N, N);
    // S is symmetric in exact arithmetic: store the symmetrised double rounding
    for (int i = 0; i < N; i++)
        for (int j = i + 1; j < N; j++) {
            const ld a = (S[(size_t)i * N + j] + S[(size_t)j * N + i]) / 2;
            S[(size_t)i * N + j] = S[(size_t)j * N + i] = a;
        }
    o.S = to_double(S);
    o.Wt = to_double_T(W, N);
    o.Sc = to_double_T(Sc, N);
    o.Sv = to_double_T(Sv, N);

    // unconstrained response: jerk = Kx w, zmp = Kz w, com = Kc w, vel = Kv w
    LVec Kx = Gm;
    chol_solve(L, Kx, N, 4);
    for (ld &v : Kx) v = -v;
    LVec Kz = toep_mul(Z.col, Kx, N, 4), Kc = toep_mul(Xp.col, Kx, N, 4), Kv = toep_mul(Xv.col, Kx, N, 4);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < 3; j++) {
            Kz[(size_t)i * 4 + j] += Z.sx[(size_t)i * 3 + j];
            Kc[(size_t)i * 4 + j] += Xp.sx[(size_t)i * 3 + j];
            Kv[(size_t)i * 4 + j] += Xv.sx[(size_t)i * 3 + j];
        }
    o.Kx = to_double(Kx);
    o.Kz = to_double(Kz);
    o.Kc = to_double(Kc);
    o.Kv = to_double(Kv);
    // f_unc = 1/2 g0 . x_unc = w^T (1/2 Gm^T Kx) w
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
            ld s = 0;
            for (int i = 0; i < N; i++) s += Gm[(size_t)i * 4 + a] * Kx[(size_t)i * 4 + b];
            o.F[a * 4 + b] = (double)(s / 2);
        }

    o.zu_col = to_double(Z.col);
    o.zx = to_double(Z.sx);
    o.xp_col = to_double(Xp.col);
    o.xpx = to_double(Xp.sx);
    o.xv_col = to_double(Xv.col);
    o.xvx = to_double(Xv.sx);
    {
        // J0 = L^-T: solve L^T J0 = I column by column (upper triangular result)
        LVec J0((size_t)N * N, 0);
        for (int c = 0; c < N; c++) {
            J0[(size_t)c * N + c] = 1 / L[(size_t)c * N + c];
            for (int i = c - 1; i >= 0; i--) {
                ld acc = 0;
                for (int k = i + 1; k <= c; k++) acc += L[(size_t)k * N + i] * J0[(size_t)k * N + c];
                J0[(size_t)i * N + c] = -acc / L[(size_t)i * N + i];
            }
        }
        o.J0 = to_double_T(J0, N); // column-major
    }

    if (form == MPCB_COUPLED_REF) {
        // The reference of computeCentroid is constant over a phase, so its effect is
        // a sum over phases of (centroid of the phase) x (response summed over the
        // phase's knots).  g0 gains -Qs Zu' zref (MPCPlanner.cpp:731-735).
        const int P = o.nphase;
        const ld Qs = p.weight_Qs;
        LVec Sp((size_t)P * N, 0), Wp((size_t)P * N, 0);
        LVec F2((size_t)4 * P, 0), F3((size_t)P * P, 0);
        for (int k = 0; k < N; k++) {
            const int ph = o.knot_phase[k];
            for (int i = 0; i < N; i++) {
                Sp[(size_t)ph * N + i] += Qs * S[(size_t)k * N + i];
                Wp[(size_t)ph * N + i] += Qs * W[(size_t)i * N + k];
            }
            for (int a = 0; a < 4; a++) {
                const ld zukx = Kz[(size_t)k * 4 + a] - (a < 3 ? Z.sx[(size_t)k * 3 + a] : 0);
                F2[(size_t)a * P + ph] -= Qs * zukx;
            }
            for (int k2 = 0; k2 < N; k2++)
                F3[(size_t)ph * P + o.knot_phase[k2]] -= Qs * Qs * S[(size_t)k * N + k2] / 2;
        }
        o.Sp = to_double(Sp);
        o.Wp = to_double(Wp);
        o.F2 = to_double(F2);
        o.F3 = to_double(F3);
    }
    return 0;
}

// Tables for per-instance CoM heights.  With c = height / gravity the ZMP matrices are
// Zu = Xpu - c Xau and Zx = Xpx - c Xax (MPCPlanner.cpp:58-63: Cz = Cx - c Ca), and for the
// formulations whose Hessian does not contain Zu (box, coupled jerk-only, coupled velocity)
// G1^-1 =: H is the same for every height, so
//     S(c)   = Zu H Zu'  = Spp - c Sps + c^2 Saa
//     W(c)^T = Zu H      = Wtp - c Wta
//     Kz(c)  = Zu Kx + Zx = Kzp - c Kza
// with the seven height-free tables built here (long double, like the others).
int build_height_tables(const mpcb_params &p, HostSetup &o, const char **err) {
    const int N = p.horizon, form = p.formulation;
    if (form != MPCB_BOX_1D && form != MPCB_COUPLED_JERK && form != MPCB_COUPLED_VEL) {
        *err = "per-instance heights need a formulation whose Hessian does not depend on the height";
        return MPCB_E_ARG;
    }
    const ld Ts = p.Ts;
    const ld Cx[3] = {1, 0, 0}, Cv[3] = {0, 1, 0}, Ca[3] = {0, 0, 1};
    Pred Xp = prediction(Cx, Ts, N), Xv = prediction(Cv, Ts, N), Xa = prediction(Ca, Ts, N);
    LVec G1((size_t)N * N, 0), Gm((size_t)N * 4, 0);
    for (int i = 0; i < N; i++) G1[(size_t)i * N + i] = p.weight_R;
    if (form == MPCB_COUPLED_VEL) {
        const LVec w = velocity_weights(p);
        LVec WX((size_t)N * N, 0), WXx((size_t)N * 4, 0);
        for (int i = 0; i < N; i++) {
            for (int k = 0; k <= i; k++) WX[(size_t)i * N + k] = w[i] * Xv.col[i - k];
            for (int j = 0; j < 3; j++) WXx[(size_t)i * 4 + j] = w[i] * Xv.sx[(size_t)i * 3 + j];
            WXx[(size_t)i * 4 + 3] = -w[i];
        }
        LVec Gv = toepT_mul(Xv.col, WX, N, N);
        for (size_t i = 0; i < G1.size(); i++) G1[i] += Gv[i];
        Gm = toepT_mul(Xv.col, WXx, N, 4);
    }
    LVec L = G1;
    if (!cholesky(L, N)) {
        *err = "condensed Hessian is not positive definite";
        return MPCB_E_NUMERIC;
    }
    // H Xp', H Xa'  (N x N): column k = H * (row k of the Toeplitz matrix)
    LVec XpT((size_t)N * N, 0), XaT((size_t)N * N, 0);
    for (int i = 0; i < N; i++)
        for (int k = i; k < N; k++) {
            XpT[(size_t)i * N + k] = Xp.col[k - i];
            XaT[(size_t)i * N + k] = Xa.col[k - i];
        }
    LVec Wp = XpT, Wa = XaT;
    chol_solve(L, Wp, N, N);
    chol_solve(L, Wa, N, N);
    LVec Spp = toep_mul(Xp.col, Wp, N, N), Spa = toep_mul(Xp.col, Wa, N, N);
    LVec Sap = toep_mul(Xa.col, Wp, N, N), Saa = toep_mul(Xa.col, Wa, N, N);
    LVec Sps((size_t)N * N);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            // symmetrised, like S
            Sps[(size_t)i * N + j] = ((Spa[(size_t)i * N + j] + Sap[(size_t)i * N + j]) +
                                      (Spa[(size_t)j * N + i] + Sap[(size_t)j * N + i])) / 2;
        }
    for (int i = 0; i < N; i++)
        for (int j = i + 1; j < N; j++) {
            const ld a = (Spp[(size_t)i * N + j] + Spp[(size_t)j * N + i]) / 2;
            Spp[(size_t)i * N + j] = Spp[(size_t)j * N + i] = a;
            const ld b = (Saa[(size_t)i * N + j] + Saa[(size_t)j * N + i]) / 2;
            Saa[(size_t)i * N + j] = Saa[(size_t)j * N + i] = b;
        }
    o.Spp = to_double(Spp);
    o.Sps = to_double(Sps);
    o.Saa = to_double(Saa);
    o.Wtp = to_double_T(Wp, N);
    o.Wta = to_double_T(Wa, N);
    LVec Kx = Gm;
    chol_solve(L, Kx, N, 4);
    for (ld &v : Kx) v = -v;
    LVec Kzp = toep_mul(Xp.col, Kx, N, 4), Kza = toep_mul(Xa.col, Kx, N, 4);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < 3; j++) {
            Kzp[(size_t)i * 4 + j] += Xp.sx[(size_t)i * 3 + j];
            Kza[(size_t)i * 4 + j] += Xa.sx[(size_t)i * 3 + j];
        }
    o.Kzp = to_double(Kzp);
    o.Kza = to_double(Kza);
    return 0;
}

} // namespace mpcb

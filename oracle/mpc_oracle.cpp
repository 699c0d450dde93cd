/*
 * mpc_oracle.cpp -- CPU oracle: scalar, Eigen-free restatement of the
 * reference planner.  TEST INFRASTRUCTURE ONLY (see mpc_oracle.h).
 *
 * Every function names the reference lines it follows.  Storage is
 * column-major like Eigen's; summation over the inner index runs in ascending
 * order (Eigen's blocked/vectorised kernels use another order, so agreement
 * with a real Eigen build is to rounding, not bitwise).
 *
 * Mode `dense` decides whether products run through the structural zeros of
 * their operands (what the reference's dense Eigen products do; this is the
 * honest CPU cost) or skip exact-zero multiplicands (identical values, since
 * each skipped term is an exact zero added in the same ascending order).
 */
#include "mpc_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

typedef std::vector<double> Vec;

struct Mat {
    int r = 0, c = 0;
    Vec a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return a[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return a[(size_t)j * r + i]; }
    double *col(int j) { return &a[(size_t)j * r]; }
    const double *col(int j) const { return &a[(size_t)j * r]; }
};

Mat transpose(const Mat &A) {
    Mat T(A.c, A.r);
    for (int j = 0; j < A.c; j++)
        for (int i = 0; i < A.r; i++) T(j, i) = A(i, j);
    return T;
}

/* C = A*B.  dense: plain triple loop.  !dense: skip exact zeros of both
 * operands; per output element the surviving terms are added in the same
 * ascending-k order, so the values are identical. */
Mat matmul(const Mat &A, const Mat &B, bool dense) {
    Mat C(A.r, B.c);
    if (dense) {
        for (int j = 0; j < B.c; j++) {
            double *cj = C.col(j);
            for (int k = 0; k < A.c; k++) {
                const double bkj = B(k, j);
                const double *ak = A.col(k);
                for (int i = 0; i < A.r; i++) cj[i] += ak[i] * bkj;
            }
        }
        return C;
    }
    std::vector<int> start(A.c + 1, 0), idx;
    Vec val;
    for (int k = 0; k < A.c; k++) {
        const double *ak = A.col(k);
        for (int i = 0; i < A.r; i++)
            if (ak[i] != 0.0) {
                idx.push_back(i);
                val.push_back(ak[i]);
            }
        start[k + 1] = (int)idx.size();
    }
    for (int j = 0; j < B.c; j++) {
        double *cj = C.col(j);
        for (int k = 0; k < A.c; k++) {
            const double bkj = B(k, j);
            if (bkj == 0.0) continue;
            for (int t = start[k]; t < start[k + 1]; t++) cj[idx[t]] += val[t] * bkj;
        }
    }
    return C;
}

Vec matvec(const Mat &A, const Vec &x) {
    Vec y(A.r, 0.0);
    for (int k = 0; k < A.c; k++) {
        const double xk = x[k];
        const double *ak = A.col(k);
        for (int i = 0; i < A.r; i++) y[i] += ak[i] * xk;
    }
    return y;
}

/* y = x^T A (row vector times matrix) */
Vec vecmat(const Vec &x, const Mat &A) {
    Vec y(A.c, 0.0);
    for (int j = 0; j < A.c; j++) {
        const double *aj = A.col(j);
        double s = 0.0;
        for (int i = 0; i < A.r; i++) s += x[i] * aj[i];
        y[j] = s;
    }
    return y;
}

double dot(const double *a, const double *b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

Mat diag(const Vec &d) {
    Mat D((int)d.size(), (int)d.size());
    for (int i = 0; i < (int)d.size(); i++) D(i, i) = d[i];
    return D;
}

const double kInf = std::numeric_limits<double>::infinity();
const double kEps = std::numeric_limits<double>::epsilon();

/* ------------------------------------------------------------------ *
 * eiquadprog.hpp
 * ------------------------------------------------------------------ */

/* eiquadprog.hpp:90-108 */
double distance(double a, double b) {
    double a1 = std::fabs(a), b1 = std::fabs(b), t;
    if (a1 > b1) {
        t = b1 / a1;
        return a1 * std::sqrt(1.0 + t * t);
    } else if (b1 > a1) {
        t = a1 / b1;
        return b1 * std::sqrt(1.0 + t * t);
    }
    return a1 * std::sqrt(2.0);
}

struct QPWork {
    int n, m;
    Mat R, J;
    Vec s, z, r, d, np, u, x_old, u_old;
    std::vector<int> A, A_old, iai;
    std::vector<char> iaexcl;
    double R_norm;
    int iq;
    /* statistics */
    int n_add = 0, n_drop = 0, n_l2a = 0;
    double falg = 0.0;
};

/* Optional record of the decisions of one solve (diagnostics for the tests; the reference keeps
 * nothing of the kind): one row per step-direction pass (l2a) with the quantities every branch
 * of that pass compares.  Set by orc_solve_box_trace around a solve of the calling thread. */
struct QPTrace {
    int max_rows = 0, rows = 0;
    double *rec = nullptr; /* [max_rows][ORC_TRACE_COLS] */
};
thread_local QPTrace *g_trace = nullptr;

/* eiquadprog.hpp:450-513 */
bool add_constraint(QPWork &w) {
    const int n = w.n;
    Mat &J = w.J;
    Vec &d = w.d;
    int &iq = w.iq;
    w.falg += 6.0 * n * (double)(n - iq - 1);
    w.n_add++;
    for (int j = n - 1; j >= iq + 1; j--) {
        double cc = d[j - 1], ss = d[j];
        double h = distance(cc, ss);
        if (h == 0.0) continue;
        d[j] = 0.0;
        ss = ss / h;
        cc = cc / h;
        if (cc < 0.0) {
            cc = -cc;
            ss = -ss;
            d[j - 1] = -h;
        } else
            d[j - 1] = h;
        double xny = ss / (1.0 + cc);
        double *j0 = J.col(j - 1), *j1 = J.col(j);
        for (int k = 0; k < n; k++) {
            double t1 = j0[k], t2 = j1[k];
            j0[k] = t1 * cc + t2 * ss;
            j1[k] = xny * (t1 + j0[k]) - t2;
        }
    }
    iq++;
    for (int i = 0; i < iq; i++) w.R(i, iq - 1) = d[i];
    if (std::fabs(d[iq - 1]) <= kEps * w.R_norm) return false;
    w.R_norm = std::max(w.R_norm, std::fabs(d[iq - 1]));
    return true;
}

/* eiquadprog.hpp:516-592 (p = 0) */
void delete_constraint(QPWork &w, int l) {
    const int n = w.n;
    Mat &R = w.R, &J = w.J;
    int &iq = w.iq;
    int qq = -1;
    for (int i = 0; i < iq; i++)
        if (w.A[i] == l) {
            qq = i;
            break;
        }
    if (qq < 0) qq = 0; /* the reference leaves qq uninitialised here */
    {
        double c = (double)(iq - 1 - qq);
        w.falg += 6.0 * n * c + 3.0 * c * (c - 1.0);
        w.n_drop++;
    }
    for (int i = qq; i < iq - 1; i++) {
        w.A[i] = w.A[i + 1];
        w.u[i] = w.u[i + 1];
        for (int k = 0; k < n; k++) R(k, i) = R(k, i + 1);
    }
    w.A[iq - 1] = w.A[iq];
    w.u[iq - 1] = w.u[iq];
    w.A[iq] = 0;
    w.u[iq] = 0.0;
    for (int j = 0; j < iq; j++) R(j, iq - 1) = 0.0;
    iq--;
    if (iq == 0) return;
    for (int j = qq; j < iq; j++) {
        double cc = R(j, j), ss = R(j + 1, j);
        double h = distance(cc, ss);
        if (h == 0.0) continue;
        cc = cc / h;
        ss = ss / h;
        R(j + 1, j) = 0.0;
        if (cc < 0.0) {
            R(j, j) = -h;
            cc = -cc;
            ss = -ss;
        } else
            R(j, j) = h;
        double xny = ss / (1.0 + cc);
        for (int k = j + 1; k < iq; k++) {
            double t1 = R(j, k), t2 = R(j + 1, k);
            R(j, k) = t1 * cc + t2 * ss;
            R(j + 1, k) = xny * (t1 + R(j, k)) - t2;
        }
        double *j0 = J.col(j), *j1 = J.col(j + 1);
        for (int k = 0; k < n; k++) {
            double t1 = j0[k], t2 = j1[k];
            j0[k] = t1 * cc + t2 * ss;
            j1[k] = xny * (j0[k] + t1) - t2;
        }
    }
}

/* In-place lower Cholesky G = L L^T (Eigen::LLT<MatrixXd,Lower>,
 * eiquadprog.hpp:140,165).  Returns false if a pivot is not positive. */
bool cholesky_lower(Mat &L) {
    const int n = L.r;
    for (int j = 0; j < n; j++) {
        double d = L(j, j);
        for (int k = 0; k < j; k++) d -= L(j, k) * L(j, k);
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        L(j, j) = d;
        /* column-oriented update of the rest of column j */
        for (int i = j + 1; i < n; i++) {
            double s = L(i, j);
            for (int k = 0; k < j; k++) s -= L(i, k) * L(j, k);
            L(i, j) = s / d;
        }
    }
    for (int j = 1; j < n; j++)
        for (int i = 0; i < j; i++) L(i, j) = 0.0;
    return true;
}

/* eiquadprog.hpp:130-447, p = 0 (no call site passes equalities).
 * CI is n x m, one column per constraint. */
double solve_quadprog(const Mat &G, const Vec &g0, const Mat &CI, const Vec &ci0,
                      Vec &x, orc_info *info, int *active, int iter_cap) {
    const int n = (int)g0.size(), m = (int)ci0.size();
    QPWork w;
    w.n = n;
    w.m = m;
    w.R = Mat(n, n);
    w.J = Mat(n, n);
    w.s.assign(m, 0.0);
    w.z.assign(n, 0.0);
    w.r.assign(m + 1, 0.0);
    w.d.assign(n, 0.0);
    w.np.assign(n, 0.0);
    w.u.assign(m + 1, 0.0);
    w.x_old.assign(n, 0.0);
    w.u_old.assign(m + 1, 0.0);
    w.A.assign(m + 1, 0);
    w.A_old.assign(m + 1, 0);
    w.iai.assign(m + 1, 0);
    w.iaexcl.assign(m + 1, 1);
    Mat &R = w.R, &J = w.J;
    Vec &s = w.s, &z = w.z, &r = w.r, &d = w.d, &np = w.np, &u = w.u;
    int &iq = w.iq;
    double f_value, psi, c1, c2, sum, ss, t, t1, t2;
    int ip, l, iter = 0, status = ORC_OK;

    x.assign(n, 0.0);

    /* :162 */
    c1 = 0.0;
    for (int i = 0; i < n; i++) c1 += G(i, i);
    /* :165 */
    Mat L = G;
    cholesky_lower(L);
    /* :168-170 */
    w.R_norm = 1.0;
    /* :174-176  J = U^{-1} with U = L^T: column j of J solves U y = e_j. */
    for (int j = 0; j < n; j++) {
        double *y = J.col(j);
        y[j] = 1.0 / L(j, j);
        for (int i = j - 1; i >= 0; i--) {
            double acc = 0.0;
            for (int k = i + 1; k <= j; k++) acc += L(k, i) * y[k];
            y[i] = -acc / L(i, i);
        }
    }
    c2 = 0.0;
    for (int i = 0; i < n; i++) c2 += J(i, i);
    /* :188-191  x = -G^{-1} g0 through the factor */
    {
        Vec y(n);
        for (int i = 0; i < n; i++) {
            double acc = g0[i];
            for (int k = 0; k < i; k++) acc -= L(i, k) * y[k];
            y[i] = acc / L(i, i);
        }
        for (int i = n - 1; i >= 0; i--) {
            double acc = y[i];
            for (int k = i + 1; k < n; k++) acc -= L(k, i) * x[k];
            x[i] = acc / L(i, i);
        }
        for (int i = 0; i < n; i++) x[i] = -x[i];
    }
    f_value = 0.5 * dot(g0.data(), x.data(), n);
    w.falg += 2.0 * n * (double)n;

    iq = 0;
    for (int i = 0; i < m; i++) w.iai[i] = i; /* :237-238 */

l1:
    iter++;
    if (iter_cap > 0 && iter > iter_cap) {
        status = ORC_ITER_CAP;
        goto done;
    }
    for (int i = 0; i < iq; i++) w.iai[w.A[i]] = -1; /* :245-249 */
    ss = 0.0;
    psi = 0.0;
    ip = 0;
    for (int i = 0; i < m; i++) { /* :255-261 */
        w.iaexcl[i] = 1;
        sum = dot(CI.col(i), x.data(), n) + ci0[i];
        s[i] = sum;
        psi += std::min(0.0, sum);
    }
    w.falg += 2.0 * m * (double)n;
    if (std::fabs(psi) <= m * kEps * c1 * c2 * 100.0) goto done; /* :267-272 */
    for (int i = 0; i < iq; i++) { /* :275-277 */
        w.u_old[i] = u[i];
        w.A_old[i] = w.A[i];
    }
    w.x_old = x;

l2: /* :279-292 */
    for (int i = 0; i < m; i++)
        if (s[i] < ss && w.iai[i] != -1 && w.iaexcl[i]) {
            ss = s[i];
            ip = i;
        }
    if (ss >= 0.0) goto done;
    std::memcpy(np.data(), CI.col(ip), sizeof(double) * n); /* :295 */
    u[iq] = 0.0;
    w.A[iq] = ip;

l2a: /* :306-311  d = J^T np ; z = J2 d2 ; r = R^{-1} d1 */
    w.n_l2a++;
    if (iter_cap > 0 && w.n_l2a > iter_cap) {
        status = ORC_ITER_CAP;
        goto done;
    }
    w.falg += 2.0 * n * (double)n + 2.0 * n * (double)(n - iq) + (double)iq * iq;
    for (int j = 0; j < n; j++) d[j] = dot(J.col(j), np.data(), n);
    std::fill(z.begin(), z.end(), 0.0);
    for (int j = iq; j < n; j++) {
        const double dj = d[j];
        const double *jj = J.col(j);
        for (int k = 0; k < n; k++) z[k] += jj[k] * dj;
    }
    for (int i = iq - 1; i >= 0; i--) {
        double acc = d[i];
        for (int k = i + 1; k < iq; k++) acc -= R(i, k) * r[k];
        r[i] = acc / R(i, i);
    }
    /* :322-334 */
    l = 0;
    t1 = kInf;
    for (int k = 0; k < iq; k++) {
        double tmp;
        if (r[k] > 0.0 && ((tmp = u[k] / r[k]) < t1)) {
            t1 = tmp;
            l = w.A[k];
        }
    }
    /* :336-339 */
    if (std::fabs(dot(z.data(), z.data(), n)) > kEps)
        t2 = -s[ip] / dot(z.data(), np.data(), n);
    else
        t2 = kInf;
    t = std::min(t1, t2);
    if (g_trace && g_trace->rows < g_trace->max_rows) {
        double *rec = g_trace->rec + (size_t)g_trace->rows++ * ORC_TRACE_COLS;
        /* runner-up of step 2's choice among the rows it may choose from, of the ratio test, and
         * how close r comes to changing sign */
        double s2 = kInf, t1b = kInf, rmax = 0.0, rmin_abs = kInf;
        for (int i = 0; i < m; i++)
            if (i != ip && w.iai[i] != -1 && w.iaexcl[i] && s[i] < s2) s2 = s[i];
        for (int k = 0; k < iq; k++) {
            rmax = std::max(rmax, std::fabs(r[k]));
            rmin_abs = std::min(rmin_abs, std::fabs(r[k]));
            if (r[k] > 0.0 && w.A[k] != l) t1b = std::min(t1b, u[k] / r[k]);
        }
        rec[0] = iter;
        rec[1] = w.n_l2a;
        rec[2] = iq;
        rec[3] = ip;
        rec[4] = s[ip];
        rec[5] = s2;
        rec[6] = t1;
        rec[7] = t1b;
        rec[8] = t2;
        rec[9] = dot(z.data(), z.data(), n);
        rec[10] = dot(z.data(), np.data(), n);
        rec[11] = iq > 0 ? rmin_abs / std::max(rmax, 1e-300) : kInf;
        rec[12] = l;
        {
            /* what add_constraint compares with eps * R_norm (:494) is |d(iq)| after its rotations,
             * i.e. the norm of d2 */
            double d2 = 0.0;
            for (int j = iq; j < n; j++) d2 += d[j] * d[j];
            rec[13] = std::sqrt(d2) / (kEps * w.R_norm);
        }
    }
    if (t >= kInf) { /* :350-356 */
        status = ORC_INFEASIBLE;
        f_value = kInf;
        goto done;
    }
    if (t2 >= kInf) { /* :358-373 */
        for (int k = 0; k < iq; k++) u[k] -= t * r[k];
        u[iq] += t;
        w.iai[l] = l;
        delete_constraint(w, l);
        goto l2a;
    }
    /* :377-382 */
    for (int k = 0; k < n; k++) x[k] += t * z[k];
    f_value += t * dot(z.data(), np.data(), n) * (0.5 * t + u[iq]);
    for (int k = 0; k < iq; k++) u[k] -= t * r[k];
    u[iq] += t;
    if (t == t2) { /* :392-426 */
        if (!add_constraint(w)) {
            w.iaexcl[ip] = 0;
            delete_constraint(w, ip);
            for (int i = 0; i < m; i++) w.iai[i] = i;
            for (int i = 0; i < iq; i++) {
                w.A[i] = w.A_old[i];
                w.iai[w.A[i]] = -1;
                u[i] = w.u_old[i];
            }
            x = w.x_old;
            goto l2;
        } else
            w.iai[ip] = -1;
        goto l1;
    }
    /* :434-446 */
    w.iai[l] = l;
    delete_constraint(w, l);
    s[ip] = dot(CI.col(ip), x.data(), n) + ci0[ip];
    goto l2a;

done:
    if (info) {
        info->f = f_value;
        info->status = status;
        info->iters = iter;
        info->n_active = iq;
        info->n_add = w.n_add;
        info->n_drop = w.n_drop;
        info->n_l2a = w.n_l2a;
        info->falg = w.falg;
    }
    if (active)
        for (int i = 0; i < iq; i++) active[i] = w.A[i];
    return f_value;
}

/* ------------------------------------------------------------------ *
 * planning.h
 * ------------------------------------------------------------------ */
struct P2 {
    double x, y;
};

/* planning.h:143-149 */
double point2_is_right_of_line(P2 p0, P2 p1, P2 p2) {
    return (p2.x - p0.x) * (p1.y - p0.y) - (p1.x - p0.x) * (p2.y - p0.y);
}

/* planning.h:182-197.  The pivot is p[0], whatever it is. */
void ccw_sort(std::vector<P2> &p) {
    const int n = (int)p.size();
    for (int i = 1; i < n - 1; i++)
        for (int j = i + 1; j < n; j++)
            if (point2_is_right_of_line(p[0], p[i], p[j]) > 0.0) std::swap(p[i], p[j]);
}

struct Line {
    double p, q, r;
};

/* planning.h:236-260 */
Line line_coeff(P2 a, P2 b, bool normalize) {
    Line l;
    l.p = a.y - b.y;
    l.q = b.x - a.x;
    l.r = -l.p * a.x - l.q * a.y;
    if (normalize) {
        double norm = std::hypot(l.p, l.q);
        l.p /= norm;
        l.q /= norm;
        l.r /= norm;
    }
    return l;
}

/* ------------------------------------------------------------------ *
 * FootScheduler.h:15-99 (only what computeSteps uses)
 * ------------------------------------------------------------------ */
struct FootScheduler {
    int seq[4] = {0, 0, 0, 0};
    int cur = 0;
    void setSequence(int a, int b, int c, int d) { /* :45-52 */
        seq[0] = a;
        seq[1] = b;
        seq[2] = c;
        seq[3] = d;
        cur = 0;
    }
    void setCurrentSwing(int leg) { /* :33-43 */
        for (int i = 0; i < 4; i++)
            if (seq[i] == leg) {
                cur = i;
                break;
            }
    }
    int getCurrentSwing() const { return seq[cur]; } /* :28-31 */
    void next() { cur = (cur + 1 == 4) ? 0 : cur + 1; } /* :89-94 */
};

/* ------------------------------------------------------------------ *
 * MPCPlanner
 * ------------------------------------------------------------------ */
struct FootState {
    Vec x, y, swing;
};

struct Planner {
    int N;
    double Ts, g;
    double R = 1e-06, Q = 1.0, Qa = 1e-06, Qs = 100.0, height = 0.5; /* :39-43 */
    double A[3][3], B[3];
    double Cx[3] = {1, 0, 0}, Cv[3] = {0, 1, 0}, Ca[3] = {0, 0, 1}, Cz[3];
    bool dense = true;
    int iter_cap = 0;
    Vec all_violations;
    double sink = 0.0; /* keeps the reference's dead products alive */

    Planner(int N_, double Ts_, double g_) : N(N_), Ts(Ts_), g(g_) { /* :22-58 */
        A[0][0] = 1;
        A[0][1] = Ts;
        A[0][2] = std::pow(Ts, 2) / 2;
        A[1][0] = 0;
        A[1][1] = 1;
        A[1][2] = Ts;
        A[2][0] = 0;
        A[2][1] = 0;
        A[2][2] = 1;
        B[0] = std::pow(Ts, 3) / 6;
        B[1] = std::pow(Ts, 2) / 2;
        B[2] = Ts;
        setHeight(height);
    }
    void setHeight(double h) {
        height = h;
        Cz[0] = 1;
        Cz[1] = 0;
        Cz[2] = -height / g;
    }
    static void rowA(const double v[3], const double A[3][3], double out[3]) {
        double t[3];
        for (int j = 0; j < 3; j++) t[j] = v[0] * A[0][j] + v[1] * A[1][j] + v[2] * A[2][j];
        out[0] = t[0];
        out[1] = t[1];
        out[2] = t[2];
    }
    static double dot3(const double a[3], const double b[3]) {
        return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    }

    /* :101-156 */
    void buildMatrix(const double C[3], Mat &Sx, Mat &Su, int size) const {
        Sx = Mat(size, 3);
        Su = Mat(size, size);
        double D1[3], el[3], dv[3];
        rowA(C, A, D1);
        double D2 = dot3(C, B);
        for (int j = 0; j < 3; j++) {
            el[j] = D1[j];
            Sx(0, j) = D1[j];
        }
        for (int i = 1; i < size; i++) {
            rowA(el, A, el);
            for (int j = 0; j < 3; j++) Sx(i, j) = el[j];
        }
        for (int i = 0; i < size; i++) Su(i, i) = D2;
        if (size > 1) {
            double v1 = dot3(D1, B);
            for (int i = 0; i + 1 < size; i++) Su(i + 1, i) = v1;
            rowA(D1, A, dv);
            if (size > 2) {
                double v2 = dot3(dv, B);
                for (int i = 0; i + 2 < size; i++) Su(i + 2, i) = v2;
            }
            for (int k = 3; k < size; k++) {
                rowA(dv, A, dv);
                double vk = dot3(dv, B);
                for (int i = 0; i + k < size; i++) Su(i + k, i) = vk;
            }
        }
    }

    /* :169-174, :196-220 */
    Vec trajectory(const double C[3], const double x0[3], const Vec &jerk) const {
        Mat Sx, Su;
        const int n = (int)jerk.size();
        buildMatrix(C, Sx, Su, n);
        Vec x0v(x0, x0 + 3);
        Vec t = matvec(Sx, x0v), t2 = matvec(Su, jerk);
        for (int i = 0; i < n; i++) t[i] += t2[i];
        return t;
    }

    /* :1161-1171 */
    static Vec makeGaussian(int length, int mean, int stddev) {
        Vec w(length);
        for (int i = 0; i < length; i++) {
            double xi = 1.0 + (double)i; /* LinSpaced(length, 1, length): unit step */
            double q = (xi - mean) / stddev;
            double arg = -0.5 * std::pow(q, 2);
            w[i] = std::exp(arg) / (std::sqrt(2 * M_PI) * stddev);
        }
        return w;
    }

    /* :223-237.  G.inverse() is PartialPivLU in Eigen; restated as LU with
     * partial pivoting, explicit inverse, then the product. */
    void solveQP(double h, const double x0[3], const Vec &zref, Vec &jerk) {
        setHeight(h);
        Mat Zx, Zu;
        buildMatrix(Cz, Zx, Zu, N);
        Mat Zut = transpose(Zu);
        Mat G = matmul(Zut, Zu, dense);
        for (int i = 0; i < N; i++) G(i, i) += R / Q;
        Vec x0v(x0, x0 + 3);
        Vec e = matvec(Zx, x0v);
        for (int i = 0; i < N; i++) e[i] -= zref[i];
        Vec b = matvec(Zut, e);
        /* LU, partial pivoting */
        Mat LU = G;
        std::vector<int> piv(N);
        for (int i = 0; i < N; i++) piv[i] = i;
        for (int k = 0; k < N; k++) {
            int pr = k;
            double best = std::fabs(LU(k, k));
            for (int i = k + 1; i < N; i++)
                if (std::fabs(LU(i, k)) > best) {
                    best = std::fabs(LU(i, k));
                    pr = i;
                }
            if (pr != k) {
                for (int j = 0; j < N; j++) std::swap(LU(k, j), LU(pr, j));
                std::swap(piv[k], piv[pr]);
            }
            for (int i = k + 1; i < N; i++) {
                LU(i, k) /= LU(k, k);
                const double lik = LU(i, k);
                for (int j = k + 1; j < N; j++) LU(i, j) -= lik * LU(k, j);
            }
        }
        Mat Ginv(N, N);
        for (int c = 0; c < N; c++) {
            Vec y(N);
            for (int i = 0; i < N; i++) {
                double acc = (piv[i] == c) ? 1.0 : 0.0;
                for (int k = 0; k < i; k++) acc -= LU(i, k) * y[k];
                y[i] = acc;
            }
            for (int i = N - 1; i >= 0; i--) {
                double acc = y[i];
                for (int k = i + 1; k < N; k++) acc -= LU(i, k) * Ginv(k, c);
                Ginv(i, c) = acc / LU(i, i);
            }
        }
        jerk = matvec(Ginv, b);
        for (int i = 0; i < N; i++) jerk[i] = -jerk[i];
    }

    /* :240-300 */
    double solveQPconstraint(double h, const double x0[3], const Vec &zmin, const Vec &zmax,
                             Vec &jerk, orc_info *info, int *active) {
        setHeight(h);
        Mat Zx, Zu;
        buildMatrix(Cz, Zx, Zu, N);
        Mat GQ(N, N);
        for (int i = 0; i < N; i++) GQ(i, i) = R;
        Vec g0(N, 0.0);
        Mat CI(2 * N, N);
        Vec ci0(2 * N);
        Vec x0v(x0, x0 + 3);
        Vec zx0 = matvec(Zx, x0v);
        for (int j = 0; j < N; j++)
            for (int i = 0; i < N; i++) {
                CI(i, j) = Zu(i, j);
                CI(N + i, j) = -Zu(i, j);
            }
        for (int i = 0; i < N; i++) {
            ci0[i] = zx0[i] - zmin[i];
            ci0[N + i] = zmax[i] - zx0[i];
        }
        Mat CIt = transpose(CI);
        return solve_quadprog(GQ, g0, CIt, ci0, jerk, info, active, iter_cap);
    }

    /* solveQPconstraintSlack, :303-424 (the live "one slack per constraint" version,
       :351-394): x = [jerk (N); slacks (2N)], G = diag(R I, Qs I), g0 = [0; 1000 Qs],
       rows 0..2N-1  (+/-Zu jerk + slack_i + ci0_i >= 0), rows 2N..4N-1 (-slack_i >= 0).
       `sol` receives all 3N variables. */
    double solveQPconstraintSlack(double h, const double x0[3], const Vec &zmin, const Vec &zmax,
                                  Vec &sol, orc_info *info, int *active) {
        setHeight(h);
        Mat Zx, Zu;
        buildMatrix(Cz, Zx, Zu, N);
        const int ns = 2 * N, n = N + ns, m = 2 * N + ns;
        Mat GQ(n, n);
        for (int i = 0; i < N; i++) GQ(i, i) = R;
        for (int i = N; i < n; i++) GQ(i, i) = Qs;
        Vec g0(n, 0.0);
        for (int i = N; i < n; i++) g0[i] = 1000 * Qs;
        Mat CI(m, n);
        Vec ci0(m, 0.0);
        Vec x0v(x0, x0 + 3);
        Vec zx0 = matvec(Zx, x0v);
        for (int j = 0; j < N; j++)
            for (int i = 0; i < N; i++) {
                CI(i, j) = Zu(i, j);
                CI(N + i, j) = -Zu(i, j);
            }
        for (int i = 0; i < N; i++) {
            ci0[i] = zx0[i] - zmin[i];
            ci0[N + i] = zmax[i] - zx0[i];
        }
        for (int i = 0; i < ns; i++) {
            CI(i, N + i) = 1.0;
            CI(2 * N + i, N + i) = -1.0;
        }
        Mat CIt = transpose(CI);
        return solve_quadprog(GQ, g0, CIt, ci0, sol, info, active, iter_cap);
    }

    /* velocity block shared by :531-571 and :672-712 */
    void velocityTerms(const double x0[3], const double y0[3], const double speed[2], int rw,
                       Mat &Gv1, Vec &gvx, Vec &gvy) {
        const int window = N;
        Mat Xvx, Xvu;
        buildMatrix(Cv, Xvx, Xvu, N);
        Mat Wq(N, N);
        if (rw == 1000) {
            const double wv = Q * N / window;
            for (int i = 0; i < N; i++) Wq(i, i) = wv;
        } else {
            Vec gw = makeGaussian(N, rw, rw);
            const double sc = Q * N;
            for (int i = 0; i < N; i++) Wq(i, i) = gw[i] * sc;
        }
        Vec sel(N, 0.0);
        for (int i = N - window; i < N; i++) sel[i] = 1.0;
        Mat S = diag(sel);
        Mat Xvut = transpose(Xvu);
        /* ((((S*Xvu^T)*Wq)*Xvu)*S), evaluated left to right, once per block */
        Mat T1 = matmul(S, Xvut, dense);
        Mat T2 = matmul(T1, Wq, dense);
        Mat T3 = matmul(T2, Xvu, dense);
        Gv1 = matmul(T3, S, dense);
        if (dense) { /* the second, identical block (:559 / :700) */
            Mat U1 = matmul(S, Xvut, true);
            Mat U2 = matmul(U1, Wq, true);
            Mat U3 = matmul(U2, Xvu, true);
            Mat U4 = matmul(U3, S, true);
            sink += U4(N - 1, N - 1);
        }
        Vec x0v(x0, x0 + 3), y0v(y0, y0 + 3);
        Vec bx = matvec(Xvx, x0v), by = matvec(Xvx, y0v);
        for (int i = 0; i < N; i++) {
            bx[i] -= speed[0];
            by[i] -= speed[1];
        }
        gvx = vecmat(vecmat(vecmat(bx, Wq), Xvu), S);
        gvy = vecmat(vecmat(vecmat(by, Wq), Xvu), S);
        /* dead acceleration block (:576-580 / :717-721) */
        if (dense) {
            Mat Xax, Xau;
            buildMatrix(Ca, Xax, Xau, N);
            Mat Xaut = transpose(Xau);
            for (double &v : Xaut.a) v *= Qa;
            for (int rep = 0; rep < 2; rep++) {
                Mat Ga = matmul(Xaut, Xau, true);
                sink += Ga(N - 1, N - 1);
            }
        }
    }

    /* :427-490 (variant 0), :493-628 (1), :632-783 (2) */
    double solveCoupled(int variant, double h, const double x0[3], const double y0[3],
                        const Vec *zrefx, const Vec *zrefy, const Mat &Apoly, const Vec &b,
                        const double speed[2], int rw, Vec &jx, Vec &jy, orc_info *info,
                        int *active) {
        setHeight(h);
        const int n = 2 * N, m = (int)b.size();
        Mat Zx, Zu;
        buildMatrix(Cz, Zx, Zu, N);
        Mat GQ(n, n);
        Vec g0(n, 0.0);
        for (int i = 0; i < n; i++) GQ(i, i) = R; /* Gjerk */
        if (variant >= ORC_COUPLED_VEL) {
            Mat Gv1;
            Vec gvx, gvy;
            velocityTerms(x0, y0, speed, rw, Gv1, gvx, gvy);
            for (int j = 0; j < N; j++)
                for (int i = 0; i < N; i++) {
                    GQ(i, j) += Gv1(i, j);
                    GQ(N + i, N + j) += Gv1(i, j);
                }
            for (int i = 0; i < N; i++) {
                g0[i] = gvx[i];
                g0[N + i] = gvy[i];
            }
        }
        if (variant == ORC_COUPLED_REF) { /* :723-740 */
            Mat Zut = transpose(Zu);
            for (double &v : Zut.a) v *= Qs;
            Mat Gr = matmul(Zut, Zu, dense);
            if (dense) {
                Mat Gr2 = matmul(Zut, Zu, true);
                sink += Gr2(N - 1, N - 1);
            }
            Vec x0v(x0, x0 + 3), y0v(y0, y0 + 3);
            Vec bx = matvec(Zx, x0v), by = matvec(Zx, y0v);
            for (int i = 0; i < N; i++) {
                bx[i] = (bx[i] - (*zrefx)[i]) * Qs;
                by[i] = (by[i] - (*zrefy)[i]) * Qs;
            }
            Vec grx = vecmat(bx, Zu), gry = vecmat(by, Zu);
            /* GQ = Gjerk + Gv + Gref ; g0 = gv + gref */
            for (int j = 0; j < N; j++)
                for (int i = 0; i < N; i++) {
                    GQ(i, j) += Gr(i, j);
                    GQ(N + i, N + j) += Gr(i, j);
                }
            for (int i = 0; i < N; i++) {
                g0[i] += grx[i];
                g0[N + i] += gry[i];
            }
        }
        /* Zuc, Zxc, Zyc; CI = A*Zuc; ci0 = b + A*(Zxc*x0 + Zyc*y0) */
        Mat Zuc(n, n);
        for (int j = 0; j < N; j++)
            for (int i = 0; i < N; i++) {
                Zuc(i, j) = Zu(i, j);
                Zuc(N + i, N + j) = Zu(i, j);
            }
        Mat CI = matmul(Apoly, Zuc, dense);
        Vec x0v(x0, x0 + 3), y0v(y0, y0 + 3);
        Vec zx = matvec(Zx, x0v), zy = matvec(Zx, y0v);
        Vec zz(n);
        for (int i = 0; i < N; i++) {
            zz[i] = zx[i];
            zz[N + i] = zy[i];
        }
        Vec ci0 = matvec(Apoly, zz);
        for (int i = 0; i < m; i++) ci0[i] = b[i] + ci0[i];
        Mat CIt = transpose(CI);
        Vec sol;
        double f = solve_quadprog(GQ, g0, CIt, ci0, sol, info, active, iter_cap);
        jx.assign(N, 0.0);
        jy.assign(N, 0.0);
        if (f != kInf) {
            for (int i = 0; i < N; i++) {
                jx[i] = sol[i];
                jy[i] = sol[N + i];
            }
        } else {
            /* the reference leaves the jerk vectors unassigned here; the
             * oracle zero-fills them and evaluates the violations at zero */
            std::fill(sol.begin(), sol.end(), 0.0);
        }
        all_violations = matvec(CI, sol);
        for (int i = 0; i < m; i++) all_violations[i] += ci0[i];
        if (info) info->falg += 2.0 * m * (double)n;
        return f;
    }

    /* :1040-1082 */
    int buildPolygonMatrix(const FootState fs[4], int start, int duration, int horizon, Mat &Ap,
                           Vec &b, int &nc) const {
        if (duration == 0) return 0;
        std::vector<P2> feet;
        for (int leg = 0; leg < 4; leg++)
            if (!fs[leg].swing[start]) feet.push_back(P2{fs[leg].x[start], fs[leg].y[start]});
        const int E = (int)feet.size();
        std::vector<Line> lc(E);
        ccw_sort(feet);
        for (int e = 0; e < E; e++) lc[e] = line_coeff(feet[e], feet[(e + 1) % E], true);
        for (int i = 0; i < duration; i++) {
            if (nc + E > Ap.r) return -2;
            for (int e = 0; e < E; e++) {
                Ap(nc + e, start + i) = lc[e].p;
                Ap(nc + e, start + i + horizon) = lc[e].q;
                b[nc + e] = lc[e].r;
            }
            nc += E;
        }
        return 0;
    }

    /* :1188-1329 */
    int computeSteps(const double speed[2], const double feet_x[4], const double feet_y[4], int steps,
                     int horizon, FootState fs[4], FootState holds[4], Mat &Ap, Vec &b, int swing0,
                     const double com0[2], const double hip[8]) const {
        if (steps <= 0 || horizon <= 0) return -1;
        FootScheduler schedule;
        schedule.setSequence(ORC_LF, ORC_RH, ORC_RF, ORC_LH);
        schedule.setCurrentSwing(swing0);
        const int step_knots = horizon / steps;
        double fvx[4], fvy[4];
        for (int l = 0; l < 4; l++) {
            fvx[l] = feet_x[l];
            fvy[l] = feet_y[l];
        }
        int nc = 0, start = 0;
        const int phase = step_knots / 2;
        Ap = Mat(4 * step_knots * steps, horizon * 2);
        b.assign((size_t)4 * step_knots * steps, 0.0);
        for (int l = 0; l < 4; l++) {
            fs[l].x.assign(horizon, 0.0);
            fs[l].y.assign(horizon, 0.0);
            fs[l].swing.assign(horizon, 0.0);
            holds[l].x.assign(steps, 0.0);
            holds[l].y.assign(steps, 0.0);
        }
        bool flag[4] = {false, false, false, false};
        double corr[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        if (!(com0[0] == 0.0 && com0[1] == 0.0)) { /* :1231-1238 */
            for (int l = 0; l < 4; l++) {
                flag[l] = true;
                corr[l][0] = ((com0[0] + hip[2 * l]) + speed[0]) - fvx[l];
                corr[l][1] = ((com0[1] + hip[2 * l + 1]) + speed[1]) - fvy[l];
            }
        }
        for (int i = 0; i < steps; i++) {
            for (int l = 0; l < 4; l++)
                for (int k = start; k < start + phase; k++) {
                    fs[l].x[k] = fvx[l];
                    fs[l].y[k] = fvy[l];
                }
            if (buildPolygonMatrix(fs, start, phase, horizon, Ap, b, nc)) return -2;
            start += phase;
            const int sw = schedule.getCurrentSwing();
            double dx, dy;
            if (flag[sw]) { /* :1266-1280 */
                dx = corr[sw][0];
                dy = corr[sw][1];
                flag[sw] = false;
            } else {
                dx = speed[0];
                dy = speed[1];
            }
            fvx[sw] += dx;
            fvy[sw] += dy;
            for (int l = 0; l < 4; l++) {
                holds[l].x[i] = fvx[l];
                holds[l].y[i] = fvy[l];
            }
            for (int k = start; k < start + phase; k++) fs[sw].swing[k] = 1.0;
            for (int l = 0; l < 4; l++)
                for (int k = start; k < start + phase; k++) {
                    fs[l].x[k] = fvx[l];
                    fs[l].y[k] = fvy[l];
                }
            if (buildPolygonMatrix(fs, start, phase, horizon, Ap, b, nc)) return -2;
            start += phase;
            schedule.next();
        }
        const int missing = horizon - start; /* :1309-1321 */
        for (int l = 0; l < 4; l++)
            for (int k = start; k < start + missing; k++) {
                fs[l].x[k] = fvx[l];
                fs[l].y[k] = fvy[l];
            }
        if (buildPolygonMatrix(fs, start, missing, horizon, Ap, b, nc)) return -2;
        /* conservativeResize to nc rows (:1324-1326) */
        Mat A2(nc, horizon * 2);
        for (int j = 0; j < horizon * 2; j++)
            for (int i = 0; i < nc; i++) A2(i, j) = Ap(i, j);
        Ap = A2;
        b.resize(nc);
        return nc;
    }

    /* :1106-1130 */
    static void computeCentroid(int N, const FootState fs[4], Vec &cx, Vec &cy) {
        cx.assign(N, 0.0);
        cy.assign(N, 0.0);
        for (int i = 0; i < N; i++) {
            int ns = 0;
            double sx = 0.0, sy = 0.0;
            for (int l = 0; l < 4; l++)
                if (!fs[l].swing[i]) {
                    sx += fs[l].x[i];
                    sy += fs[l].y[i];
                    ns++;
                }
            cx[i] = sx / ns;
            cy[i] = sy / ns;
        }
    }

    /* :1132-1153 */
    static void constraintViolation(int N, const FootState fs[4], const Vec &viol, Vec &out) {
        out.assign(N, 0.0);
        int nc = 0;
        for (int i = 0; i < N; i++) {
            int ne = 0;
            for (int l = 0; l < 4; l++)
                if (!fs[l].swing[i]) ne++;
            double mn = kInf;
            for (int e = 0; e < ne; e++) mn = std::min(mn, viol[nc + e]);
            nc += ne;
            out[i] = mn;
        }
    }
};

Planner make_planner(const orc_params *p) {
    Planner pl(p->N, p->Ts, p->g);
    pl.R = p->R;
    pl.Q = p->Q;
    pl.Qs = p->Qs;
    pl.dense = p->dense != 0;
    pl.iter_cap = p->iter_cap;
    pl.setHeight(p->height);
    return pl;
}

const double *pick_C(const Planner &pl, int which) {
    switch (which) {
    case 0: return pl.Cx;
    case 1: return pl.Cv;
    case 2: return pl.Ca;
    default: return pl.Cz;
    }
}

Mat from_rowmajor(const double *a, int r, int c) {
    Mat M(r, c);
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) M(i, j) = a[(size_t)i * c + j];
    return M;
}

void to_rowmajor(const Mat &M, double *a) {
    for (int i = 0; i < M.r; i++)
        for (int j = 0; j < M.c; j++) a[(size_t)i * M.c + j] = M(i, j);
}

template <class F> void parallel_for(int B, int nthreads, F fn) {
    if (nthreads <= 1) {
        fn(0, B, 0);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) {
        int lo = (int)((long long)B * t / nthreads), hi = (int)((long long)B * (t + 1) / nthreads);
        th.emplace_back(fn, lo, hi, t);
    }
    for (auto &t : th) t.join();
}

} // namespace

/* ------------------------------------------------------------------ *
 * C interface
 * ------------------------------------------------------------------ */
extern "C" {

int orc_build_matrix(const orc_params *p, int which, int size, double *Sx, double *Su) {
    Planner pl = make_planner(p);
    Mat mx, mu;
    pl.buildMatrix(pick_C(pl, which), mx, mu, size);
    if (Sx) to_rowmajor(mx, Sx);
    if (Su) to_rowmajor(mu, Su);
    return 0;
}

int orc_trajectory(const orc_params *p, int which, const double x0[3], const double *jerk, int n,
                   double *traj) {
    Planner pl = make_planner(p);
    Vec j(jerk, jerk + n);
    Vec t = pl.trajectory(pick_C(pl, which), x0, j);
    std::copy(t.begin(), t.end(), traj);
    return 0;
}

void orc_com_update(const orc_params *p, double st[3], double u) {
    Planner pl = make_planner(p);
    double o[3];
    for (int i = 0; i < 3; i++)
        o[i] = (pl.A[i][0] * st[0] + pl.A[i][1] * st[1] + pl.A[i][2] * st[2]) + pl.B[i] * u;
    st[0] = o[0];
    st[1] = o[1];
    st[2] = o[2];
}

void orc_make_gaussian(int length, int mean, int stddev, double *w) {
    Vec g = Planner::makeGaussian(length, mean, stddev);
    std::copy(g.begin(), g.end(), w);
}

int orc_solve_qp(const orc_params *p, const double x0[3], const double *zref, double *jerk) {
    Planner pl = make_planner(p);
    Vec z(zref, zref + p->N), j;
    pl.solveQP(p->height, x0, z, j);
    std::copy(j.begin(), j.end(), jerk);
    return 0;
}

int orc_solve_box(const orc_params *p, const double x0[3], const double *zmin, const double *zmax,
                  double *jerk, orc_info *info, int *active) {
    Planner pl = make_planner(p);
    Vec lo(zmin, zmin + p->N), hi(zmax, zmax + p->N), j;
    orc_info li;
    pl.solveQPconstraint(p->height, x0, lo, hi, j, &li, active);
    if (li.status != ORC_OK) std::fill(j.begin(), j.end(), 0.0);
    std::copy(j.begin(), j.end(), jerk);
    if (info) *info = li;
    return li.status;
}

int orc_solve_box_trace(const orc_params *p, const double x0[3], const double *zmin, const double *zmax,
                        double *jerk, orc_info *info, int *active, int max_rows, double *rec) {
    QPTrace tr;
    tr.max_rows = max_rows;
    tr.rec = rec;
    g_trace = &tr;
    orc_solve_box(p, x0, zmin, zmax, jerk, info, active);
    g_trace = nullptr;
    return tr.rows;
}

double orc_point2_is_right_of_line(const double p0[2], const double p1[2], const double p2[2]) {
    return point2_is_right_of_line(P2{p0[0], p0[1]}, P2{p1[0], p1[1]}, P2{p2[0], p2[1]});
}

void orc_ccw_sort(double *pts, int n) {
    std::vector<P2> v(n);
    for (int i = 0; i < n; i++) v[i] = P2{pts[2 * i], pts[2 * i + 1]};
    ccw_sort(v);
    for (int i = 0; i < n; i++) {
        pts[2 * i] = v[i].x;
        pts[2 * i + 1] = v[i].y;
    }
}

void orc_line_coeff(const double p0[2], const double p1[2], int normalize, double pqr[3]) {
    Line l = line_coeff(P2{p0[0], p0[1]}, P2{p1[0], p1[1]}, normalize != 0);
    pqr[0] = l.p;
    pqr[1] = l.q;
    pqr[2] = l.r;
}

int orc_compute_steps(int N, int steps, const double speed[2], const double feet_x[4],
                      const double feet_y[4], int swing0, const double com0[2], const double *hip,
                      double *A, double *b, double *fx, double *fy, double *swing, double *hold_x,
                      double *hold_y) {
    Planner pl(N, 0.1, 9.81);
    FootState fs[4], holds[4];
    Mat Ap;
    Vec bv;
    const double zero2[2] = {0, 0};
    const double zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int m = pl.computeSteps(speed, feet_x, feet_y, steps, N, fs, holds, Ap, bv, swing0,
                            com0 ? com0 : zero2, hip ? hip : zero8);
    if (m < 0) return m;
    if (A) to_rowmajor(Ap, A);
    if (b) std::copy(bv.begin(), bv.end(), b);
    for (int l = 0; l < 4; l++) {
        if (fx) std::copy(fs[l].x.begin(), fs[l].x.end(), fx + (size_t)l * N);
        if (fy) std::copy(fs[l].y.begin(), fs[l].y.end(), fy + (size_t)l * N);
        if (swing) std::copy(fs[l].swing.begin(), fs[l].swing.end(), swing + (size_t)l * N);
        if (hold_x) std::copy(holds[l].x.begin(), holds[l].x.end(), hold_x + (size_t)l * steps);
        if (hold_y) std::copy(holds[l].y.begin(), holds[l].y.end(), hold_y + (size_t)l * steps);
    }
    return m;
}

static void load_feet(int N, const double *fx, const double *fy, const double *swing,
                      FootState fs[4]) {
    for (int l = 0; l < 4; l++) {
        if (fx) fs[l].x.assign(fx + (size_t)l * N, fx + (size_t)(l + 1) * N);
        if (fy) fs[l].y.assign(fy + (size_t)l * N, fy + (size_t)(l + 1) * N);
        fs[l].swing.assign(swing + (size_t)l * N, swing + (size_t)(l + 1) * N);
    }
}

void orc_centroid(int N, const double *fx, const double *fy, const double *swing, double *cx,
                  double *cy) {
    FootState fs[4];
    load_feet(N, fx, fy, swing, fs);
    Vec vx, vy;
    Planner::computeCentroid(N, fs, vx, vy);
    std::copy(vx.begin(), vx.end(), cx);
    std::copy(vy.begin(), vy.end(), cy);
}

void orc_constraint_violation(int N, const double *swing, const double *viol_all,
                              double *viol_knot) {
    FootState fs[4];
    load_feet(N, nullptr, nullptr, swing, fs);
    int m = 0;
    for (int i = 0; i < N; i++)
        for (int l = 0; l < 4; l++)
            if (!fs[l].swing[i]) m++;
    Vec v(viol_all, viol_all + m), out;
    Planner::constraintViolation(N, fs, v, out);
    std::copy(out.begin(), out.end(), viol_knot);
}

int orc_solve_coupled(const orc_params *p, int variant, const double x0[3], const double y0[3],
                      const double *zrefx, const double *zrefy, const double *A, const double *b,
                      int m, const double speed[2], int rw, double *jerk_x, double *jerk_y,
                      double *viol_all, orc_info *info, int *active) {
    Planner pl = make_planner(p);
    const int N = p->N;
    Mat Ap = from_rowmajor(A, m, 2 * N);
    Vec bv(b, b + m), zx, zy, jx, jy;
    if (variant == ORC_COUPLED_REF) {
        zx.assign(zrefx, zrefx + N);
        zy.assign(zrefy, zrefy + N);
    }
    const double zero2[2] = {0, 0};
    orc_info li;
    pl.solveCoupled(variant, p->height, x0, y0, &zx, &zy, Ap, bv, speed ? speed : zero2, rw, jx,
                    jy, &li, active);
    std::copy(jx.begin(), jx.end(), jerk_x);
    std::copy(jy.begin(), jy.end(), jerk_y);
    if (viol_all) std::copy(pl.all_violations.begin(), pl.all_violations.end(), viol_all);
    if (info) *info = li;
    return li.status;
}

int orc_solve_quadprog(int n, int m, const double *G, const double *g0, const double *CI,
                       const double *ci0, double *x, orc_info *info, int *active, int iter_cap) {
    Mat Gm = from_rowmajor(G, n, n), CIm = from_rowmajor(CI, n, m);
    Vec g(g0, g0 + n), c(ci0, ci0 + m), xs;
    orc_info li;
    solve_quadprog(Gm, g, CIm, c, xs, &li, active, iter_cap);
    std::copy(xs.begin(), xs.end(), x);
    if (info) *info = li;
    return li.status;
}

int orc_coupled_batch(const orc_params *p, int variant, int steps, int rw, int B, const double *x0,
                      const double *y0, const double *speed, const double *feet_x,
                      const double *feet_y, const int *swing0, const double *com0,
                      const double *hip, double *jerk_x,
                      double *jerk_y, double *com_x, double *com_y, double *zmp_x, double *zmp_y,
                      double *viol_knot, double *cost, int *status, int *n_active, int *active,
                      int active_stride, int *iters, double *falg, int nthreads) {
    const int N = p->N;
    std::vector<int> err(std::max(1, nthreads), 0);
    parallel_for(B, nthreads, [&](int lo, int hi, int tid) {
        Planner pl = make_planner(p);
        std::vector<int> act;
        for (int i = lo; i < hi; i++) {
            FootState fs[4], holds[4];
            Mat Ap;
            Vec bv, cx, cy, jx, jy, viol;
            const double zero2[2] = {0, 0};
            const double zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int m = pl.computeSteps(speed + 2 * (size_t)i, feet_x + 4 * (size_t)i,
                                    feet_y + 4 * (size_t)i, steps, N, fs, holds, Ap, bv,
                                    swing0 ? swing0[i] : ORC_LF,
                                    com0 ? com0 + 2 * (size_t)i : zero2, hip ? hip : zero8);
            if (m < 0) {
                err[tid] = m;
                return;
            }
            if (variant == ORC_COUPLED_REF) Planner::computeCentroid(N, fs, cx, cy);
            act.assign(m + 1, 0);
            orc_info li;
            pl.solveCoupled(variant, p->height, x0 + 3 * (size_t)i, y0 + 3 * (size_t)i, &cx, &cy,
                            Ap, bv, speed + 2 * (size_t)i, rw, jx, jy, &li, act.data());
            Planner::constraintViolation(N, fs, pl.all_violations, viol);
            const size_t o = (size_t)i * N;
            if (jerk_x) std::copy(jx.begin(), jx.end(), jerk_x + o);
            if (jerk_y) std::copy(jy.begin(), jy.end(), jerk_y + o);
            if (com_x) {
                Vec t = pl.trajectory(pl.Cx, x0 + 3 * (size_t)i, jx);
                std::copy(t.begin(), t.end(), com_x + o);
            }
            if (com_y) {
                Vec t = pl.trajectory(pl.Cx, y0 + 3 * (size_t)i, jy);
                std::copy(t.begin(), t.end(), com_y + o);
            }
            if (zmp_x) {
                Vec t = pl.trajectory(pl.Cz, x0 + 3 * (size_t)i, jx);
                std::copy(t.begin(), t.end(), zmp_x + o);
            }
            if (zmp_y) {
                Vec t = pl.trajectory(pl.Cz, y0 + 3 * (size_t)i, jy);
                std::copy(t.begin(), t.end(), zmp_y + o);
            }
            if (viol_knot) std::copy(viol.begin(), viol.end(), viol_knot + o);
            if (cost) cost[i] = li.f;
            if (status) status[i] = li.status;
            if (n_active) n_active[i] = li.n_active;
            if (iters) iters[i] = li.iters;
            if (falg) falg[i] = li.falg;
            if (active)
                for (int k = 0; k < active_stride; k++)
                    active[(size_t)i * active_stride + k] = (k < li.n_active) ? act[k] : -1;
        }
    });
    for (int e : err)
        if (e) return e;
    return 0;
}

int orc_replanning_batch(const orc_params *p, int steps, int n_stages, int use_com_correction,
                         const double *hip, int B, double *x0, double *y0, const double *speed,
                         double *feet_x, double *feet_y, int *swing0, const double *disturbance,
                         int disturbance_start, double *stage_cost, int *stage_status,
                         int *stage_nactive, double *jerk_x, double *jerk_y, int nthreads) {
    const int N = p->N;
    const int rw = N / steps; /* replanningWindow (testPlannerXYconstraintMPC.cpp:95) */
    const double zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!hip) hip = zero8;
    std::vector<int> err(std::max(1, nthreads), 0);
    parallel_for(B, nthreads, [&](int lo, int hi, int tid) {
        Planner pl = make_planner(p);
        for (int i = lo; i < hi; i++) {
            double sx[3] = {x0[3 * i], x0[3 * i + 1], x0[3 * i + 2]};
            double sy[3] = {y0[3 * i], y0[3 * i + 1], y0[3 * i + 2]};
            double fx[4], fy[4];
            for (int l = 0; l < 4; l++) {
                fx[l] = feet_x[4 * i + l];
                fy[l] = feet_y[4 * i + l];
            }
            FootScheduler sched;
            sched.setSequence(ORC_LF, ORC_RH, ORC_RF, ORC_LH); /* :45 */
            sched.setCurrentSwing(swing0[i]);
            FootState fs[4], holds[4];
            Vec jx(N, 0.0), jy(N, 0.0);
            int sampleW = 0;
            for (int sample = 0; sample < n_stages * rw; sample++) {
                if (sample % rw == 0) { /* :112 */
                    const int stage = sample / rw;
                    if (sample > 0) { /* :117-134 */
                        for (int l = 0; l < 4; l++) {
                            fx[l] = fs[l].x[sampleW];
                            fy[l] = fs[l].y[sampleW];
                        }
                        sched.next();
                    }
                    Mat Ap;
                    Vec bv;
                    const double zero2[2] = {0, 0};
                    const double com[2] = {sx[0], sy[0]};
                    int m = pl.computeSteps(speed + 2 * (size_t)i, fx, fy, steps, N, fs, holds, Ap, bv,
                                            sched.getCurrentSwing(), use_com_correction ? com : zero2, hip);
                    if (m < 0) {
                        err[tid] = m;
                        return;
                    }
                    std::vector<int> act(m + 1);
                    orc_info li;
                    pl.solveCoupled(ORC_COUPLED_VEL, p->height, sx, sy, nullptr, nullptr, Ap, bv,
                                    speed + 2 * (size_t)i, rw, jx, jy, &li, act.data());
                    if (stage_cost) stage_cost[(size_t)stage * B + i] = li.f;
                    if (stage_status) stage_status[(size_t)stage * B + i] = li.status;
                    if (stage_nactive) stage_nactive[(size_t)stage * B + i] = li.n_active;
                    sampleW = 0;
                } else
                    sampleW++;
                /* :206-207 computeCOMupdate */
                const double ux = jx[sampleW];
                const double uy = jy[sampleW] + (sample >= disturbance_start ? disturbance[i] : 0.0);
                double nx[3], ny[3];
                for (int r3 = 0; r3 < 3; r3++) {
                    nx[r3] = (pl.A[r3][0] * sx[0] + pl.A[r3][1] * sx[1] + pl.A[r3][2] * sx[2]) + pl.B[r3] * ux;
                    ny[r3] = (pl.A[r3][0] * sy[0] + pl.A[r3][1] * sy[1] + pl.A[r3][2] * sy[2]) + pl.B[r3] * uy;
                }
                for (int r3 = 0; r3 < 3; r3++) {
                    sx[r3] = nx[r3];
                    sy[r3] = ny[r3];
                }
            }
            /* state the next replan would start from */
            for (int l = 0; l < 4; l++) {
                feet_x[4 * i + l] = fs[l].x[sampleW];
                feet_y[4 * i + l] = fs[l].y[sampleW];
            }
            sched.next();
            swing0[i] = sched.getCurrentSwing();
            for (int r3 = 0; r3 < 3; r3++) {
                x0[3 * i + r3] = sx[r3];
                y0[3 * i + r3] = sy[r3];
            }
            if (jerk_x) std::copy(jx.begin(), jx.end(), jerk_x + (size_t)i * N);
            if (jerk_y) std::copy(jy.begin(), jy.end(), jerk_y + (size_t)i * N);
        }
    });
    for (int e : err)
        if (e) return e;
    return 0;
}

int orc_box_batch(const orc_params *p, int B, const double *x0, const double *y0,
                  const double *xmin, const double *xmax, const double *ymin, const double *ymax,
                  double *jerk_x, double *jerk_y, double *cost, int *status, int *n_active,
                  int *active, int active_stride, int *iters, double *falg, int nthreads) {
    const int N = p->N;
    parallel_for(B, nthreads, [&](int lo, int hi, int) {
        Planner pl = make_planner(p);
        std::vector<int> act(2 * N + 1);
        for (int i = lo; i < hi; i++) {
            for (int ax = 0; ax < 2; ax++) {
                const double *s0 = (ax ? y0 : x0) + 3 * (size_t)i;
                const double *mn = (ax ? ymin : xmin) + (size_t)i * N;
                const double *mx = (ax ? ymax : xmax) + (size_t)i * N;
                Vec lo_(mn, mn + N), hi_(mx, mx + N), j;
                orc_info li;
                pl.solveQPconstraint(p->height, s0, lo_, hi_, j, &li, act.data());
                if (li.status != ORC_OK) std::fill(j.begin(), j.end(), 0.0);
                double *dst = (ax ? jerk_y : jerk_x);
                if (dst) std::copy(j.begin(), j.end(), dst + (size_t)i * N);
                const size_t o = (size_t)i * 2 + ax;
                if (cost) cost[o] = li.f;
                if (status) status[o] = li.status;
                if (n_active) n_active[o] = li.n_active;
                if (iters) iters[o] = li.iters;
                if (falg) falg[o] = li.falg;
                if (active)
                    for (int k = 0; k < active_stride; k++)
                        active[o * active_stride + k] = (k < li.n_active) ? act[k] : -1;
            }
        }
    });
    return 0;
}

/* solveQPconstraintSlack for B instances of one axis.  sol: [B][3N] (jerk, then the 2N
   slacks); active: [B][active_stride]. */
int orc_slack_batch(const orc_params *p, int B, const double *x0, const double *zmin,
                    const double *zmax, double *sol, double *cost, int *status, int *n_active,
                    int *active, int active_stride, int *iters, double *falg, int nthreads) {
    const int N = p->N;
    parallel_for(B, nthreads, [&](int lo, int hi, int) {
        Planner pl = make_planner(p);
        std::vector<int> act(4 * N + 1);
        for (int i = lo; i < hi; i++) {
            const double *mn = zmin + (size_t)i * N, *mx = zmax + (size_t)i * N;
            Vec lo_(mn, mn + N), hi_(mx, mx + N), x;
            orc_info li;
            pl.solveQPconstraintSlack(p->height, x0 + 3 * (size_t)i, lo_, hi_, x, &li, act.data());
            if (li.status != ORC_OK) std::fill(x.begin(), x.end(), 0.0);
            if (sol) std::copy(x.begin(), x.end(), sol + (size_t)i * 3 * N);
            if (cost) cost[i] = li.f;
            if (status) status[i] = li.status;
            if (n_active) n_active[i] = li.n_active;
            if (iters) iters[i] = li.iters;
            if (falg) falg[i] = li.falg;
            if (active)
                for (int k = 0; k < active_stride; k++)
                    active[(size_t)i * active_stride + k] = (k < li.n_active) ? act[k] : -1;
        }
    });
    return 0;
}

} /* extern "C" */

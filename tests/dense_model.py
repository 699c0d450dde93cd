"""Executable model (numpy) of the dense second-stage kernel's factor updates.

The reference's dual active-set solver (eiquadprog.hpp:130-592) keeps J = L^-T Q (n x n) and
the triangular R of the working set, and updates both by sequences of Givens rotations
(add_constraint :450-513, delete_constraint :516-592).  The LDS-resident HIP kernel keeps
the same J but
  * adds a constraint with ONE Householder reflection of the free columns J2 (a rank-1
    update, J2 <- J2 - w v'), which leaves span(J2) and therefore z = J2 J2' n+ unchanged and
    makes the new column iq equal to z / (+-||d2||), and
  * keeps T = R^-1 instead of R: r = T d1 is a product instead of a back substitution, an
    added constraint appends the column [-r; 1] / rho, and a dropped constraint qq rotates
    the columns qq .. iq-1 of T and of J by the plane rotations that move row qq of T into
    its last entry, after which row qq and the last column of T go.
Pivot rules, tolerances and step formulas are the reference's.  This file lets the CPU tests
compare that variant with the oracle on the box problems (G = R I, so L^-T = I / sqrt(R)).
"""
import numpy as np

EPS = np.finfo(np.float64).eps
OK, INFEASIBLE, ITER_CAP = 0, 1, 2


def solve_box(Zx, zcol, weight_R, x0, zmin, zmax, iter_cap=0):
    """solveQPconstraint (MPCPlanner.cpp:240-300) with the factor updates described above.
    Zx: N x 3 state matrix, zcol: Toeplitz first column of Zu.  Returns dict(jerk, status,
    iters, passes, active, f)."""
    N = zcol.size
    n, m = N, 2 * N
    Zu = np.zeros((N, N))
    for k in range(N):
        Zu += np.diag(np.full(N - k, zcol[k]), -k)
    zfree = Zx @ x0
    J = np.eye(n) / np.sqrt(weight_R)
    c1 = n * weight_R
    c2 = n / np.sqrt(weight_R)
    x = np.zeros(n)
    f = 0.0
    T = np.zeros((n, n))
    A = []
    u = []
    R_norm = 1.0
    iters = passes = 0
    status = OK

    def normal(i):
        k = i if i < N else i - N
        v = np.zeros(n)
        v[:k + 1] = zcol[k::-1] * (1.0 if i < N else -1.0)
        return v

    def rows(xv):
        z = zfree + Zu @ xv
        return np.concatenate([z - zmin, zmax - z])

    def drop(l):
        nonlocal T, J
        qq = A.index(l)
        iq = len(A)
        del A[qq]
        del u[qq]
        g = T[qq, qq:iq].copy()
        # rotations in planes (j, j+1), j = qq .. iq-2, that move g into its last entry
        rho = abs(g[0])
        for jj in range(iq - 1 - qq):
            j = qq + jj
            a, b = (g[0] if jj == 0 else rho), g[jj + 1]
            h = np.hypot(a, b)
            if h == 0.0:
                continue
            c, s = b / h, a / h      # (a, b) -> (0, h): [c -s; s c] applied to columns
            tj, tj1 = T[:iq, j].copy(), T[:iq, j + 1].copy()
            T[:iq, j] = c * tj - s * tj1
            T[:iq, j + 1] = s * tj + c * tj1
            cj, cj1 = J[:, j].copy(), J[:, j + 1].copy()
            J[:, j] = c * cj - s * cj1
            J[:, j + 1] = s * cj + c * cj1
            rho = h
        # remove row qq and the last column
        T[qq:iq - 1, :] = T[qq + 1:iq, :]
        T[iq - 1, :] = 0.0
        T[:, iq - 1] = 0.0

    while True:
        iters += 1
        if iter_cap and iters > iter_cap:
            status = ITER_CAP
            break
        s = rows(x)
        excl = np.ones(m, dtype=bool)
        psi = np.minimum(0.0, s).sum()
        if abs(psi) <= m * EPS * c1 * c2 * 100.0:
            break
        A_old, u_old, x_old = list(A), list(u), x.copy()
        ss_, ip = 0.0, 0
        restart = True
        done = False
        while restart:
            restart = False
            inactive = np.ones(m, dtype=bool)
            inactive[A] = False
            cand = np.where(inactive & excl, s, np.inf)
            i = int(np.argmin(cand))
            if cand[i] < ss_:
                ss_, ip = cand[i], i
            if ss_ >= 0.0:
                done = True
                break
            u = u[:len(A)] + [0.0]
            npv = normal(ip)
            s_ip = s[ip]
            while True:
                passes += 1
                if iter_cap and passes > iter_cap:
                    status = ITER_CAP
                    done = True
                    break
                iq = len(A)
                d = J.T @ npv
                z = J[:, iq:] @ d[iq:]
                r = T[:iq, :iq] @ d[:iq]
                t1, l = np.inf, 0
                for k in range(iq):
                    if r[k] > 0.0 and u[k] / r[k] < t1:
                        t1, l = u[k] / r[k], A[k]
                zn = z @ npv
                t2 = -s_ip / zn if abs(z @ z) > EPS else np.inf
                t = min(t1, t2)
                if t >= np.inf:
                    status = INFEASIBLE
                    f = np.inf
                    done = True
                    break
                if t2 >= np.inf:
                    for k in range(iq):
                        u[k] -= t * r[k]
                    u[iq] += t
                    drop(l)
                    continue
                x = x + t * z
                f += t * zn * (0.5 * t + u[iq])
                for k in range(iq):
                    u[k] -= t * r[k]
                u[iq] += t
                if t == t2:
                    d2 = d[iq:]
                    nrm = np.sqrt(d2 @ d2)
                    alpha = -nrm if d2[0] >= 0.0 else nrm
                    if n - iq > 1 and nrm > 0.0:
                        v = d2.copy()
                        v[0] -= alpha
                        beta = 1.0 / (nrm * (nrm + abs(d2[0])))   # 2 / v'v
                        w = beta * (z - alpha * J[:, iq])
                        J[:, iq:] -= np.outer(w, v)
                    rho = alpha if n - iq > 1 else d2[0]
                    if abs(rho) <= EPS * R_norm:
                        excl[ip] = False
                        nq = len(A)
                        A = A_old[:nq]
                        u = u_old[:nq]
                        x = x_old.copy()
                        restart = True
                        break
                    T[:iq, iq] = -r / rho
                    T[iq, iq] = 1.0 / rho
                    A.append(ip)
                    R_norm = max(R_norm, abs(rho))
                    break
                s_ip_old = s_ip
                drop(l)
                s_ip = rows(x)[ip]
                del s_ip_old
            if done:
                break
        if done:
            break
    return dict(jerk=x if status != INFEASIBLE else np.zeros(n), status=status, iters=iters,
                passes=passes, active=np.array(A[:len(A)], dtype=np.int64), f=f)

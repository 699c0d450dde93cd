"""Executable model (numpy, scalar) of the solver the HIP kernel implements.

The condensed QP of the planner has a block-diagonal Hessian G = blkdiag(G1, G1)
that does not depend on the instance, and every inequality row is a half-plane on
the ZMP at ONE knot:  s_j = p_j*zx[k_j] + q_j*zy[k_j] + r_j >= 0  with
zx = Zx*x0 + Zu*jerk_x.  Writing the dual active-set method of Goldfarb & Idnani
in terms of the multipliers u of the working set A gives

    jerk = jerk_unc + W[:, k_A] * (p_A * u)          W = G1^-1 Zu^T      (N x N)
    zmp  = zmp_unc  + S[:, k_A] * (p_A * u)          S = Zu W            (N x N)
    M[i, j] = n_i^T G^-1 n_j = (p_i p_j + q_i q_j) * S[k_i, k_j]

so the solver needs only the q x q factor R (R^T R = M[A, A]) and never forms the
n x n matrix J nor the m x n constraint matrix.  The pivot rules, tolerances and
step formulas are those of eiquadprog.hpp:240-446; this file is used by the CPU
tests to check that formulation against the oracle, and documents the kernel.
"""
import numpy as np

EPS = np.finfo(np.float64).eps
OK, INFEASIBLE, ITER_CAP, OVERFLOW = 0, 1, 2, 3


def toeplitz_first_col(C, Ts, N):
    """First column of the lower-triangular Toeplitz input matrix of buildMatrix
    (MPCPlanner.cpp:133-153) and the N x 3 state matrix, by the same iteration."""
    A = np.array([[1, Ts, Ts ** 2 / 2], [0, 1, Ts], [0, 0, 1.0]])
    B = np.array([Ts ** 3 / 6, Ts ** 2 / 2, Ts])
    C = np.asarray(C, dtype=np.float64)
    Sx = np.zeros((N, 3))
    col = np.zeros(N)
    row = C @ A
    col[0] = C @ B
    for i in range(N):
        Sx[i] = row
        if i + 1 < N:
            col[i + 1] = row @ B
        row = row @ A
    return Sx, col


def lower_toeplitz(col):
    N = col.size
    M = np.zeros((N, N))
    for k in range(N):
        M += np.diag(np.full(N - k, col[k]), -k)
    return M


class Setup:
    """Instance-independent matrices for one (N, Ts, g, h, R, Q, Qs, variant, rw)."""

    def __init__(self, N, Ts, g, h, R, Q=1.0, Qs=100.0, variant=1, rw=1000, box=False):
        self.N = N
        self.Zx, zcol = toeplitz_first_col([1, 0, -h / g], Ts, N)
        self.Zu = lower_toeplitz(zcol)
        self.Xvx, vcol = toeplitz_first_col([0, 1, 0], Ts, N)
        self.Xvu = lower_toeplitz(vcol)
        G1 = R * np.eye(N)
        Gm = np.zeros((N, 4))           # g0 = Gm @ [x0; v_target] (+ zref term)
        if not box and variant >= 1:
            if rw == 1000:
                w = np.full(N, Q * N / N)
            else:
                i = np.arange(1, N + 1, dtype=np.float64)
                w = np.exp(-0.5 * ((i - rw) / rw) ** 2) / (np.sqrt(2 * np.pi) * rw) * Q * N
            G1 = G1 + self.Xvu.T @ (w[:, None] * self.Xvu)
            Gm[:, :3] = self.Xvu.T @ (w[:, None] * self.Xvx)
            Gm[:, 3] = -self.Xvu.T @ w
        self.Qs = 0.0
        if not box and variant == 2:
            G1 = G1 + Qs * self.Zu.T @ self.Zu
            Gm[:, :3] += Qs * self.Zu.T @ self.Zx
            self.Qs = Qs
        L = np.linalg.cholesky(G1)
        self.c1 = 2 * np.trace(G1) if not box else np.trace(G1)
        self.c2 = (2 if not box else 1) * np.sum(1.0 / np.diag(L))
        H1 = np.linalg.inv(G1)
        self.W = H1 @ self.Zu.T
        self.S = self.Zu @ self.W
        self.Gm = Gm
        self.Kx = -H1 @ Gm                           # jerk_unc = Kx @ w
        self.Kz = self.Zu @ self.Kx
        self.Kz[:, :3] += self.Zx                    # zmp_unc  = Kz @ w
        self.F = 0.5 * Gm.T @ self.Kx                # f_unc    = w^T F w

    def unconstrained(self, x0, v, zref=None):
        w = np.array([x0[0], x0[1], x0[2], v])
        jerk = self.Kx @ w
        zmp = self.Kz @ w
        g0 = self.Gm @ w
        if zref is not None and self.Qs != 0.0:
            g0 = g0 - self.Qs * (self.Zu.T @ zref)
            jerk = jerk + self.Qs * (self.W @ zref)
            zmp = zmp + self.Qs * (self.S @ zref)
        return jerk, zmp, 0.5 * g0 @ jerk


def solve(setup, knot, p, q, r, jx_unc, jy_unc, zx_unc, zy_unc, f_unc, iter_cap=0, qmax=10 ** 9,
          trace=None):
    """Dual active-set solve in ZMP space.  Returns dict(jerk_x, jerk_y, f, status,
    iters, active, u, s)."""
    S, W = setup.S, setup.W
    m = knot.size
    A = []          # working set (constraint indices, in the reference's order)
    u = []          # multipliers, u[len(A)] is the candidate's
    R = np.zeros((0, 0))
    R_norm = 1.0
    f = f_unc
    iters = 0
    passes = 0
    status = OK
    excl = np.ones(m, dtype=bool)
    s = None

    def mcol(j, rows):
        return (p[rows] * p[j] + q[rows] * q[j]) * S[knot[rows], knot[j]]

    def zmp_of(Aset, uu):
        zx, zy = zx_unc.copy(), zy_unc.copy()
        for a, ua in zip(Aset, uu):
            zx += (ua * p[a]) * S[:, knot[a]]
            zy += (ua * q[a]) * S[:, knot[a]]
        return zx, zy

    def drop(l):
        nonlocal R, A, u
        qq = A.index(l)
        iq = len(A)
        # eiquadprog.hpp:535-549 -- u has iq+1 entries, the last is the candidate's
        del A[qq]
        del u[qq]
        R = np.delete(R, qq, axis=1)
        for j in range(qq, iq - 1):          # :557-591 (rows of R only)
            cc, ss = R[j, j], R[j + 1, j]
            h = np.hypot(cc, ss)
            if h == 0.0:
                continue
            cc, ss = cc / h, ss / h
            rj = cc * R[j] + ss * R[j + 1]
            rj1 = -ss * R[j] + cc * R[j + 1]
            R[j], R[j + 1] = rj, rj1
            R[j + 1, j] = 0.0
        R = R[:iq - 1, :]

    while True:                                   # l1
        iters += 1
        if iter_cap and iters > iter_cap:
            status = ITER_CAP
            break
        zx, zy = zmp_of(A, u)
        s = p * zx[knot] + q * zy[knot] + r
        excl[:] = True
        psi = np.minimum(0.0, s).sum()
        if abs(psi) <= m * EPS * setup.c1 * setup.c2 * 100.0:
            break
        A_old, u_old = list(A), list(u)
        ss_ = 0.0
        ip = 0
        restart = True
        while restart:                            # l2
            restart = False
            inactive = np.ones(m, dtype=bool)
            inactive[A] = False
            for i in range(m):
                if s[i] < ss_ and inactive[i] and excl[i]:
                    ss_ = s[i]
                    ip = i
            if ss_ >= 0.0:
                status = OK
                return _finish(setup, knot, p, q, r, A, u, jx_unc, jy_unc, zx_unc, zy_unc, f,
                               status, iters)
            u = u[:len(A)] + [0.0]
            s_ip = s[ip]
            while True:                           # l2a
                passes += 1
                if iter_cap and passes > iter_cap:
                    return _finish(setup, knot, p, q, r, A, u, jx_unc, jy_unc, zx_unc, zy_unc, f,
                                   ITER_CAP, iters)
                iq = len(A)
                rows = np.array(A, dtype=np.int64)
                gvec = mcol(ip, rows) if iq else np.zeros(0)
                mii = (p[ip] ** 2 + q[ip] ** 2) * S[knot[ip], knot[ip]]
                if iq:
                    d1 = np.linalg.solve(R.T, gvec)
                    rr = np.linalg.solve(R, d1)
                else:
                    d1 = np.zeros(0)
                    rr = np.zeros(0)
                zn = mii - d1 @ d1
                # z = H (n_ip - N r): explicit, for the z != 0 test of :336
                zvx = p[ip] * W[:, knot[ip]]
                zvy = q[ip] * W[:, knot[ip]]
                for a, ra in zip(A, rr):
                    zvx = zvx - (ra * p[a]) * W[:, knot[a]]
                    zvy = zvy - (ra * q[a]) * W[:, knot[a]]
                zz = zvx @ zvx + zvy @ zvy
                if trace is not None:
                    trace.append((ip, iq, zn, mii, zz))
                t1, l = np.inf, 0
                for k in range(iq):
                    if rr[k] > 0.0 and u[k] / rr[k] < t1:
                        t1 = u[k] / rr[k]
                        l = A[k]
                t2 = -s_ip / zn if abs(zz) > EPS else np.inf
                t = min(t1, t2)
                if t >= np.inf:
                    status = INFEASIBLE
                    f = np.inf
                    return _finish(setup, knot, p, q, r, A, u, jx_unc, jy_unc, zx_unc, zy_unc, f,
                                   status, iters)
                if t2 >= np.inf:                  # dual step only
                    for k in range(iq):
                        u[k] -= t * rr[k]
                    u[iq] += t
                    drop(l)
                    continue
                f += t * zn * (0.5 * t + u[iq])
                for k in range(iq):
                    u[k] -= t * rr[k]
                u[iq] += t
                if t == t2:                       # full step: add
                    rho = np.sqrt(max(zn, 0.0))
                    Rn = np.zeros((iq + 1, iq + 1))
                    Rn[:iq, :iq] = R
                    Rn[:iq, iq] = d1
                    Rn[iq, iq] = rho
                    R = Rn
                    A.append(ip)
                    if len(A) > qmax:
                        status = OVERFLOW
                        return _finish(setup, knot, p, q, r, A, u, jx_unc, jy_unc, zx_unc,
                                       zy_unc, f, status, iters)
                    if abs(rho) <= EPS * R_norm:  # degenerate (:400-418)
                        excl[ip] = False
                        drop(ip)
                        nq = len(A)
                        A = A_old[:nq]
                        u = u_old[:nq]
                        restart = True
                        break
                    R_norm = max(R_norm, abs(rho))
                    break                         # goto l1
                s_ip += t * zn                    # partial step (:434-446)
                drop(l)
        # falls through to l1
    return _finish(setup, knot, p, q, r, A, u, jx_unc, jy_unc, zx_unc, zy_unc, f, status, iters)


def _finish(setup, knot, p, q, r, A, u, jx_unc, jy_unc, zx_unc, zy_unc, f, status, iters):
    jx, jy = jx_unc.copy(), jy_unc.copy()
    zx, zy = zx_unc.copy(), zy_unc.copy()
    if status in (OK, ITER_CAP):
        for a, ua in zip(A, u):
            jx += (ua * p[a]) * setup.W[:, knot[a]]
            jy += (ua * q[a]) * setup.W[:, knot[a]]
            zx += (ua * p[a]) * setup.S[:, knot[a]]
            zy += (ua * q[a]) * setup.S[:, knot[a]]
    else:
        jx[:] = 0.0
        jy[:] = 0.0
    return dict(jerk_x=jx, jerk_y=jy, f=f, status=status, iters=iters,
                active=np.array(A[:len(A)], dtype=np.int64), u=np.array(u[:len(A)]),
                s=p * zx[knot] + q * zy[knot] + r)


def rows_from_A(A, b, N):
    """(knot, p, q, r) of every row of the polygon matrix built by buildPolygonMatrix
    (MPCPlanner.cpp:1072-1079): p at column k, q at column k+N."""
    m = b.size
    knot = np.zeros(m, dtype=np.int64)
    p = np.zeros(m)
    q = np.zeros(m)
    for i in range(m):
        nz = np.nonzero(A[i])[0]
        k = nz[0] % N if nz.size else 0
        knot[i] = k
        p[i] = A[i, k]
        q[i] = A[i, k + N]
    return knot, p, q, b.copy()

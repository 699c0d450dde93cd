#!/usr/bin/env python3
"""Writes tests/golden/table_reference.npz: the instance-independent tables of a few parameter
sets computed with 60-digit decimal arithmetic straight from the model's formulas -- a third
construction, independent of the library's GPU (double-double) and host (long double) ones and
far more precise than either, so it can tell which of the two is off when they disagree.

    prediction  : A = [[1,T,T^2/2],[0,1,T],[0,0,1]], B = (T^3/6, T^2/2, T)', rows C A^(i+1),
                  Toeplitz sub-diagonals C A^k B                (MPCPlanner.cpp:22-70, 101-156)
    Hessian     : G1 = R I [+ Xvu' W Xvu] [+ Qs Zu' Zu]          (MPCPlanner.cpp:523-559, 723-739)
                  G1 = Zu' Zu + (R/Q) I for solveQP              (MPCPlanner.cpp:233)
    tables      : W = G1^-1 Zu', S = Zu W, Kx = -G1^-1 Gm, Kz = Zu Kx + Zx, J0 = L^-T

The inputs that are not exact doubles (height / gravity, Gaussian weights) are taken at the
extended precision the library itself starts from (numpy.longdouble = the C long double here).
Nothing of the reference is read or run: this is arithmetic on the formulas above.

    python tests/golden/make_table_reference.py            # rewrites the fixture
"""
import os
import sys
from decimal import Decimal, getcontext

import numpy as np

getcontext().prec = 60
D = Decimal
PI_LD = np.longdouble("3.141592653589793238462643383279502884")
HERE = os.path.dirname(os.path.abspath(__file__))

# (name, N, form, R, Q, Qs, rw, height)
CASES = [
    ("vel_R1e-2_N48", 48, "VEL", 0.01, 1.0, 100.0, 1000, 0.5),
    ("vel_R1e-6_N80", 80, "VEL", 1e-6, 1.0, 100.0, 1000, 0.5),
    ("ref_Qs100_N64", 64, "REF", 0.01, 1.0, 100.0, 1000, 0.5),
    ("ref_R1e-6_h0.43_N64", 64, "REF", 1e-6, 1.0, 100.0, 1000, 0.43),
    ("vel_gauss10_N48", 48, "VEL", 0.01, 1.0, 100.0, 10, 0.55),
    ("unc_R1e-6_N64", 64, "UNC", 1e-6, 1.0, 100.0, 1000, 0.5),
    ("jerk_N32", 32, "JERK", 0.01, 1.0, 100.0, 1000, 0.5),
]


def ld_to_decimal(x):
    """Exact value of a numpy.longdouble as a Decimal (two-double split is exact for 64 bits)."""
    x = np.longdouble(x)
    hi = float(x)
    lo = float(x - np.longdouble(hi))
    return D(hi) + D(lo)


def weights(N, Q, rw):
    if rw == 1000:
        return [D(Q)] * N
    ld = np.longdouble
    out = []
    for i in range(N):
        q = (ld(i + 1) - ld(rw)) / ld(rw)
        w = np.exp(ld(-0.5) * q * q) / (np.sqrt(ld(2) * PI_LD) * ld(rw)) * ld(Q) * ld(N)
        out.append(ld_to_decimal(w))
    return out


def prediction(C, T, N):
    """sx[i] = C A^(i+1) (3 entries), col[k] = C A^k B."""
    sx, col = [], []
    B = (T ** 3 / 6, T ** 2 / 2, T)
    for i in range(N):
        tau, s = T * (i + 1), T * i
        sx.append((C[0], C[0] * tau + C[1], C[0] * tau * tau / 2 + C[1] * tau + C[2]))
        ab = (B[0] + s * B[1] + s * s / 2 * B[2], B[1] + s * B[2], B[2])
        col.append(C[0] * ab[0] + C[1] * ab[1] + C[2] * ab[2])
    return sx, col


def toeplitz(col, N):
    return [[col[i - k] if k <= i else D(0) for k in range(N)] for i in range(N)]


def matmul(A, B):
    Bt = list(zip(*B))
    return [[sum(a * b for a, b in zip(row, colv)) for colv in Bt] for row in A]


def transpose(A):
    return [list(r) for r in zip(*A)]


def cholesky(G):
    N = len(G)
    L = [[D(0)] * N for _ in range(N)]
    for j in range(N):
        d = G[j][j] - sum(L[j][k] * L[j][k] for k in range(j))
        assert d > 0
        L[j][j] = d.sqrt()
        for i in range(j + 1, N):
            L[i][j] = (G[i][j] - sum(L[i][k] * L[j][k] for k in range(j))) / L[j][j]
    return L


def chol_solve(L, Bm):
    """(L L') X = B for a matrix B (list of rows)."""
    N, c = len(L), len(Bm[0])
    X = [list(r) for r in Bm]
    for i in range(N):
        for k in range(i):
            l = L[i][k]
            if l:
                X[i] = [x - l * y for x, y in zip(X[i], X[k])]
        X[i] = [x / L[i][i] for x in X[i]]
    for i in range(N - 1, -1, -1):
        for k in range(i + 1, N):
            l = L[k][i]
            if l:
                X[i] = [x - l * y for x, y in zip(X[i], X[k])]
        X[i] = [x / L[i][i] for x in X[i]]
    return X


def tables(N, form, R, Q, Qs, rw, height, Ts=0.1, gravity=9.81):
    T = D(Ts)
    c = ld_to_decimal(np.longdouble(height) / np.longdouble(gravity))
    zx, zc = prediction((D(1), D(0), -c), T, N)
    vx, vc = prediction((D(0), D(1), D(0)), T, N)
    Zu, Xvu = toeplitz(zc, N), toeplitz(vc, N)
    diag = ld_to_decimal(np.longdouble(R) / np.longdouble(Q)) if form == "UNC" else D(R)
    G = [[diag if i == j else D(0) for j in range(N)] for i in range(N)]
    Gm = [[D(0)] * 4 for _ in range(N)]

    def add(M, A, f=D(1)):
        for i in range(len(M)):
            M[i] = [m + f * a for m, a in zip(M[i], A[i])]

    if form in ("VEL", "REF"):
        w = weights(N, Q, rw)
        WX = [[w[i] * x for x in Xvu[i]] for i in range(N)]
        add(G, matmul(transpose(Xvu), WX))
        WXx = [[w[i] * vx[i][0], w[i] * vx[i][1], w[i] * vx[i][2], -w[i]] for i in range(N)]
        add(Gm, matmul(transpose(Xvu), WXx))
    if form in ("REF", "UNC"):
        f = D(Qs) if form == "REF" else D(1)
        add(G, matmul(transpose(Zu), Zu), f)
        Zx4 = [[zx[i][0], zx[i][1], zx[i][2], D(0)] for i in range(N)]
        add(Gm, matmul(transpose(Zu), Zx4), f)
    L = cholesky(G)
    W = chol_solve(L, transpose(Zu))
    S = matmul(Zu, W)
    Kx = [[-v for v in r] for r in chol_solve(L, Gm)]
    Kz = matmul(Zu, Kx)
    for i in range(N):
        Kz[i] = [Kz[i][a] + (zx[i][a] if a < 3 else D(0)) for a in range(4)]
    F = [[sum(Gm[i][a] * Kx[i][b] for i in range(N)) / 2 for b in range(4)] for a in range(4)]
    # J0 = L^-T: column c solves L' x = e_c
    J0 = [[D(0)] * N for _ in range(N)]
    for cidx in range(N):
        J0[cidx][cidx] = 1 / L[cidx][cidx]
        for i in range(cidx - 1, -1, -1):
            J0[i][cidx] = -sum(L[k][i] * J0[k][cidx] for k in range(i + 1, cidx + 1)) / L[i][i]
    f = lambda M: np.array([[float(v) for v in r] for r in M])
    return dict(S=f(S), Wt=f(transpose(W)), Kx=f(Kx), Kz=f(Kz), F=f(F), J0=f(J0),
                zu_col=np.array([float(v) for v in zc]), zx=f(zx),
                traceG1=float(sum(G[i][i] for i in range(N))), traceJ1=float(sum(1 / L[i][i] for i in range(N))))


def main():
    out = {}
    meta = []
    for name, N, form, R, Q, Qs, rw, h in CASES:
        t = tables(N, form, R, Q, Qs, rw, h)
        if name != CASES[1][0]:
            del t["J0"]  # (N x N like S and W: kept for the worst-conditioned case only)
        for k, v in t.items():
            out["%s/%s" % (name, k)] = np.asarray(v)
        meta.append("%s %d %s %r %r %r %d %r" % (name, N, form, R, Q, Qs, rw, h))
        print(name, "done", file=sys.stderr)
    out["cases"] = np.array(meta)
    np.savez_compressed(os.path.join(HERE, "table_reference.npz"), **out)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""numpy model of the chain-order pose solve (visual-inertial-odometry_amd/csrc/vio_pose_solve_chain.h).

The damped system (H_pp_schur + lambda I) dx = b (VM/src/backend/problem.cc:434-439) is factored WITHOUT pivoting in a static
order: the 11 speed-bias blocks (9 variables, ordered bg, ba, v) from both ends of their block-tridiagonal chain towards the
middle (0, 10, 1, 9, 2, 8, 3, 7, 4, 6, 5), then the camera block [pose 0 .. pose 10 | ext] in tiles of 16.  The tiles keep
L (u / d as a true quotient); updates are (L D) L^T.

  python tools/chain_solve_model.py            accuracy on tests/golden/ldlt.npz against the 50-digit solutions

ChainModel(H, b, lam) mirrors the kernel's data flow tile by tile (same operations, numpy's summation order inside a product),
so that tests/test_gpu_chain_solve.py can compare the kernel's LDS image (vio_debug_chain_solve) with it block by block.
"""
import os
import sys

import numpy as np

NS, TS, SCSZ, S9SZ = 11, 11, 176, 99
OFF_SC = 0
OFF_SO = OFF_SC + NS * 5 * SCSZ
OFF_SD = OFF_SO + 10 * S9SZ
OFF_CC = OFF_SD + NS * S9SZ
OFF_Y = OFF_CC + 15 * 272
YC = 176
PACKED = (OFF_Y + 256 + 1) & ~1
OFF_SM = PACKED
OFF_MC = OFF_SM + NS * S9SZ
OFF_D = OFF_MC + 272
OFF_X = OFF_D + 256
LDS_CORE = OFF_X + 256 + S9SZ + 272


def sc(e, t):
    return OFF_SC + (e * 5 + t) * SCSZ


def so(e):
    return OFF_SO + (e if e < 5 else e - 1) * S9SZ


def sd(e):
    return OFF_SD + e * S9SZ


def sm(e):
    return OFF_SM + e * S9SZ


def cc(I, J):
    return OFF_CC + (I * (I + 1) // 2 + J) * 272


def dim(i):
    """natural index of H_pp_schur_ -> chain dimension (speed-bias block e, local k: 16 e + k; camera variable c: 176 + c)"""
    if i < 6:
        return YC + 66 + i
    f, r = divmod(i - 6, 15)
    if r < 6:
        return YC + 6 * f + r
    c = r - 6
    return f * 16 + (6 + c if c < 3 else (c if c < 6 else c - 6))


def succ(e):
    return e + 1 if e < 5 else e - 1


def rdiv(a, d):
    """a / d, 0 where d == 0 (d_fast_rcp(0) == 0)"""
    d = np.asarray(d, dtype=np.float64)
    return np.where(d != 0, a / np.where(d == 0, 1.0, d), 0.0)


def factor(T):
    """unpivoted LDL^T of a small block; returns M = L^-T (unit upper triangular) and the pivots"""
    T = T.copy()
    k = T.shape[0]
    L = np.eye(k)
    d = np.zeros(k)
    for j in range(k):
        d[j] = T[j, j]
        L[j + 1:, j] = rdiv(T[j + 1:, j], d[j])
        T[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], T[j, j + 1:])
    M = np.eye(k)
    # the kernel forms M by running the identity's rows through the same column operations
    for j in range(k):
        for r in range(k):
            M[r, j + 1:] -= M[r, j] * L[j + 1:, j]
    return M, d, L


class ChainModel:
    def __init__(self, H, b, lam):
        n = 171
        A = np.array(H, dtype=np.float64) + lam * np.eye(n)
        self.dims = np.array([dim(i) for i in range(n)])
        SD = np.zeros((NS, 9, 9)); SO = np.zeros((NS, 9, 9)); SC = np.zeros((NS, 80, 9)); CC = np.zeros((80, 80))
        yS = np.zeros((NS, 9)); yC = np.zeros(80)
        for i in range(72, 80):
            CC[i, i] = 1.0
        for i in range(n):
            di = self.dims[i]
            if di < YC:
                yS[di >> 4, di & 15] = b[i]
            else:
                yC[di - YC] = b[i]
            for j in range(n):
                dj = self.dims[j]
                v = A[i, j]
                if di >= YC and dj >= YC:
                    CC[di - YC, dj - YC] = v
                elif di >= YC:
                    SC[dj >> 4, di - YC, dj & 15] = v
                elif dj < YC:
                    f, g = di >> 4, dj >> 4
                    if f == g:
                        SD[f, di & 15, dj & 15] = v
                    elif g != 5 and succ(g) == f and abs(f - g) == 1:
                        SO[g, di & 15, dj & 15] = v           # rows succ(g), columns g
                    elif f != 5 and succ(f) == g and abs(f - g) == 1:
                        pass
                    elif v != 0.0:
                        raise ValueError("entry (%d, %d) couples speed-bias blocks %d and %d: outside the chain pattern" % (i, j, f, g))
        M = {}; D = {}
        for lev in range(6):
            for e in ([lev, 10 - lev] if lev < 5 else [5]):
                M[e], D[e], _ = factor(SD[e])
                SD[e] = np.nan                                   # (the kernel leaves U there; not compared)
                SC[e] = rdiv(SC[e] @ M[e], D[e])                 # L = (A M) / d
                yS[e] = M[e].T @ yS[e]                           # w_e = L_ee^-1 y_e
                if e != 5:
                    nx = succ(e)
                    SO[e] = rdiv(SO[e] @ M[e], D[e])
                    SD[nx] -= (SO[e] * D[e]) @ SO[e].T
                    SC[nx] -= (SC[e] * D[e]) @ SO[e].T
                    yS[nx] -= SO[e] @ yS[e]
                CC -= (SC[e] * D[e]) @ SC[e].T
                yC -= SC[e] @ yS[e]
        Mc = {}; Dc = np.zeros(80)
        for K in range(5):
            s, t = 16 * K, 16 * K + 16
            nreal = 16 if K < 4 else 8
            m, d, _ = factor(CC[s:s + nreal, s:s + nreal])
            Mk = np.eye(16); Mk[:nreal, :nreal] = m
            dk = np.ones(16); dk[:nreal] = d
            Mc[K] = Mk; Dc[s:t] = dk
            yC[s:t] = Mk.T @ yC[s:t]
            if t < 80:
                Lr = rdiv(CC[t:, s:t] @ Mk, dk)
                CC[t:, t:] -= (Lr * dk) @ Lr.T
                yC[t:] -= Lr @ yC[s:t]
                CC[t:, s:t] = Lr
        xC = np.zeros(80)
        for K in range(4, -1, -1):
            s, t = 16 * K, 16 * K + 16
            xC[s:t] = Mc[K] @ (rdiv(yC[s:t], Dc[s:t]) - CC[t:, s:t].T @ xC[t:])
        xS = np.zeros((NS, 9))
        for lev in range(5, -1, -1):
            for e in ([lev, 10 - lev] if lev < 5 else [5]):
                acc = SC[e].T @ xC
                if e != 5:
                    acc = acc + SO[e].T @ xS[succ(e)]
                xS[e] = M[e] @ (rdiv(yS[e], D[e]) - acc)
        self.SC, self.SO, self.CC, self.M, self.D, self.Mc, self.Dc = SC, SO, CC, M, D, Mc, Dc
        self.wS, self.wC, self.xS, self.xC = yS, yC, xS, xC
        x = np.zeros(n)
        for i in range(n):
            di = self.dims[i]
            x[i] = xS[di >> 4, di & 15] if di < YC else xC[di - YC]
        self.x = x

    def compare_dump(self, dump):
        """largest scaled difference between the kernel's LDS image and this model, per kind of block"""
        dump = np.asarray(dump)
        out = {}

        def rel(a, b):
            s = max(np.abs(b).max(), 1e-300)
            return float(np.abs(a - b).max() / s)
        w = []
        for e in range(NS):
            for t in range(5):
                tile = dump[sc(e, t):sc(e, t) + SCSZ].reshape(16, TS)[:, :9]
                w.append(rel(tile, self.SC[e, 16 * t:16 * t + 16]))
        out["L_SC"] = max(w)
        # (the kernel's back-substitution leaves G_e = M_e L_SO[e]^T in the place of L_SO[e]: x_e = M_e v_e - G_e x_succ(e))
        out["G_SO"] = max(rel(dump[so(e):so(e) + S9SZ].reshape(9, TS)[:, :9], self.M[e] @ self.SO[e].T) for e in range(NS) if e != 5)
        out["M_S"] = max(rel(dump[sm(e):sm(e) + S9SZ].reshape(9, TS)[:, :9], self.M[e]) for e in range(NS))
        out["D_S"] = max(rel(dump[OFF_D + 16 * e:OFF_D + 16 * e + 9], self.D[e]) for e in range(NS))
        out["D_C"] = rel(dump[OFF_D + YC:OFF_D + YC + 80], self.Dc)
        w = []
        for I in range(5):
            for J in range(I):
                tile = dump[cc(I, J):cc(I, J) + 272].reshape(16, 17)[:, :16]
                w.append(rel(tile, self.CC[16 * I:16 * I + 16, 16 * J:16 * J + 16]))
        out["L_CC"] = max(w)
        w = []
        for K in range(5):
            tile = dump[cc(K, K):cc(K, K) + 272].reshape(16, 17)[:, :16].copy()
            np.fill_diagonal(tile, 1.0)                           # (the pivots stay on the diagonal; M's own diagonal is 1)
            w.append(rel(np.triu(tile), np.triu(self.Mc[K])))
        out["M_C"] = max(w)
        out["w_S"] = max(rel(dump[OFF_Y + 16 * e:OFF_Y + 16 * e + 9], self.wS[e]) for e in range(NS))
        out["w_C"] = rel(dump[OFF_Y + YC:OFF_Y + YC + 80], self.wC)
        out["x_S"] = max(rel(dump[OFF_X + 16 * e:OFF_X + 16 * e + 9], self.xS[e]) for e in range(NS))
        out["x_C"] = rel(dump[OFF_X + YC:OFF_X + YC + 80], self.xC)
        return out


def chain_pattern_system(rng, scale_bias=1e16, with_prior=True):
    """a random symmetric positive definite 171 x 171 system with the sparsity and the scaling of a window's reduced system:
    dense camera block, IMU-like 30 x 30 blocks between neighbouring frames (bias random walk of weight scale_bias), optionally a
    prior that couples speed-bias block 0 with everything in the camera block"""
    n = 171
    H = np.zeros((n, n))
    cam = [i for i in range(n) if dim(i) >= YC]
    J = rng.standard_normal((200, len(cam))) * 30
    H[np.ix_(cam, cam)] += J.T @ J
    for k in range(10):
        idx = list(range(6 + 15 * k, 6 + 15 * k + 30))
        Jk = rng.standard_normal((15, 30)) * 100
        H[np.ix_(idx, idx)] += Jk.T @ Jk
        for c in range(9, 15):                                   # ba, bg random walk between frames k and k + 1
            a, b2 = 6 + 15 * k + c, 6 + 15 * (k + 1) + c
            wgt = scale_bias * (1.0 if c >= 12 else 1e-2)
            H[a, a] += wgt; H[b2, b2] += wgt; H[a, b2] -= wgt; H[b2, a] -= wgt
    if with_prior:
        idx = cam + list(range(12, 21))
        Jp = rng.standard_normal((40, len(idx))) * 50
        H[np.ix_(idx, idx)] += Jp.T @ Jp
    b = rng.standard_normal(n) * 1e3
    return H, b


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = np.load(os.path.join(root, "tests", "golden", "ldlt.npz"))
    e = np.load(os.path.join(root, "tests", "golden", "ldlt_exact.npz"))
    for k in range(3):
        lam = float(d["lambda_%d" % k])
        m = ChainModel(d["Hs"], d["bs"], lam)
        xe, xr = e["x_exact_%d" % k], d["x_%d" % k]
        print("lambda = %-8.3g  chain order: %.2e from exact   Eigen LDLT: %.2e from exact   chain - Eigen: %.2e"
              % (lam, np.abs(m.x - xe).max(), np.abs(xr - xe).max(), np.abs(m.x - xr).max()))

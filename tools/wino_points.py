#!/usr/bin/env python3
"""Which interpolation points should the F(4x4,3x3) transforms use?  (the Winograd plan is the less accurate plan.)

Builds the Toom-Cook matrices A^T (4 x 6), G (6 x 3), B^T (6 x 6) for a set of five finite points + infinity in exact rational
arithmetic, checks the bilinear identity  y = A^T [(G g) . (B^T d)]  == valid correlation, and measures on the CPU how far an fp32
evaluation of a 2-D layer (C input channels accumulated in fp32, as the batched GEMM does) lands from the fp64 convolution:
forward, data gradient (same transforms, flipped filter) and the Winograd-form weight gradient (adjoint transforms).

    python tools/wino_points.py            # table for the candidate point sets
"""
import itertools
import sys
from fractions import Fraction as Fr

import numpy as np


def matrices(points, m=4, r=3):
    """wincnn-style construction, fractions in G: returns (AT, G, BT) as lists of Fractions."""
    a = [Fr(p) for p in points]
    alpha = m + r - 1
    assert len(a) == alpha - 1
    # A^T[i][j] = a_j^i, last column = delta(i, m - 1)
    AT = [[(a[j] ** i if j < alpha - 1 else Fr(int(i == m - 1))) for j in range(alpha)] for i in range(m)]
    # G[j][k] = a_j^k / N_j, N_j = prod_{l != j} (a_j - a_l); last row = delta(k, r - 1)
    G = []
    for j in range(alpha - 1):
        N = Fr(1)
        for l in range(alpha - 1):
            if l != j:
                N *= (a[j] - a[l])
        G.append([a[j] ** k / N for k in range(r)])
    G.append([Fr(int(k == r - 1)) for k in range(r)])
    # B^T: solve  sum_j AT[i][j] G[j][k] BT[j][l] = delta(l, i + k)  for every l (exact Gaussian elimination on the
    # (m r) x alpha system; consistent by construction)
    rows = [(i, k) for i in range(m) for k in range(r)]
    Msys = [[AT[i][j] * G[j][k] for j in range(alpha)] for (i, k) in rows]
    BT = [[Fr(0)] * alpha for _ in range(alpha)]
    for l in range(alpha):
        rhs = [Fr(int(l == i + k)) for (i, k) in rows]
        sol = solve(Msys, rhs, alpha)
        for j in range(alpha):
            BT[j][l] = sol[j]
    return AT, G, BT


def solve(Mx, rhs, n):
    A = [list(row) + [b] for row, b in zip(Mx, rhs)]
    piv_cols, r = [], 0
    for c in range(n):
        p = next((i for i in range(r, len(A)) if A[i][c] != 0), None)
        if p is None:
            continue
        A[r], A[p] = A[p], A[r]
        inv = 1 / A[r][c]
        A[r] = [v * inv for v in A[r]]
        for i in range(len(A)):
            if i != r and A[i][c] != 0:
                f = A[i][c]
                A[i] = [vi - f * vr for vi, vr in zip(A[i], A[r])]
        piv_cols.append(c)
        r += 1
    assert all(all(v == 0 for v in row[:-1]) and row[-1] == 0 for row in A[r:]), 'inconsistent system'
    sol = [Fr(0)] * n
    for i, c in enumerate(piv_cols):
        sol[c] = A[i][-1]
    return sol


def rescale(AT, G, BT):
    """Move the denominators: scale row j of G by s_j and row j of B^T by 1 / s_j so that B^T is integral / small (exact powers of
    two preferred).  Here: make every row of B^T have integer entries with gcd 1 (the usual published form)."""
    from math import gcd
    alpha = len(G)
    for j in range(alpha):
        den = 1
        for v in BT[j]:
            den = den * v.denominator // gcd(den, v.denominator)
        num = 0
        for v in BT[j]:
            num = gcd(num, abs(int(v * den)))
        s = Fr(den, max(num, 1))
        BT[j] = [v * s for v in BT[j]]
        G[j] = [v / s for v in G[j]]
    return AT, G, BT


def f32(M):
    return np.array([[float(v) for v in row] for row in M], dtype=np.float32)


def experiment(points, C=256, T=64, K=8, seed=0):
    AT, G, BT = rescale(*matrices(points))
    at, g, bt = f32(AT), f32(G), f32(BT)
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((T, C, 6, 6)).astype(np.float32)
    w = (rng.standard_normal((K, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    # fp32 pipeline
    V = np.einsum('ij,tcjk,lk->tcil', bt, d, bt, optimize=True).astype(np.float32)
    U = np.einsum('ij,kcjl,ml->kcim', g, w, g, optimize=True).astype(np.float32)
    M = np.zeros((T, K, 6, 6), np.float32)
    for c0 in range(0, C, 32):                 # fp32 accumulation over channels, chunked like the GEMM's K loop
        M += np.einsum('kcim,tcim->tkim', U[:, c0:c0 + 32], V[:, c0:c0 + 32], optimize=True).astype(np.float32)
    Y = np.einsum('ij,tkjl,ml->tkim', at, M, at, optimize=True).astype(np.float32)
    # fp64 direct
    d64, w64 = d.astype(np.float64), w.astype(np.float64)
    Yr = np.zeros((T, K, 4, 4))
    for i in range(4):
        for j in range(4):
            Yr[:, :, i, j] = np.einsum('tcrs,kcrs->tk', d64[:, :, i:i + 3, j:j + 3], w64)
    e_fwd = np.abs(Y - Yr).max() / np.abs(Yr).max()
    r_fwd = np.sqrt(((Y - Yr) ** 2).mean()) / np.sqrt((Yr ** 2).mean())
    # Winograd-form weight gradient: dW = G^T [ sum_t (B^T d B) . (A dY A^T) ] G   (adjoint transforms)
    dy = rng.standard_normal((T, K, 4, 4)).astype(np.float32)
    Z = np.einsum('ji,tkjl,lm->tkim', at, dy, at, optimize=True).astype(np.float32)          # A dY A^T : (6 x 4)(4 x 4)(4 x 6)
    dU = np.einsum('tkim,tcim->kcim', Z, V, optimize=True).astype(np.float32)
    dW = np.einsum('ji,kcjl,lm->kcim', g, dU, g, optimize=True).astype(np.float32)           # G^T dU G
    dWr = np.zeros((K, C, 3, 3))
    dy64 = dy.astype(np.float64)
    for rr in range(3):
        for ss in range(3):
            dWr[:, :, rr, ss] = np.einsum('tkij,tcij->kc', dy64, d64[:, :, rr:rr + 4, ss:ss + 4])
    e_wg = np.abs(dW - dWr).max() / np.abs(dWr).max()
    r_wg = np.sqrt(((dW - dWr) ** 2).mean()) / np.sqrt((dWr ** 2).mean())
    return dict(points=points, fwd_max=e_fwd, fwd_rms=r_fwd, wg_max=e_wg, wg_rms=r_wg,
                bt_max=float(np.abs(bt).max()), at_max=float(np.abs(at).max()), g_max=float(np.abs(g).max()), mats=(AT, G, BT))


def show(M, name):
    print(name + ' = ' + '; '.join('[' + ', '.join(str(v) for v in row) + ']' for row in M))


if __name__ == '__main__':
    half = Fr(1, 2)
    cands = [(0, 1, -1, 2, -2), (0, 1, -1, half, -half), (0, 1, -1, 2, -half), (0, 1, -1, half, -2), (0, half, -half, 2, -2),
             (0, 1, -1, Fr(3, 2), -Fr(3, 2)), (0, 1, -1, Fr(1, 2), 2), (0, Fr(1, 2), -Fr(1, 2), 1, -2), (0, 1, -1, Fr(3, 4), -Fr(3, 4)),
             (0, half, -half, Fr(3, 2), -Fr(3, 2))]
    out = []
    for pts in cands:
        acc = [experiment(pts, seed=s) for s in range(3)]
        e = {k: float(np.mean([a[k] for a in acc])) for k in ('fwd_max', 'fwd_rms', 'wg_max', 'wg_rms')}
        out.append((pts, e, acc[0]))
        print('%-28s fwd max %.2e rms %.2e | wgrad max %.2e rms %.2e | max|BT| %.3g max|AT| %.3g max|G| %.3g' % (
            str(tuple(str(p) for p in pts)), e['fwd_max'], e['fwd_rms'], e['wg_max'], e['wg_rms'], acc[0]['bt_max'], acc[0]['at_max'], acc[0]['g_max']))
    if '--mats' in sys.argv:
        best = min(out, key=lambda t: t[1]['fwd_rms'])
        print('\nbest (forward rms):', best[0])
        AT, G, BT = best[2]['mats']
        show(BT, 'BT'); show(G, 'G'); show(AT, 'AT')

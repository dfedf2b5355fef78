#!/usr/bin/env python
"""Derivation and numerics of the F(6,3)/F(4,3) Winograd tiling of csrc/wino63_kernels.hip (no GPU needed).
  * Cook-Toom matrices A^T, G, B^T of F(6,3) on the points {0, 1, -1, 2, -2, 1/2, -1/2, inf} and of F(4,3) on the subset
    {0, 1, -1, 2, -2, inf}, as exact fractions (the constants of w63_bt / w63_at / w63_g);
  * rho with G4_j = rho_j * G8_j on the shared points: the factor folded into the F(4,3) tiles' data transform so that ONE set of
    64 transformed filters serves every tile;
  * the 14 = 6 + 4 + 4 tiling against a direct convolution: exact in float64 (~1e-13), and its float32 error (transforms in fp32,
    products accumulated in fp32 in steps of two like v_mfma_f32_32x32x2_f32) beside the uniform F(4,3) tiling's.
      python tools/wino63_numerics.py"""
import numpy as np
from fractions import Fraction as Fr
def cook_toom(m, r, pts):
    """pts: finite points (n-1 of them) ; returns AT (m x n), G (n x r), BT (n x n) as Fractions, n = m + r - 1, last point = inf.
    Y = AT [(G g) .* (BT d)],  y_i = sum_q g_q d_{i+q}"""
    n = m + r - 1
    assert len(pts) == n - 1
    a = [Fr(p) for p in pts]
    AT = [[a[j] ** i for j in range(n - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    G = []
    for j in range(n - 1):
        N = Fr(1)
        for k in range(n - 1):
            if k != j: N *= (a[j] - a[k])
        G.append([a[j] ** k / N for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    # BT rows: finite j: coefficients of M_j(x) = prod_{k != j} (x - a_k) (degree n-2), inf row: M(x) = prod_k (x - a_k) (degree n-1)
    def polymul(p, q):
        out = [Fr(0)] * (len(p) + len(q) - 1)
        for i, x in enumerate(p):
            for j, y in enumerate(q): out[i + j] += x * y
        return out
    BT = []
    for j in range(n - 1):
        p = [Fr(1)]
        for k in range(n - 1):
            if k != j: p = polymul(p, [-a[k], Fr(1)])
        BT.append(p + [Fr(0)] * (n - len(p)))
    p = [Fr(1)]
    for k in range(n - 1): p = polymul(p, [-a[k], Fr(1)])
    BT.append(p)
    return AT, G, BT
def check(m, r, pts):
    AT, G, BT = cook_toom(m, r, pts)
    n = m + r - 1
    f = lambda M: np.array([[float(x) for x in row] for row in M])
    A, Gm, B = f(AT), f(G), f(BT)
    rng = np.random.default_rng(0)
    d = rng.standard_normal(n); g = rng.standard_normal(r)
    y = A @ ((Gm @ g) * (B @ d))
    ref = np.array([sum(g[q] * d[i + q] for q in range(r)) for i in range(m)])
    return np.abs(y - ref).max(), AT, G, BT
P8 = [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2)]
P6 = [0, 1, -1, 2, -2]
e8, AT8, G8, BT8 = check(6, 3, P8)
e6, AT6, G6, BT6 = check(4, 3, P6)
print("err", e8, e6)
pr = lambda name, M: print(name, [[str(x) for x in row] for row in M])
pr("AT8", AT8); pr("G8", G8); pr("BT8", BT8)
pr("AT6", AT6); pr("G6", G6); pr("BT6", BT6)
S6 = [0, 1, 2, 3, 4, 7]
rho = []
for jj, j in enumerate(S6):
    ratios = set()
    for k in range(3):
        if G8[j][k] != 0: ratios.add(G6[jj][k] / G8[j][k])
    rho.append(ratios)
print("rho (G6_j = rho_j * G8_j):", [[str(x) for x in r] for r in rho])

# ---------------- fp32 error of the mixed tilings on a 14x14 map, Cin = Cout = 256 ----------------
f32 = np.float32
def mat(M): return np.array([[float(x) for x in row] for row in M], dtype=np.float64)
A8, Gm8, B8 = mat(AT8), mat(G8), mat(BT8)
A6, Gm6, B6 = mat(AT6), mat(G6), mat(BT6)
rho6 = np.array([-0.25, 0.75, 0.75, 3.75, 3.75, 1.0])
def conv_wino(x, w, sizes, fp):
    """x [14,14,Ci] (zero padded SAME), w [3,3,Ci,Co]; sizes per dim e.g. (6,4,4) or (4,4,4,2); transforms and accumulation in dtype fp"""
    H = x.shape[0]; Ci, Co = w.shape[2], w.shape[3]
    xp = np.zeros((H + 2 + 8, H + 2 + 8, Ci), fp); xp[1:H + 1, 1:H + 1] = x
    # 8-point filters (always): U8[i,j] = G8 w G8^T
    U8 = np.einsum("ia,abcd,jb->ijcd", Gm8.astype(fp), w.astype(fp), Gm8.astype(fp)).astype(fp)
    S6 = [0, 1, 2, 3, 4, 7]
    y = np.zeros((H, H, Co), fp)
    starts = np.cumsum([0] + list(sizes))[:-1]
    def dirmats(m):
        if m == 6: return A8.astype(fp), B8.astype(fp), list(range(8))
        if m == 4: return A6.astype(fp), (rho6[:, None] * B6).astype(fp), S6
        raise ValueError
    for ty, (sy, my) in enumerate(zip(starts, sizes)):
        Ay, By, Py = dirmats(my)
        for tx, (sx, mx) in enumerate(zip(starts, sizes)):
            Ax, Bx, Px = dirmats(mx)
            d = xp[sy:sy + my + 2, sx:sx + mx + 2]                      # patch (my+2) x (mx+2) x Ci
            V = np.einsum("ia,abc->ibc", By, d).astype(fp)
            V = np.einsum("jb,ibc->ijc", Bx, V).astype(fp)             # |Py| x |Px| x Ci
            U = U8[np.ix_(Py, Px)]                                      # |Py| x |Px| x Ci x Co
            M = np.zeros((len(Py), len(Px), Co), fp)
            for k in range(0, Ci, 2):                                   # fp32 accumulation in steps of 2 (MFMA 32x32x2)
                M = (M + np.einsum("ijc,ijcd->ijd", V[:, :, k:k + 2].astype(np.float64), U[:, :, k:k + 2].astype(np.float64))).astype(fp)
            o = np.einsum("ai,ijd->ajd", Ay, M).astype(fp)
            o = np.einsum("bj,ajd->abd", Ax, o).astype(fp)
            yy, xx = min(my, H - sy), min(mx, H - sx)
            y[sy:sy + yy, sx:sx + xx] = o[:yy, :xx]
    return y
rng = np.random.default_rng(3)
Ci = Co = 256
for trial in range(2):
    x = np.maximum(rng.standard_normal((14, 14, Ci)), 0).astype(f32) * (1.0 if trial == 0 else 3.0)       # post-ReLU activations
    w = (rng.standard_normal((3, 3, Ci, Co)) * 0.03).astype(f32)
    ref = conv_wino(x.astype(np.float64), w.astype(np.float64), (6, 4, 4), np.float64)
    # exact check of the algorithm itself in fp64 against a direct convolution
    xp = np.zeros((16, 16, Ci)); xp[1:15, 1:15] = x
    direct = sum(np.einsum("yxc,cd->yxd", xp[ky:ky + 14, kx:kx + 14], w[ky, kx].astype(np.float64)) for ky in range(3) for kx in range(3))
    print("fp64 mixed (6,4,4) vs direct: %.2e" % np.abs(ref - direct).max())
    sc = np.abs(direct).max()
    # current tiling: F(4,3) everywhere (the F(2,3) tiles are better conditioned still) ~ sizes (4,4,4,4) on a padded map
    for name, sizes in (("F(6,3)+F(4,3)+F(4,3)  [400 point-tiles]", (6, 4, 4)),):
        y = conv_wino(x, w, sizes, f32)
        e = np.abs(y - direct)
        print("trial %d %-40s max %.3e rms %.3e (relative to max|y| = %.2f)" % (trial, name, e.max() / sc, np.sqrt((e ** 2).mean()) / sc, sc))
    # F(4,3) uniform for comparison: emulate with sizes (4,4,4,4): last tile partially outside (zero padded input, outputs cropped)
    def conv_f43(x, w, fp):
        H = 14
        xp = np.zeros((H + 2 + 8, H + 2 + 8, Ci), fp); xp[1:H + 1, 1:H + 1] = x
        U = np.einsum("ia,abcd,jb->ijcd", Gm6.astype(fp), w.astype(fp), Gm6.astype(fp)).astype(fp)
        y = np.zeros((16, 16, Co), fp)
        for sy in range(0, 16, 4):
            for sx in range(0, 16, 4):
                d = xp[sy:sy + 6, sx:sx + 6]
                V = np.einsum("ia,abc->ibc", B6.astype(fp), d).astype(fp); V = np.einsum("jb,ibc->ijc", B6.astype(fp), V).astype(fp)
                M = np.zeros((6, 6, Co), fp)
                for k in range(0, Ci, 2):
                    M = (M + np.einsum("ijc,ijcd->ijd", V[:, :, k:k + 2].astype(np.float64), U[:, :, k:k + 2].astype(np.float64))).astype(fp)
                o = np.einsum("ai,ijd->ajd", A6.astype(fp), M).astype(fp); o = np.einsum("bj,ajd->abd", A6.astype(fp), o).astype(fp)
                y[sy:sy + 4, sx:sx + 4] = o
        return y[:14, :14]
    y = conv_f43(x, w, f32); e = np.abs(y - direct)
    print("trial %d %-40s max %.3e rms %.3e" % (trial, "F(4,3) uniform            [576]", e.max() / sc, np.sqrt((e ** 2).mean()) / sc))

"""The minimal-filtering identities csrc/wino.hip and csrc/wino_wgrad.hip implement, checked in float64 numpy (no GPU):
F(2x2, 3x3)  Y = A^T [(G g G^T) . (B^T d B)] A   (forward / input gradient), its 5x5 form as 2 x 2 zero-padded sub-filters 3 pixels
apart, and the transposition F(3x3, 2x2)  g = A_w^T [(G_w p G_w^T) . (B_w^T q B_w)] A_w  (weight gradient) -- with the very
matrices (and the split of the transform rows over the two waves of a pair) that the kernels hard-code."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)     # wino.hip: bt4 (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)               # wino_filter_kernel
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)                                   # wino_finish
BTW = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, -1, 0, 1]], float)    # wino_wgrad.hip (last row: d3 - d1)
GW = np.array([[1, 0], [.5, .5], [.5, -.5], [0, 1]], float)
ATW = np.array([[1, 1, 1, 0], [0, 1, -1, 0], [0, 1, 1, 1]], float)


def corr(d, g):
    oh, ow = d.shape[0] - g.shape[0] + 1, d.shape[1] - g.shape[1] + 1
    return np.array([[(d[a:a + g.shape[0], b:b + g.shape[1]] * g).sum() for b in range(ow)] for a in range(oh)])


def test_f2x2_3x3_and_the_row_split_of_the_output_transform():
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal((4, 4)), rng.standard_normal((3, 3))
    m = (G @ g @ G.T) * (BT @ d @ BT.T)
    assert np.allclose(AT @ m @ AT.T, corr(d, g), atol=1e-12)
    # wave ph = 0 holds transform rows 0, 1 and ph = 1 rows 2, 3; each forms its part of both output rows (wino_finish)
    s = [np.stack([m[0] + m[1], m[1]]), np.stack([m[2], -m[2] - m[3]])]
    parts = [np.stack([si[:, 0] + si[:, 1] + si[:, 2], si[:, 1] - si[:, 2] - si[:, 3]], axis=1) for si in s]
    assert np.allclose(parts[0] + parts[1], corr(d, g), atol=1e-12)


def test_5x5_as_four_subfilters_of_3x3():
    rng = np.random.default_rng(1)
    # 2x2 outputs of a 5x5 correlation need a 6x6 patch; the displaced sub-filters' 4x4 windows reach one row / column further,
    # where the zero-padded taps sit (whatever the kernel reads there is multiplied by zero)
    d, g = rng.standard_normal((7, 7)), rng.standard_normal((5, 5))
    gp = np.zeros((6, 6)); gp[:5, :5] = g
    y = np.zeros((2, 2))
    for sa in range(2):
        for sb in range(2):
            sub = gp[3 * sa:3 * sa + 3, 3 * sb:3 * sb + 3]
            patch = d[3 * sa:3 * sa + 4, 3 * sb:3 * sb + 4]              # the same 4x4 data transform, 3 pixels further down / right
            y += AT @ ((G @ sub @ G.T) * (BT @ patch @ BT.T)) @ AT.T
    assert np.allclose(y, corr(d[:6, :6], g), atol=1e-12)


def test_f3x3_2x2_weight_gradient_with_unscaled_accumulation():
    rng = np.random.default_rng(2)
    taps = np.zeros((3, 3)); u_unscaled = np.zeros((4, 4))
    ref = np.zeros((3, 3))
    for _ in range(7):                                                   # tiles accumulate in the transform domain
        q, p = rng.standard_normal((4, 4)), rng.standard_normal((2, 2))  # 4x4 patch of x, 2x2 tile of dy
        ref += corr(q, p)
        taps += ATW @ ((GW @ p @ GW.T) * (BTW @ q @ BTW.T)) @ ATW.T
        # the kernel leaves G's halves out of the loop: rows / columns 1, 2 carry a factor 2 each until the output transform
        z = (2 * GW[:, :] * np.array([[.5], [1], [1], [.5]]) ) @ p @ (2 * GW * np.array([[.5], [1], [1], [.5]])).T
        u_unscaled += z * (BTW @ q @ BTW.T)
    assert np.allclose(taps, ref, atol=1e-12)
    f = np.array([1, .5, .5, 1])
    assert np.allclose(ATW @ (u_unscaled * np.outer(f, f)) @ ATW.T, ref, atol=1e-12)


# ---- csrc/wino4.hip: F(4x4, 3x3) with the matrices the kernel hard-codes (wino4_xform.hpp: bt6_cols / bt6_row, at6_pk, wino4_filter_kernel) and
# the forms it takes.  Interpolation points 0, +-a, +-b, infinity with a = 3/4, b = 5/4 (round 5; rounds 3-4: a = 1, b = 2).
WA, WB = 0.75, 1.25
WA2, WB2, WA2B2, WS2 = WA * WA, WB * WB, WA * WA * WB * WB, WA * WA + WB * WB
WN0, WNA, WNB = WA2B2, 2 * WA2 * (WA2 - WB2), 2 * WB2 * (WB2 - WA2)          # N_j = prod_{l != j} (p_j - p_l) over the finite points


def w4_matrices(a, b):
    """unnormalised Toom-Cook matrices of F(4, 3) / F(3, 4) at the points 0, a, -a, b, -b, infinity (the filter side carries 1 / N_j)"""
    a2, b2 = a * a, b * b
    bt = np.array([[a2 * b2, 0, -(a2 + b2), 0, 1, 0], [0, -a * b2, -b2, a, 1, 0], [0, a * b2, -b2, -a, 1, 0], [0, -a2 * b, -a2, b, 1, 0],
                   [0, a2 * b, -a2, -b, 1, 0], [0, a2 * b2, 0, -(a2 + b2), 0, 1]], float)
    n0, na, nb = a2 * b2, 2 * a2 * (a2 - b2), 2 * b2 * (b2 - a2)
    pts = [(0.0, n0), (a, na), (-a, na), (b, nb), (-b, nb)]
    g = np.array([[1 / n, p / n, p * p / n] for p, n in pts] + [[0, 0, 1]], float)
    at = np.array([[p ** i for p, _ in pts] + [1.0 if i == 3 else 0.0] for i in range(4)], float)
    v4 = np.array([[1, p, p * p, p ** 3] for p, _ in pts] + [[0, 0, 0, 1]], float)
    d6 = np.array([1 / n for _, n in pts] + [1.0])
    at3 = np.array([[p ** i for p, _ in pts] + [1.0 if i == 2 else 0.0] for i in range(3)], float)
    return bt, g, at, v4, d6, at3


BT6, G6, AT6, V4, D6, AT3 = w4_matrices(WA, WB)


def f4(d6, g3):
    return AT6 @ ((G6 @ g3 @ G6.T) * (BT6 @ d6 @ BT6.T)) @ AT6.T


def test_f4x4_3x3_and_its_transforms_as_the_kernel_evaluates_them():
    rng = np.random.default_rng(3)
    d, g = rng.standard_normal((6, 6)), rng.standard_normal((3, 3))
    assert np.allclose(f4(d, g), corr(d, g), atol=1e-11)
    # every constant the kernels hard-code is exact in fp32 (dyadic points)
    for v in (WA, WB, WA2, WB2, WA2B2, WS2, WA ** 3, WB ** 3):
        assert float(np.float32(v)) == v
    assert WA2 - WB2 == -1.0
    # bt6_cols / bt6_row: E_a = d4 - b2 d2, O_a = d3 - b2 d1, rows +-a = E_a +- a O_a; the same with (a2, b) for +-b; rows 0 / infinity
    x = rng.standard_normal(6)
    ea, oa, eb, ob = x[4] - WB2 * x[2], x[3] - WB2 * x[1], x[4] - WA2 * x[2], x[3] - WA2 * x[1]
    got = [WA2B2 * x[0] + (x[4] - WS2 * x[2]), ea + WA * oa, ea - WA * oa, eb + WB * ob, eb - WB * ob, WA2B2 * x[1] + (x[5] - WS2 * x[3])]
    assert np.allclose(got, BT6 @ x)
    # the packed horizontal pass: (E_a, E_b) = t2 (-b2, -a2) + t4, (O_a, O_b) = t1 (-b2, -a2) + t3, (v1, v3) = (O) (a, b) + (E), (v2, v4) = (O) (-a, -b) + (E),
    # (v0, v5) = a2b2 (t0, t1) + ((t4, t5) - s2 (t2, t3))
    k1, k2 = np.array([-WB2, -WA2]), np.array([WA, WB])
    E, O = x[2] * k1 + x[4], x[1] * k1 + x[3]
    v13, v24 = O * k2 + E, O * -k2 + E
    v05 = WA2B2 * x[[0, 1]] + (x[[4, 5]] - WS2 * x[[2, 3]])
    assert np.allclose([v05[0], v13[0], v24[0], v13[1], v24[1], v05[1]], BT6 @ x)
    # at6_pk
    m = rng.standard_normal(6)
    p, q, uu, w = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
    assert np.allclose([m[0] + p + uu, WB * w + WA * q, WB2 * uu + WA2 * p, WB ** 3 * w + WA ** 3 * q + m[5]], AT6 @ m)
    # the points of rounds 3-4 come out of the same construction (Lavin & Gray's matrices)
    bt_l, g_l, at_l, *_ = w4_matrices(1.0, 2.0)
    assert np.allclose(bt_l[0], [4, 0, -5, 0, 1, 0]) and np.allclose(g_l[3], [1 / 24, 1 / 12, 1 / 6]) and np.allclose(at_l[3], [0, 1, -1, 8, -8, 1])


def _fp32_model(a, b, C, T=32, seed=0):
    """kernel-faithful fp32 model of F(4x4, 3x3) at the points 0, +-a, +-b, inf: filter transform in float64 rounded once, data transform in fp32
    fma form, products accumulated over C channels sequentially in fp32 (what the MFMA chain does), output transform in fp32; -> max error
    against the float64 direct correlation over the output scale"""
    bt, g6, at, *_ = w4_matrices(a, b)
    f32 = lambda v: np.asarray(v, dtype=np.float32)
    fma = lambda x, y, z: f32(np.float64(x) * np.float64(y) + np.float64(z))
    a2, b2, a2b2, s2 = (np.float32(v) for v in (a * a, b * b, a * a * b * b, a * a + b * b))
    fa, fb, a3, b3 = (np.float32(v) for v in (a, b, a ** 3, b ** 3))
    rng = np.random.default_rng(seed)
    d = rng.uniform(-1, 1, (T, C, 6, 6))
    g = rng.uniform(-1, 1, (C, 3, 3)) * (C * 9) ** -0.5
    ref = sum(np.einsum("tcij,c->tij", d[:, :, i:i + 4, j:j + 4], g[:, i, j]) for i in range(3) for j in range(3))
    U = f32(np.einsum("xa,cab,yb->cxy", g6, g, g6))

    def btf(v, axis):
        d0, d1, d2, d3, d4, d5 = np.moveaxis(v, axis, 0)
        ea, oa, eb, ob = fma(-b2, d2, d4), fma(-b2, d1, d3), fma(-a2, d2, d4), fma(-a2, d1, d3)
        o = [fma(a2b2, d0, fma(-s2, d2, d4)), fma(fa, oa, ea), fma(-fa, oa, ea), fma(fb, ob, eb), fma(-fb, ob, eb), fma(a2b2, d1, fma(-s2, d3, d5))]
        return np.moveaxis(np.stack(o), 0, axis)

    def atf(v, axis):
        m0, m1, m2, m3, m4, m5 = np.moveaxis(v, axis, 0)
        p, q, u, w = f32(m1 + m2), f32(m1 - m2), f32(m3 + m4), f32(m3 - m4)
        o = [f32(f32(m0 + p) + u), fma(fb, w, f32(fa * q)), fma(b2, u, f32(a2 * p)), f32(fma(b3, w, f32(a3 * q)) + m5)]
        return np.moveaxis(np.stack(o), 0, axis)
    V = btf(btf(f32(d), 2), 3)
    acc = np.zeros((T, 6, 6), dtype=np.float32)
    for c in range(C):
        acc = fma(U[c], V[:, c], acc)
    y = atf(atf(acc, 1), 2)
    return float(np.abs(y.astype(np.float64) - ref).max() / np.abs(ref).max())


def test_point_set_error_model():
    """Why the kernels left Lavin & Gray's points: in fp32 the accumulation over K happens in the transform domain and the output transform
    cancels large terms.  The model reproduces what the GPU measured with 0, +-1, +-2 (round 4: 5e-6 at 96 channels, 1.2e-5 .. 2.3e-5 on
    the 5x5 stride-2 layers = 1 024 accumulated terms) and puts 0, +-3/4, +-5/4 three to five times lower -- inside the 1e-5 gate of the
    plan replay (tests/test_gpu_tuned_plans.py) with a factor of two to spare."""
    old = [max(_fp32_model(1.0, 2.0, c, seed=s) for s in range(2)) for c in (96, 1024)]
    new = [max(_fp32_model(WA, WB, c, seed=s) for s in range(2)) for c in (96, 1024)]
    assert 2e-6 < old[0] < 1e-5 and 1e-5 < old[1] < 4e-5, old        # the round-4 measurements
    assert new[0] < 3e-6 and new[1] < 7e-6, new
    assert new[1] < old[1] / 3, (old, new)


def test_5x5_stride2_conv_as_four_parity_subfilters():
    """out[o] = sum_t w[t] x[2 o - 2 + t], t = 2 a + p: sub-filter (ph, pw) element (a, b) = w[2 a + ph][2 b + pw] correlated with the parity
    plane x[2 m + ph][2 n + pw] at pad 1 (wino4_launch, mode 2)"""
    rng = np.random.default_rng(4)
    H = W = 16
    x, w = rng.standard_normal((H, W)), rng.standard_normal((5, 5))
    xp = np.pad(x, 2)
    ref = np.array([[(xp[2 * a:2 * a + 5, 2 * b:2 * b + 5] * w).sum() for b in range(W // 2)] for a in range(H // 2)])
    out = np.zeros_like(ref)
    for ph in range(2):
        for pw in range(2):
            sub = np.zeros((3, 3))
            for a in range(3):
                for b in range(3):
                    if 2 * a + ph < 5 and 2 * b + pw < 5:
                        sub[a, b] = w[2 * a + ph, 2 * b + pw]
            plane = np.pad(x[ph::2, pw::2], 1)
            for ty in range(0, H // 2, 4):          # F(4x4) tiles of the sub-convolution, accumulated over the four planes
                for tx in range(0, W // 2, 4):
                    out[ty:ty + 4, tx:tx + 4] += f4(plane[ty:ty + 6, tx:tx + 6], sub)
    assert np.allclose(out, ref, atol=1e-10)


def test_5x5_stride2_transposed_conv_as_four_output_phases():
    """out[2 u + py] = sum_a' w[2 (2 - a') + py] in[u - 1 + a']: phase (py, px) is a 3x3 pad-1 correlation of the input with
    w[2 (2 - a') + py][2 (2 - b') + px] (absent taps zero) written to every second output pixel (wino4_launch, mode 3)"""
    rng = np.random.default_rng(5)
    H = W = 8
    x, w = rng.standard_normal((H, W)), rng.standard_normal((5, 5))
    ref = np.zeros((2 * H + 4, 2 * W + 4))          # ConvTranspose2d k5 s2 p2 op1: scatter, then crop 2 at the top / left, 1 at the bottom / right
    for i in range(H):
        for j in range(W):
            ref[2 * i:2 * i + 5, 2 * j:2 * j + 5] += x[i, j] * w
    ref = ref[2:2 + 2 * H, 2:2 + 2 * W]
    out = np.zeros_like(ref)
    xp = np.pad(x, 1)
    for py in range(2):
        for px in range(2):
            sub = np.zeros((3, 3))
            for a in range(3):
                for b in range(3):
                    r, c = 2 * (2 - a) + py, 2 * (2 - b) + px
                    if r < 5 and c < 5:
                        sub[a, b] = w[r, c]
            for ty in range(0, H, 4):
                for tx in range(0, W, 4):
                    out[py + 2 * ty:py + 2 * ty + 8:2, px + 2 * tx:px + 2 * tx + 8:2] = f4(xp[ty:ty + 6, tx:tx + 6], sub)
    assert np.allclose(out, ref, atol=1e-10)


def test_5x5_stride1_as_four_shifted_subfilters_and_k_splits():
    """taps (3 bi + a, 3 bj + b): sub-filter (bi, bj) over the patch displaced by (3 bi, 3 bj) (mode 4); and the linearity the K splits rely
    on: the output transform of a sum of partial products = the sum of the output transforms (wino4_finish, SPLIT)"""
    rng = np.random.default_rng(6)
    d, g = rng.standard_normal((9, 9)), rng.standard_normal((5, 5))          # a 4x4 output tile of a 5x5 correlation needs 8x8; the
    gp = np.zeros((6, 6)); gp[:5, :5] = g                                    # displaced 6x6 windows reach one row / column further
    y = sum(f4(d[3 * bi:3 * bi + 6, 3 * bj:3 * bj + 6], gp[3 * bi:3 * bi + 3, 3 * bj:3 * bj + 3]) for bi in range(2) for bj in range(2))
    assert np.allclose(y, corr(d[:8, :8], g), atol=1e-10)
    C = 12
    dc, gc = rng.standard_normal((C, 6, 6)), rng.standard_normal((C, 3, 3))
    prod = np.stack([(G6 @ gc[c] @ G6.T) * (BT6 @ dc[c] @ BT6.T) for c in range(C)])
    whole = AT6 @ prod.sum(0) @ AT6.T
    parts = sum(AT6 @ prod[s:s + 4].sum(0) @ AT6.T for s in range(0, C, 4))
    assert np.allclose(whole, parts, atol=1e-11) and np.allclose(whole, sum(corr(dc[c], gc[c]) for c in range(C)), atol=1e-10)


# ---- csrc/wino4_wgrad.hip: F(3x3, 4x4), the transposition of F(4x4, 3x3) (same points, same B^T): V4 = rows [1, x, x^2, x^3] (v4_pk), D6 = 1 / N_j
# (left out of the K loop), AT3 (the epilogue) -- all from w4_matrices above


def wg4(p4, q6):
    """3x3 taps of the correlation of a 6x6 patch with a 4x4 tile: accumulate (V p V^T) . (B^T q B), scale by D D^T once, then A^T . A"""
    u = (V4 @ p4 @ V4.T) * (BT6 @ q6 @ BT6.T)
    return AT3 @ (u * np.outer(D6, D6)) @ AT3.T


def test_f3x3_4x4_weight_gradient_and_its_5x5_forms():
    rng = np.random.default_rng(7)
    p, q = rng.standard_normal((4, 4)), rng.standard_normal((6, 6))
    ref = np.array([[(p * q[r:r + 4, s:s + 4]).sum() for s in range(3)] for r in range(3)])
    assert np.allclose(wg4(p, q), ref, atol=1e-11)
    # v4_pk() as the kernel evaluates it: rows +-a = E_a +- a D_a, E_a = p0 + a2 p2, D_a = p1 + a2 p3
    x = rng.standard_normal(4)
    ea, da, eb, db = x[0] + WA2 * x[2], x[1] + WA2 * x[3], x[0] + WB2 * x[2], x[1] + WB2 * x[3]
    assert np.allclose([x[0], ea + WA * da, ea - WA * da, eb + WB * db, eb - WB * db, x[3]], V4 @ x)
    # the epilogue's A^T (3 x 6) on scaled accumulators
    u = rng.standard_normal(6)
    pp, qq, uu, ww = u[1] + u[2], u[1] - u[2], u[3] + u[4], u[3] - u[4]
    assert np.allclose([u[0] + pp + uu, WB * ww + WA * qq, WB2 * uu + WA2 * pp + u[5]], AT3 @ u)
    assert np.allclose(D6, [1 / WN0, 1 / WNA, 1 / WNA, 1 / WNB, 1 / WNB, 1.0])
    # the scaling commutes with the sum over tiles (it is applied once per workgroup)
    ps, qs = rng.standard_normal((5, 4, 4)), rng.standard_normal((5, 6, 6))
    u = sum((V4 @ a @ V4.T) * (BT6 @ b @ BT6.T) for a, b in zip(ps, qs))
    assert np.allclose(AT3 @ (u * np.outer(D6, D6)) @ AT3.T, sum(wg4(a, b) for a, b in zip(ps, qs)), atol=1e-10)
    # 5x5 stride 1, pad 2: g[r][s] = sum P[y][x] Q[y - 2 + r][x - 2 + s]; sub-filter (bi, bj) = taps (3 bi + a, 3 bj + b) from the patch displaced
    # by (3 bi, 3 bj); one 4x4 tile of P at the origin, Q given with its halo (index + 2)
    P, Q = rng.standard_normal((4, 4)), rng.standard_normal((4 + 7, 4 + 7))
    ref5 = np.array([[(P * Q[r:r + 4, s:s + 4]).sum() for s in range(5)] for r in range(5)])
    got5 = np.zeros((5, 5))
    for bi in range(2):
        for bj in range(2):
            g = wg4(P, Q[3 * bi:3 * bi + 6, 3 * bj:3 * bj + 6])
            na, nb = (2 if bi else 3), (2 if bj else 3)
            got5[3 * bi:3 * bi + na, 3 * bj:3 * bj + nb] = g[:na, :nb]
    assert np.allclose(got5, ref5, atol=1e-10)
    # 5x5 stride 2, pad 2: g[r][s] = sum P[y][x] Q[2 y - 2 + r][2 x - 2 + s], r = 2 a + ph: sub-filter (ph, pw) over the parity plane
    Q2 = rng.standard_normal((2 * 4 + 5, 2 * 4 + 5))          # index + 2 (halo)
    ref2 = np.array([[sum(P[y, x] * Q2[2 * y + r, 2 * x + s] for y in range(4) for x in range(4)) for s in range(5)] for r in range(5)])
    got2 = np.zeros((5, 5))
    for ph in range(2):
        for pw in range(2):
            plane = Q2[ph::2, pw::2][:6, :6]                   # rows 2 (y + a) + ph: patch row u = y + a
            g = wg4(P, plane)
            na, nb = (2 if ph else 3), (2 if pw else 3)
            got2[ph::2, pw::2][:na, :nb] = g[:na, :nb]
    assert np.allclose(got2, ref2, atol=1e-10)

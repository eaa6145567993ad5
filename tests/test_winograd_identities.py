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

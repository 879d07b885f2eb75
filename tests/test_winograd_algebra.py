"""The algebra behind csrc/wn_wino.hip (Winograd F(4,3) form of the WN dilated convolution), restated in numpy: the transform
matrices, the split of the conditioning term over the products, and the rank-1 structure that the dilation-16 layer uses.
The HIP kernels hard-code these constants; the GPU tests (tests/test_waveglow_gpu.py) check the kernels against the oracle."""
import numpy as np

# Lavin & Gray's F(4,3) for the correlation y_j = sum_t g_t d_{j + t}  (our taps: g = (W-, W0, W+), d_i = x[l + (i - 1) d])
BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
               [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
              [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], float)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)

# coefficient of output j's conditioning term c_j in product k, per K slice (rows: the products of the slice's subset)
SUBSETS = {'A': [0, 1, 2, 5], 'B': [0, 3, 4, 5], 'C': [1, 2, 3, 4]}
COEF = {'A': np.array([[1, 0, -1, 0], [0, .5, .5, 0], [0, -.5, .5, 0], [0, -1, 0, 1]]),
        'B': np.array([[1, 0, -.25, 0], [0, .25, .125, 0], [0, -.25, .125, 0], [0, -4, 0, 1]]),
        'C': np.array([[2 / 3, 2 / 3, -1 / 6, -1 / 6], [2 / 3, -2 / 3, -1 / 6, 1 / 6],
                       [-1 / 6, -1 / 12, 1 / 6, 1 / 12], [-1 / 6, 1 / 12, 1 / 6, -1 / 12]])}


def test_f43_reproduces_the_three_tap_convolution():
    rng = np.random.default_rng(0)
    d = rng.standard_normal((6, 7))                      # six inputs x[l - d] .. x[l + 4 d], 7 channels
    w = rng.standard_normal((3, 5, 7))                   # taps (W-, W0, W+), 5 outputs x 7 channels
    direct = np.stack([sum(w[t] @ d[j + t] for t in range(3)) for j in range(4)])
    U = BT @ d                                           # transformed inputs, one row per product
    Gk = np.einsum('kt,tnc->knc', G, w)                  # transformed weights
    P = np.stack([Gk[k] @ U[k] for k in range(6)])       # the six products
    assert np.allclose(AT @ P, direct, atol=1e-12)


def test_conditioning_slices_reconstruct_every_output():
    """A K slice carried by a product subset Q with a_k = sum_j coef[k][j] c_j must give back c_j through the output transform:
    AT[:, Q] @ coef = identity -- i.e. coef is the inverse of those four columns, which therefore have rank 4."""
    for name, Q in SUBSETS.items():
        cols = AT[:, Q]
        assert np.linalg.matrix_rank(cols) == 4
        assert np.allclose(cols @ COEF[name], np.eye(4), atol=1e-12), name
    # the three subsets load the six products equally: every product carries exactly two slices
    counts = np.zeros(6, int)
    for Q in SUBSETS.values():
        counts[Q] += 1
    assert (counts == 2).all()
    # slice widths and the column layout of a product's K = 224
    SA, SB, SC = 112, 96, 112
    assert SA + SB + SC == 320 and max(SA + SB, SA + SC, SB + SC) == 224 and 224 % 16 == 0


def test_dilation16_rows_are_outer_products_of_frame_and_phase_coefficients():
    """Dilation 16: c_j = mel(t_j) V(p_j) with (t_j, p_j) = (t, p0), (t, p0 + 16), (t + 1, p0), (t + 1, p0 + 16).  A product's
    conditioning sum_j coef[j] c_j is ONE product (mel combination) x (weight combination) iff the 2 x 2 matrix
    [[coef0, coef1], [coef2, coef3]] (rows: frames, columns: phases) has rank 1.  True for every product of subset C -- and for no
    mixed row of A or B, which is why that layer puts the whole conditioning on products 1 .. 4."""
    for row, (u, v) in zip(COEF['C'], [((2 / 3, -1 / 6), (1, 1)), ((2 / 3, -1 / 6), (1, -1)),
                                       ((-1, 1), (1 / 6, 1 / 12)), ((-1, 1), (1 / 6, -1 / 12))]):
        m = row.reshape(2, 2)
        assert np.linalg.matrix_rank(m) == 1
        assert np.allclose(m, np.outer(u, v), atol=1e-12)            # the combinations wn_wino.hip builds
    assert np.linalg.matrix_rank(COEF['A'][1].reshape(2, 2)) == 2 and np.linalg.matrix_rank(COEF['B'][1].reshape(2, 2)) == 2
    # end to end on random data: products 1 .. 4 with K = 320 conditioning, products 0 and 5 without
    rng = np.random.default_rng(1)
    mel = rng.standard_normal((2, 320))                                # frames t, t + 1
    V = rng.standard_normal((2, 9, 320))                               # phases p0, p0 + 16: 9 outputs
    c = np.stack([V[j & 1] @ mel[j >> 1] for j in range(4)])           # c_j in the output order above
    a = np.zeros((6, 9))
    for k, (u, v) in zip((1, 2, 3, 4), [((2 / 3, -1 / 6), (1, 1)), ((2 / 3, -1 / 6), (1, -1)),
                                        ((-1, 1), (1 / 6, 1 / 12)), ((-1, 1), (1 / 6, -1 / 12))]):
        a[k] = (v[0] * V[0] + v[1] * V[1]) @ (u[0] * mel[0] + u[1] * mel[1])
    assert np.allclose(AT @ a, c, atol=1e-12)


def test_group_maps_cover_every_position_once():
    """Phase groups (d = 2, 4, 8): p0 = (gp / d) 4d + gp % d over 8 group phases; frame groups (s = 1, 2, 4): t0 = (g / s) 4s +
    g % s over 4 ceil(T / 16) groups of an utterance; dilation 16: p0 < 16, t0 = 2 g."""
    for d in (2, 4, 8):
        seen = sorted((gp // d) * 4 * d + gp % d + j * d for gp in range(8) for j in range(4))
        assert seen == list(range(32))
    for T in (1, 5, 16, 121, 126, 800):
        G = (T + 15) // 16 * 4
        for s in (1, 2, 4):
            seen = sorted(t for g in range(G) for t in ((g // s) * 4 * s + g % s + j * s for j in range(4)) if t < T)
            assert seen == list(range(T)), (T, s)
        seen = sorted((p0 + 16 * (j & 1), 2 * g + (j >> 1)) for p0 in range(16) for g in range((T + 1) // 2) for j in range(4)
                      if 2 * g + (j >> 1) < T)
        assert seen == sorted((p, t) for p in range(32) for t in range(T))

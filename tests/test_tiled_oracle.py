"""CPU checks of the oracle's restatement of the tile-scaled systematic resampler (oracle/gjx_oracle.c
gjxo_resample_systematic_tiled; scheme defined in include/gjx.h, GJX_WEIGHTS_TILE_SCALED): the scheme against a float64
systematic resampler and its own edge cases.  The GPU parity tests (tests/test_gpu_tiled.py) compare the device with it."""
import numpy as np
import pytest

from oracle import cpu as oracle


def _exact_systematic(lw, u, N):
    """float64 inverse-CDF systematic resampling on the exact normalised weights"""
    w = np.exp(lw.astype(np.float64) - np.nanmax(lw[np.isfinite(lw)]))
    w[~np.isfinite(w)] = 0.0
    c = np.cumsum(w)
    t = (np.arange(N) + u) * (c[-1] / N)
    return np.minimum(np.searchsorted(c, t, side="right"), lw.size - 1)


@pytest.mark.parametrize("K,N", [(1, 1), (5, 5), (1024, 1024), (1025, 1025), (5000, 5000), (70_001, 70_001), (4096, 1000), (3000, 9001)])
def test_tiled_matches_float64_systematic(K, N):
    rs = np.random.default_rng(K + N)
    lw = (rs.standard_normal(K) * 3.0 - 40.0).astype(np.float32)
    anc, q, e, dead = oracle.resample_systematic_tiled(lw, 0.37, N)
    assert not dead
    assert anc.min() >= 0 and anc.max() < K and (np.diff(anc) >= 0).all()
    ref = _exact_systematic(lw, 0.37, N)
    # a threshold may fall within the quantisation step of a particle boundary: rare, and then the neighbour is taken
    bad = anc != ref
    assert bad.mean() <= 2e-3 + 2.0 / N
    wx = np.exp(lw.astype(np.float64) - lw.max())
    cx = np.concatenate([[0.0], np.cumsum(wx)])
    lo, hi = np.minimum(anc[bad], ref[bad]), np.maximum(anc[bad], ref[bad])
    assert ((cx[hi] - cx[lo + 1]) <= 5e-3 * cx[-1] / N).all()         # float32 exp2: thresholds move by ~1e-3 of a comb step
    # offspring counts within one of N w
    wn = np.exp(lw.astype(np.float64) - lw.max())
    wn /= wn.sum()
    assert np.abs(np.bincount(anc, minlength=K) - N * wn).max() <= 1.0 + 1e-3 * N * wn.max()
    # the tile exponents: 2^(e-1) < max weight of the tile <= 2^e, largest q in (2^28, 2^29]
    for b in range(e.size):
        m = lw[1024 * b:1024 * b + 1024].max() * np.float32(1.44269504)
        assert e[b] - 1 < m <= e[b]
        assert (1 << 28) < q[1024 * b:1024 * b + 1024].max() <= (1 << 29)


def test_tiled_scale_separated_tiles():
    """tiles whose maxima differ by more than the fixed point can hold contribute nothing, exactly like particles whose
    weight underflows under the global-maximum scheme; a dead tile (-inf / NaN) is skipped"""
    K = 5 * 1024
    lw = np.full(K, -5.0, np.float32)
    lw[1024:2048] = -200.0                    # 2^-281 of the rest: shifted out
    lw[2048:3072] = -np.inf
    lw[3072:4096] = np.nan
    lw[4096:] = -5.0 + np.log(3.0)            # three times the weight of tile 0
    anc, q, e, dead = oracle.resample_systematic_tiled(lw, 0.5, K)
    assert not dead
    cnt = np.bincount(anc // 1024, minlength=5)
    assert cnt[1] == cnt[2] == cnt[3] == 0
    assert abs(cnt[0] - K / 4) <= 1 and abs(cnt[4] - 3 * K / 4) <= 1
    assert e[2] == e[3] == -524288 and (q[2048:4096] == 0).all()
    # inside tile 0 all particles are equal
    assert np.abs(np.bincount(anc[anc < 1024], minlength=1024) - 1.25).max() < 1     # equal particles: 1280 offspring over 1024


def test_tiled_dead_collection_and_huge_range():
    lw = np.full(3000, -np.inf, np.float32)
    anc, _, e, dead = oracle.resample_systematic_tiled(lw, 0.2, 3000)
    assert dead and (anc == np.arange(3000)).all() and (e == -524288).all()
    # one live particle among dead ones
    lw[1234] = -1e4
    anc, _, e, dead = oracle.resample_systematic_tiled(lw, 0.2, 3000)
    assert not dead and (anc == 1234).all() and e[1] == int(np.ceil(np.float32(-1e4) * np.float32(1.44269504)))
    # far below anything a filter produces, still exact: the exponent has 20 bits
    lw = np.array([-300000.0, -300000.5, -300001.0], np.float32)
    anc, q, e, dead = oracle.resample_systematic_tiled(lw, 0.0, 3000)
    assert not dead and e[0] == int(np.ceil(np.float32(-300000.0) * np.float32(1.44269504)))
    cnt = np.bincount(anc, minlength=3)
    assert np.abs(cnt - 3000 * np.exp([0, -0.5, -1.0]) / np.exp([0, -0.5, -1.0]).sum()).max() <= 40     # float32 spacing of log w is 1/32 here
    # beyond the clamp (|log w| > 3.6e5) the collection is reported dead instead of being resampled wrongly
    lw = np.array([-400000.0, -400000.5], np.float32)
    assert oracle.resample_systematic_tiled(lw, 0.0, 10)[3]


def test_tiled_uses_given_quantisation():
    rs = np.random.default_rng(0)
    lw = rs.standard_normal(4000).astype(np.float32)
    a0, q0, e0, _ = oracle.resample_systematic_tiled(lw, 0.9, 4000)
    a1, q1, e1, _ = oracle.resample_systematic_tiled(lw, 0.9, 4000, q=q0)
    assert (a0 == a1).all() and (q0 == q1).all() and (e0 == e1).all()
    q2 = q0.copy()
    q2[:] = 1 << 20                            # flat weights given: every particle once
    a2, _, _, _ = oracle.resample_systematic_tiled(lw, 0.9, 4000, q=q2)
    assert (a2 == np.arange(4000)).all()


# ---- multinomial resampling by sorted uniforms (gjxo_resample_sorted_multinomial_tiled; include/gjx.h,
#      gjx_resample_sorted_multinomial_tiled) ----
def test_exp_spacing_is_the_fixed_point_negative_log2():
    """e(w) = floor(2^20 * -log2(x / 2^24)), x = 2 (w >> 9) + 1: against float64 within the polynomial's error (7e-8 in log2 ->
    under 1 count of 2^-20 ... a few counts after the float32 chain), exact at the ends, monotone in x"""
    rs = np.random.default_rng(0)
    words = np.concatenate([rs.integers(0, 1 << 32, 20000, dtype=np.uint64), [0, 511, 512, (1 << 32) - 1, 1 << 31, (1 << 31) - 1]]).astype(np.uint64)
    got = np.array([oracle.exp_spacing(int(w)) for w in words], np.float64)
    x = 2.0 * (words >> np.uint64(9)).astype(np.float64) + 1.0
    want = -np.log2(x / 2.0 ** 24) * 2.0 ** 20
    assert np.abs(got - want).max() < 8.0, np.abs(got - want).max()
    assert oracle.exp_spacing(0) == oracle.exp_spacing(511) == 24 << 20          # x = 1
    assert oracle.exp_spacing((1 << 32) - 1) in (0, 1)                            # x = 2^24 - 1: -log2 = 8.6e-8 -> 0.09 counts
    order = np.argsort(x, kind="stable")
    assert (np.diff(got[order]) <= 0).all()


def _exact_sorted_multinomial(lw, key, N):
    """the same sorted uniforms (from the oracle's own spacings), float64 inverse CDF on the exact normalised weights"""
    from oracle.cpu import threefry2x32
    sp = np.array([oracle.exp_spacing(int(threefry2x32(key[0], key[1], 0, j)[0])) for j in range(N + 1)], np.float64)
    u = np.cumsum(sp)[:N] / sp.sum()
    w = np.exp(lw.astype(np.float64) - np.nanmax(lw[np.isfinite(lw)]))
    w[~np.isfinite(w)] = 0.0
    c = np.cumsum(w)
    return np.minimum(np.searchsorted(c, u * c[-1], side="right"), lw.size - 1)


@pytest.mark.parametrize("K,N", [(1, 1), (5, 5), (1024, 1024), (1025, 1025), (5000, 5000), (4096, 1000), (3000, 9001)])
def test_sorted_multinomial_matches_float64_inverse_cdf(K, N):
    rs = np.random.default_rng(K + N)
    lw = (rs.standard_normal(K) * 3.0 - 40.0).astype(np.float32)
    anc, dead = oracle.resample_sorted_multinomial_tiled(lw, (11, 13), N)
    assert not dead and anc.min() >= 0 and anc.max() < K and (np.diff(anc) >= 0).all()
    ref = _exact_sorted_multinomial(lw, (11, 13), N)
    bad = anc != ref                 # a threshold within the quantisation step of a particle boundary: rare, then a neighbour
    assert bad.mean() <= 2e-3 + 2.0 / N
    assert (np.abs(anc[bad].astype(np.int64) - ref[bad]) <= 2).all() or K < 16


def test_sorted_multinomial_is_a_multinomial_draw():
    """counts over repeated draws: mean N p_i and the multinomial's variance N p_i (1 - p_i) — a systematic comb would show a variance
    below 1/4; the uniforms are order statistics of iid draws: the FIRST one is Beta(1, N), mean 1 / (N + 1)"""
    K = 64
    rs = np.random.default_rng(5)
    lw = rs.standard_normal(K).astype(np.float32)
    p = np.exp(lw.astype(np.float64)); p /= p.sum()
    N, R = 256, 600
    counts = np.stack([np.bincount(oracle.resample_sorted_multinomial_tiled(lw, (3, r), N)[0], minlength=K) for r in range(R)])
    se = np.sqrt(N * p * (1 - p) / R)
    assert (np.abs(counts.mean(0) - N * p) < 4.5 * se).all()
    ratio = counts.var(0, ddof=1) / (N * p * (1 - p))
    assert 0.85 < ratio.mean() < 1.15 and ratio.min() > 0.6, (ratio.mean(), ratio.min())


def test_sorted_multinomial_dead_collection_and_given_quantisation():
    lw = np.full(3000, -np.inf, np.float32)
    anc, dead = oracle.resample_sorted_multinomial_tiled(lw, (1, 2))
    assert dead and (anc == np.arange(3000)).all()
    q = np.zeros(3000, np.uint32)
    q[[7, 2999]] = 5
    anc, dead = oracle.resample_sorted_multinomial_tiled(np.zeros(3000, np.float32), (1, 2), 4000, q=q)
    assert not dead and set(np.unique(anc)) == {7, 2999} and (np.diff(anc) >= 0).all()
    assert abs((anc == 7).mean() - 0.5) < 0.05

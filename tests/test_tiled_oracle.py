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

"""Scan steps on the CPU oracle: the chained step-key rule (include/gjx.h "Scan steps", scan.py:268) and Scans longer
than the 1023 site numbers of one FLAT stream.  The GPU side of the same checks is tests/test_gpu_scan.py."""
import numpy as np

import helpers as H
from genjax_amd import _abi as A
from genjax_amd import core


def chain_keys(key, T, scan_id=0):
    """key_t = fold_in(key_{t-1}, t), key_{-1} = fold_in(run key, 0x80000000 | scan_id) — computed with the host's
    own Threefry, independently of the oracle's walk over the site list."""
    k = core.fold_in(key, 0x80000000 | scan_id)
    out = []
    for t in range(T):
        k = core.fold_in(k, t)
        out.append(k)
    return out


def test_step_streams_are_the_single_site_stream_under_the_chained_key(oracle):
    T, K, key = 7, 64, (1234, 5678)
    one = H.one_site("normal", 0.0, 1.0)
    for scan_id in (0, 3):
        prog, _ = H.scan_chain(T, carry=False, scan_id=scan_id)
        got = oracle.run_program(prog, key, K)["choices"]                       # [T, K]
        for t, kt in enumerate(chain_keys(key, T, scan_id)):
            want = oracle.run_program(one, kt, K)["choices"][0]                # site number 1 under key_t
            np.testing.assert_array_equal(got[t], want)
    # site numbers count within the step: an observed y_t behind x_t leaves every x stream where it was
    a, _ = H.scan_chain(5, carry=True)
    b, _ = H.scan_chain(5, carry=True, observe=True)
    xa = oracle.run_program(a, key, K)["choices"]
    xb = oracle.run_program(b, key, K)["choices"]
    np.testing.assert_array_equal(xa, xb[: xa.shape[0]])


def test_scan_longer_than_one_stream_has_sites(oracle):
    T, K = 1500, 32                                                             # 1500 sites > GJX_FLAT_MAX_SITES
    prog, _ = H.scan_chain(T, carry=True, sigma=0.1)
    out = oracle.run_program(prog, (7, 9), K)
    x = out["choices"]
    assert x.shape == (T, K) and np.isfinite(x).all()
    inc = np.diff(np.vstack([np.zeros((1, K), np.float32), x]), axis=0) / 0.1   # standard normal increments
    assert abs(inc.mean()) < 0.02 and abs(inc.std() - 1.0) < 0.02
    # independent steps: increments of different steps are uncorrelated (a repeated key would make them equal)
    assert not np.array_equal(inc[0], inc[1]) and not np.array_equal(inc[5], inc[1029])
    want = (-0.5 * inc.astype(np.float64) ** 2 - 0.5 * np.log(2 * np.pi) - np.log(0.1)).sum(axis=0)
    np.testing.assert_allclose(out["score"], want, rtol=2e-5)

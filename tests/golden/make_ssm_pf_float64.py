"""Generates tests/golden/ssm_pf_float64.json: the Monte-Carlo spread of an ideal (float64, NumPy) bootstrap particle
filter at BASELINE config 3 — linear-Gaussian SSM d_x = d_y = 8, T = 256, K = 2^18 particles, systematic resampling every
step — over 16 seeds, against the float64 Kalman log-likelihood.

    python tests/golden/make_ssm_pf_float64.py             (about 8 minutes on 8 cores)

The device filter quantises weights to fixed point and computes in float32; this fixture says how much of its log-ML
error is plain Monte-Carlo spread (tests/test_gpu_parity.py::test_bootstrap_filter_seeds_vs_float64_filter compares
the rms of 32 device runs with the rms recorded here).  No code or stream is shared with the device path.
"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from genjax_amd import workloads as W      # noqa: E402  (problem generator: NumPy)
from oracle import closed_form as cf       # noqa: E402


def run(seed, K=1 << 18):
    s = W.ssm_problem()
    A, y, q, r = s["A"].astype(np.float64), s["y"].astype(np.float64), float(s["q"]), float(s["r"])
    T, dx = y.shape
    rs = np.random.default_rng(1000 + seed)
    x = rs.standard_normal((K, dx))
    log_ml = 0.0
    c = -0.5 * dx * np.log(2 * np.pi) - dx * np.log(r)
    for t in range(T):
        if t > 0:
            x = x @ A.T + q * rs.standard_normal((K, dx))
        d = (y[t] - x) / r
        lw = c - 0.5 * (d * d).sum(1)
        m = lw.max()
        w = np.exp(lw - m)
        tot = w.sum()
        log_ml += m + np.log(tot) - np.log(K)
        cum = np.cumsum(w)
        u = (np.arange(K) + rs.uniform()) * (tot / K)
        anc = np.minimum(np.searchsorted(cum, u, side="right"), K - 1)
        x = x[anc]
    return log_ml


def main():
    s = W.ssm_problem()
    exact, _, _ = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"])
    with mp.Pool(8) as pool:
        est = pool.map(run, range(16))
    rel = [(e - exact) / abs(exact) for e in est]
    out = dict(config="SSM dx=dy=8 q=0.5 r=2.0 T=256 data seed 0; bootstrap PF K=2^18, systematic resampling every step",
               filter="NumPy float64, seeds 1000..1015", kalman_log_lik=exact, log_ml=est, rel_err=rel,
               rms_rel_err=float(np.sqrt(np.mean(np.square(rel)))), max_abs_rel_err=float(np.max(np.abs(rel))))
    with open(os.path.join(HERE, "ssm_pf_float64.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)


if __name__ == "__main__":
    main()

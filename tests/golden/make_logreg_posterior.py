"""Generates tests/golden/logreg_posterior.json: the posterior mean of (log_tau, beta) of BASELINE config 5
(hierarchical logistic regression N = 1024, P = 16, data seed 0 — genjax_amd/workloads.py::logreg_problem) from a LONG
float64 run, the check SURVEY.md §8(d) row 5 asks for ("posterior mean of beta vs long float64 reference run, 3 sigma_MC").

    python tests/golden/make_logreg_posterior.py            (about a minute on 8 cores; NumPy float64 only)

The sampler is an ordinary HMC with Metropolis correction and jittered trajectory lengths (so it shares no integrator
setting, random stream or code with the device kernels): 512 chains, 300 burn-in moves, 1200 kept moves.  Recorded:
mean, posterior standard deviation, and the Monte-Carlo standard error of the mean (from the between-chain variance of
the per-chain means, which is valid whatever the within-chain autocorrelation), plus the split-R-hat of each coordinate.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from genjax_amd import workloads as W  # noqa: E402   (problem generator only: NumPy)


def log_density_and_grad(q, X, y):
    """q[chains, 1 + P] = (log_tau, beta).  log p = N(log_tau; 0, 1) + sum_p N(beta_p; 0, exp(log_tau)) +
    sum_n bernoulli_logits(y_n; X_n . beta)  (reference model: SURVEY.md §8(d) row 5)."""
    lt, b = q[:, 0], q[:, 1:]
    P = b.shape[1]
    s2 = np.exp(-2.0 * lt)
    bb = (b * b).sum(1)
    z = b @ X.T                                           # [chains, N]
    lp = -0.5 * lt * lt - P * lt - 0.5 * bb * s2 + (y * z - np.logaddexp(0.0, z)).sum(1)
    g = np.empty_like(q)
    g[:, 0] = -lt - P + bb * s2
    g[:, 1:] = -b * s2[:, None] + (y - 1.0 / (1.0 + np.exp(-z))) @ X
    return lp, g


def main():
    pr = W.logreg_problem(1024, 16, 0)
    X, y = pr["X"].astype(np.float64), pr["y"].astype(np.float64)
    rs = np.random.default_rng(20260930)
    C, burn, keep = 512, 300, 1200
    q = rs.standard_normal((C, 17)) * 0.1
    lp, g = log_density_and_grad(q, X, y)
    sums = np.zeros((C, 17))
    sq = np.zeros((C, 17))
    half = np.zeros((2, C, 17))
    acc = 0.0
    for it in range(burn + keep):
        eps = 0.012 * rs.uniform(0.8, 1.2)
        L = int(rs.integers(15, 45))
        p = rs.standard_normal(q.shape)
        q1, g1, p1 = q.copy(), g.copy(), p.copy()
        for _ in range(L):
            p1 += 0.5 * eps * g1
            q1 += eps * p1
            lp1, g1 = log_density_and_grad(q1, X, y)
            p1 += 0.5 * eps * g1
        dH = (lp1 - 0.5 * (p1 * p1).sum(1)) - (lp - 0.5 * (p * p).sum(1))
        ok = np.log(rs.uniform(size=C)) < dH
        q[ok], g[ok], lp[ok] = q1[ok], g1[ok], lp1[ok]
        if it >= burn:
            acc += ok.mean()
            sums += q
            sq += q * q
            half[(it - burn) * 2 // keep] += q
    chain_means = sums / keep
    mean = chain_means.mean(0)
    var = (sq / keep).mean(0) - (chain_means ** 2).mean(0) + chain_means.var(0)
    se = chain_means.std(0, ddof=1) / np.sqrt(C)
    hm = half / (keep / 2)                                # split halves: [2, C, 17] means
    W_ = var
    B_ = np.concatenate([hm[0], hm[1]]).var(0, ddof=1)    # variance of half-chain means
    rhat = np.sqrt(1.0 + B_ / W_)
    out = dict(model="log_tau~N(0,1); beta_p~N(0,exp(log_tau)) P=16; y_n~bernoulli_logits(X_n.beta) N=1024; data seed 0",
               sampler="float64 NumPy HMC + MH, 512 chains, 300 burn-in + 1200 kept moves, eps 0.012*U(0.8,1.2), L U{15..44}",
               accept_rate=acc / keep, mean=mean.tolist(), sd=np.sqrt(var).tolist(), mc_se=se.tolist(),
               split_rhat_minus_1_of_half_means=(rhat - 1.0).tolist())
    with open(os.path.join(HERE, "logreg_posterior.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("accept", acc / keep, "max se", se.max(), "sd range", np.sqrt(var).min(), np.sqrt(var).max())
    print("mean", np.round(mean, 4))
    print("beta_true", np.round(pr["beta_true"], 3))


if __name__ == "__main__":
    main()

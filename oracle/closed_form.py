"""float64 closed-form answers the hot path is checked against.  TEST INFRASTRUCTURE ONLY.

NumPy only (no scipy at run time on the GPU box is assumed).  Sources of the cases:
  - beta-bernoulli          /root/reference/README.md:89-123
  - flip-flip               /root/reference/tests/inference/test_smc.py:32-87
  - Gaussian mixture        SURVEY.md §8(d) config 2 (model shape from test_smc.py:89-98)
  - linear-Gaussian SSM     SURVEY.md §8(d) config 3/4  (Kalman filter log-likelihood)
"""
from __future__ import annotations

import math

import numpy as np

from genjax_amd.workloads import gmm_problem, logreg_problem, ssm_problem  # noqa: F401  (workload definitions)


def log_normal_pdf(x, mu, sd):
    x, mu, sd = np.asarray(x, np.float64), np.asarray(mu, np.float64), np.asarray(sd, np.float64)
    z = (x - mu) / sd
    return -0.5 * z * z - np.log(sd) - 0.5 * math.log(2 * math.pi)


def logsumexp(a, axis=None):
    a = np.asarray(a, np.float64)
    m = np.max(a, axis=axis, keepdims=True)
    out = np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True)) + m
    return np.squeeze(out, axis=axis) if axis is not None else float(out.ravel()[0])


# ---- config 1 ------------------------------------------------------------------------------
def beta_bernoulli_log_ml(a: float, b: float, obs: bool) -> float:
    """p ~ beta(a,b); v ~ flip(p): P(v=True) = a/(a+b)."""
    return math.log(a / (a + b)) if obs else math.log(b / (a + b))


def beta_bernoulli_posterior_mean(a: float, b: float, obs: bool) -> float:
    return (a + 1) / (a + b + 1) if obs else a / (a + b + 1)


# ---- config 2 ------------------------------------------------------------------------------
def gmm_log_ml(logits, mu, sigma, r, y) -> float:
    """log sum_c pi_c prod_d N(y_d; mu_cd, sqrt(sigma_cd^2 + r_d^2))."""
    logits = np.asarray(logits, np.float64)
    logpi = logits - logsumexp(logits)
    sd = np.sqrt(np.asarray(sigma, np.float64) ** 2 + np.asarray(r, np.float64)[None, :] ** 2)
    ll = log_normal_pdf(np.asarray(y, np.float64)[None, :], np.asarray(mu, np.float64), sd).sum(axis=1)
    return float(logsumexp(logpi + ll))


def gmm_posterior_z(logits, mu, sigma, r, y) -> np.ndarray:
    logits = np.asarray(logits, np.float64)
    logpi = logits - logsumexp(logits)
    sd = np.sqrt(np.asarray(sigma, np.float64) ** 2 + np.asarray(r, np.float64)[None, :] ** 2)
    ll = log_normal_pdf(np.asarray(y, np.float64)[None, :], np.asarray(mu, np.float64), sd).sum(axis=1)
    lp = logpi + ll
    return np.exp(lp - logsumexp(lp))


# ---- config 3/4 ----------------------------------------------------------------------------
def kalman_log_lik(A, y, q, r, q0=1.0, H=None):
    """float64 Kalman filter: returns (total log-likelihood, per-step increments, filtered means)."""
    A = np.asarray(A, np.float64)
    y = np.asarray(y, np.float64)
    T, dy = y.shape
    dx = A.shape[0]
    Hm = np.eye(dx)[:dy] if H is None else np.asarray(H, np.float64)
    Q = q * q * np.eye(dx)
    R = r * r * np.eye(dy)
    m = np.zeros(dx)
    P = q0 * q0 * np.eye(dx)
    incs = np.zeros(T)
    means = np.zeros((T, dx))
    for t in range(T):
        if t > 0:
            m = A @ m
            P = A @ P @ A.T + Q
        S = Hm @ P @ Hm.T + R
        v = y[t] - Hm @ m
        Sinv = np.linalg.inv(S)
        sign, logdet = np.linalg.slogdet(S)
        incs[t] = -0.5 * (v @ Sinv @ v + logdet + dy * math.log(2 * math.pi))
        Kg = P @ Hm.T @ Sinv
        m = m + Kg @ v
        P = (np.eye(dx) - Kg @ Hm) @ P
        means[t] = m
    return float(incs.sum()), incs, means


def rts_smoother(A, y, q, r, q0=1.0, H=None):
    """float64 Rauch-Tung-Striebel smoother of the same model: E[x_t | y_{1:T}] for every t (and the covariances).
    The yardstick for trajectories reconstructed from the particle filter's ancestor history (scan.py:56-97 keeps the
    whole stacked trace; the build keeps per-step states + ancestors and follows them back)."""
    A = np.asarray(A, np.float64)
    y = np.asarray(y, np.float64)
    T, dy = y.shape
    dx = A.shape[0]
    Hm = np.eye(dx)[:dy] if H is None else np.asarray(H, np.float64)
    Q = q * q * np.eye(dx)
    R = r * r * np.eye(dy)
    m = np.zeros(dx)
    P = q0 * q0 * np.eye(dx)
    mf, Pf, mp, Pp = [], [], [], []
    for t in range(T):
        if t > 0:
            m = A @ m
            P = A @ P @ A.T + Q
        mp.append(m.copy()); Pp.append(P.copy())
        S = Hm @ P @ Hm.T + R
        Kg = P @ Hm.T @ np.linalg.inv(S)
        m = m + Kg @ (y[t] - Hm @ m)
        P = (np.eye(dx) - Kg @ Hm) @ P
        mf.append(m.copy()); Pf.append(P.copy())
    ms, Ps = [None] * T, [None] * T
    ms[-1], Ps[-1] = mf[-1], Pf[-1]
    for t in range(T - 2, -1, -1):
        G = Pf[t] @ A.T @ np.linalg.inv(Pp[t + 1])
        ms[t] = mf[t] + G @ (ms[t + 1] - mp[t + 1])
        Ps[t] = Pf[t] + G @ (Ps[t + 1] - Pp[t + 1]) @ G.T
    return np.stack(ms), np.stack(Ps)


# ---- config 5 ------------------------------------------------------------------------------
def logreg_log_joint(log_tau, beta, X, y):
    """log p(log_tau, beta, y) of: log_tau~N(0,1); beta_p~N(0, exp(log_tau)); y_n~Bernoulli(logits=X beta)."""
    log_tau = float(log_tau)
    beta = np.asarray(beta, np.float64)
    s = np.asarray(X, np.float64) @ beta
    yy = np.asarray(y, np.float64)
    ll = np.sum(yy * -np.logaddexp(0, -s) + (1 - yy) * -np.logaddexp(0, s))
    return float(log_normal_pdf(log_tau, 0.0, 1.0) + log_normal_pdf(beta, 0.0, math.exp(log_tau)).sum() + ll)

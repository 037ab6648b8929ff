"""Synthetic workloads of BASELINE.json (problem generators and their model programs).  NumPy only.

    config 2  gmm_problem / gmm_program        Gaussian-mixture Target, ImportanceK
    config 3  ssm_problem                      linear-Gaussian state-space model, bootstrap filter
    config 5  logreg_problem / logreg_program  hierarchical logistic regression, HMC
Shapes and seeds follow SURVEY.md §8(d).
"""
from __future__ import annotations

import math

import numpy as np

from . import _abi as A
from .program import PackedProgram, Param, SiteList


def _logsumexp(a):
    a = np.asarray(a, np.float64)
    m = a.max()
    return float(m + np.log(np.exp(a - m).sum()))


def gmm_problem(C: int = 8, D: int = 16, seed: int = 0, mu_range: float = 1.0, sigma: float = 1.0,
                r: float = 4.0):
    """Synthetic GMM of SURVEY §8(d): z~categorical(logits), x~N(mu[z], sigma), y~N(x, r)."""
    rng = np.random.default_rng(seed)
    logits = rng.standard_normal(C)
    mu = rng.uniform(-mu_range, mu_range, size=(C, D))
    sig = np.full((C, D), sigma)
    rr = np.full(D, r)
    z = rng.choice(C, p=np.exp(logits - _logsumexp(logits)))
    x = mu[z] + sig[z] * rng.standard_normal(D)
    y = x + rr * rng.standard_normal(D)
    f = np.float32
    return dict(logits=logits.astype(f), mu=mu.astype(f), sigma=sig.astype(f), r=rr.astype(f), y=y.astype(f))



def ssm_problem(dx: int = 8, T: int = 256, q: float = 0.5, r: float = 2.0, seed: int = 0):
    """A = block-diag of 2x2 blocks 0.9*Rot(theta_i), theta_i = 0.3 + 0.1*i (i = block start), H = I."""
    A = np.zeros((dx, dx))
    for i in range(0, dx, 2):
        th = 0.3 + 0.1 * i
        c, s = math.cos(th), math.sin(th)
        A[i:i + 2, i:i + 2] = 0.9 * np.array([[c, -s], [s, c]])
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(dx)
    ys = np.zeros((T, dx))
    for t in range(T):
        if t > 0:
            x = A @ x + q * rng.standard_normal(dx)
        ys[t] = x + r * rng.standard_normal(dx)
    return dict(A=A.astype(np.float32), y=ys.astype(np.float32), q=q, r=r, q0=1.0)



def logreg_problem(N: int = 1024, P: int = 16, seed: int = 0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, P))
    beta = rng.standard_normal(P)
    y = (rng.uniform(size=N) < 1.0 / (1.0 + np.exp(-X @ beta))).astype(np.float32)
    return dict(X=X.astype(np.float32), y=y, beta_true=beta.astype(np.float32))




def gmm_program(D=16, C=8, rng=A.RNG_FLAT, seed=0):
    """z ~ categorical(logits); x ~ mv_normal_diag(mu[z], sigma[z]); y ~ mv_normal_diag(x, r) observed."""
    g = gmm_problem(C=C, D=D, seed=seed)
    sl = SiteList()
    sl.add("z", A.CATEGORICAL_LOGITS, [g["logits"]])
    sl.add("x", A.MVNORMAL_DIAG, [Param.gather(g["mu"], "z"), Param.gather(g["sigma"], "z")], dim=D)
    sl.add("y", A.MVNORMAL_DIAG, [Param.value("x", D), Param.const(g["r"])], dim=D)
    return PackedProgram(sl, {"y": A.MODE_OBS_TAB}, {"y": g["y"]}, rng_mode=rng), g


def logreg_program(N=1024, P=16, rng=A.RNG_FLAT, seed=0):
    """log_tau ~ N(0,1); beta ~ N(0, exp(log_tau)); y ~ bernoulli(logits = X beta); every site constrained
    (an HMC move keeps the values in choices[][]), log_tau and beta selected."""
    pr = logreg_problem(N, P, seed)
    sl = SiteList()
    sl.add("log_tau", A.NORMAL, [0.0, 1.0])
    sl.add("beta", A.NORMAL, [Param.const(0.0), Param.value("log_tau", xf=A.XF_EXP)], dim=P)
    sl.add("y", A.BERNOULLI_LOGITS, [Param.affine(pr["X"], "beta")], dim=N)
    modes = {"y": A.MODE_OBS_TAB, "log_tau": A.MODE_OBS_SLOT, "beta": A.MODE_OBS_SLOT}
    return PackedProgram(sl, modes, {"y": pr["y"]}, selected=("log_tau", "beta"), rng_mode=rng), pr


def logreg_importance_program(N=1024, P=16, rng=A.RNG_FLAT, seed=0):
    """The config-5 model as an ImportanceK target: log_tau and beta sampled from the prior, y observed (prior as
    proposal; the likelihood X beta is a [N x P] contraction per particle)."""
    pr = logreg_problem(N, P, seed)
    sl = SiteList()
    sl.add("log_tau", A.NORMAL, [0.0, 1.0])
    sl.add("beta", A.NORMAL, [Param.const(0.0), Param.value("log_tau", xf=A.XF_EXP)], dim=P)
    sl.add("y", A.BERNOULLI_LOGITS, [Param.affine(pr["X"], "beta")], dim=N)
    return PackedProgram(sl, {"y": A.MODE_OBS_TAB}, {"y": pr["y"]}, rng_mode=rng), pr


def lgssm_scan(dx: int = 8, T: int = 256):
    """config 3's model written as a user would: ``@gen`` step + ``.scan`` (the generic filter's input).
    -> (Scan combinator, initial carry, problem dict)"""
    import genjax_amd as genjax
    s = ssm_problem(dx=dx, T=T)
    Am, q, r = np.asarray(s["A"], np.float32), float(s["q"]), float(s["r"])

    @genjax.gen
    def step(x_prev, _):
        x = genjax.mv_normal_diag(Am @ x_prev, np.full(dx, q, np.float32)) @ "x"
        genjax.mv_normal_diag(x, np.full(dx, r, np.float32)) @ "y"
        return x, None

    return step.scan(n=T), np.zeros(dx, np.float32), s


def lgssm_scan_step_program(dx: int = 8):
    """the one-step program (step 1 of 4) the generic filter generates its kernels from: build steps pre-compile it"""
    from . import C
    from .inference.scan_filter import ScanBootstrapFilter
    scan, carry0, s = lgssm_scan(dx, 4)
    bf = ScanBootstrapFilter(scan, 1024)
    return bf.step_programs(C["y"].set(np.asarray(s["y"], np.float32)[:4]), (carry0, None))[1]

"""Generates the committed golden fixtures (run once in the build container: python tests/golden/make_golden.py).

The reference cannot be imported here (Python 3.10 < its required 3.11; jax / tensorflow-probability
absent), so its outputs cannot be recorded.  The fixtures are instead:
  * threefry_kat.json     Random123 known-answer vectors for Threefry-2x32-20 (the generator of jax.random)
  * logpdf_table.json     float64 log-densities from scipy.stats for every in-scope distribution on a grid
                          that includes the edge cases (tiny scale, p at 0/1, support edges)
  * erfinv_table.json     scipy.special.erfinv on a grid (pins the Giles polynomial)
  * closed_form.json      closed-form answers of the reference's own test cases
                          (tests/inference/test_smc.py:32-87, README.md:89-123) and of BASELINE configs 2-3
"""
import json
import math
import os

import numpy as np
import scipy.special as sp
import scipy.stats as st

HERE = os.path.dirname(os.path.abspath(__file__))


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)


def main():
    # Random123 kat_vectors, threefry2x32 20 rounds: key, counter -> output
    dump("threefry_kat.json", [
        dict(key=[0x00000000, 0x00000000], ctr=[0x00000000, 0x00000000], out=[0x6b200159, 0x99ba4efe]),
        dict(key=[0xffffffff, 0xffffffff], ctr=[0xffffffff, 0xffffffff], out=[0x1cb996fc, 0xbb002be7]),
        dict(key=[0x13198a2e, 0x03707344], ctr=[0x243f6a88, 0x85a308d3], out=[0xc4923a9c, 0x483df7a0]),
    ])

    rows = []
    f32 = lambda v: float(np.float32(v))   # the kernels see float32 inputs: evaluate the truth at exactly those
    def add(kind, x, a, b, lp):
        rows.append(dict(kind=kind, x=float(x), a=float(a), b=float(b), lp=(None if not np.isfinite(lp) else float(lp)),
                         neg_inf=bool(np.isneginf(lp))))
    for mu, sd in [(0, 1), (3.0, 0.01), (-2.5, 7.0), (100.0, 0.5), (0.0, 1e4)]:
        for x in [mu - 4 * sd, mu - sd, mu, mu + 0.3 * sd, mu + 5 * sd]:
            x, m_, s_ = f32(x), f32(mu), f32(sd)
            add("normal", x, m_, s_, st.norm.logpdf(x, m_, s_))
    for p in [0.5, 0.7, 0.001, 0.999, 1.0, 0.0]:
        for x in [0.0, 1.0]:
            lp = math.log(p) if x == 1.0 and p > 0 else (math.log1p(-p) if x == 0.0 and p < 1 else -np.inf)
            add("flip", x, p, 0, lp)
    for l in [-30.0, -2.0, 0.0, 0.3, 25.0]:
        for x in [0.0, 1.0]:
            add("bernoulli_logits", x, l, 0, -np.logaddexp(0.0, -l) if x == 1.0 else -np.logaddexp(0.0, l))
    for a, b in [(2, 2), (0.5, 0.5), (1, 1), (5, 1.5), (30, 70)]:
        for x in [0.01, 0.3, 0.5, 0.9, 0.999]:
            add("beta", x, a, b, st.beta.logpdf(x, a, b))
    for lo, hi in [(0, 1), (-3, 5)]:
        for x in [lo - 0.1, lo, (lo + hi) / 2, hi, hi + 0.1]:
            add("uniform", x, lo, hi, st.uniform.logpdf(x, lo, hi - lo))
    for rate in [0.1, 1.0, 30.0]:
        for x in [-0.5, 0.0, 0.2, 10.0]:
            add("exponential", x, rate, 0, st.expon.logpdf(x, scale=1 / rate))
    for s in [0.2, 1.0, 9.0]:
        for x in [-0.1, 0.0, 0.5, 20.0]:
            add("half_normal", x, s, 0, st.halfnorm.logpdf(x, scale=s))
    for mu, b in [(0, 1), (2, 0.1), (-1, 5)]:
        for x in [mu - 3 * b, mu, mu + 0.5 * b]:
            add("laplace", x, mu, b, st.laplace.logpdf(x, mu, b))
    for mu, s in [(0, 1), (1, 0.25)]:
        for x in [0.01, 0.5, 1.0, 7.0]:
            add("log_normal", x, mu, s, st.lognorm.logpdf(x, s, scale=math.exp(mu)))
    for mu, s in [(0, 1), (3, 0.5)]:
        for x in [mu - 10 * s, mu, mu + s]:
            add("cauchy", x, mu, s, st.cauchy.logpdf(x, mu, s))
    for a, rate in [(0.5, 1), (1, 2), (3, 0.5), (20, 4)]:
        for x in [0.01, 0.7, 3.0, 12.0]:
            add("gamma", x, a, rate, st.gamma.logpdf(x, a, scale=1 / rate))
    # ---- wider set: three/four-parameter forms carry "c", "d" ----
    def add4(kind, x, a, b, c, d, lp):
        rows.append(dict(kind=kind, x=float(x), a=float(a), b=float(b), c=float(c), d=float(d),
                         lp=(None if not np.isfinite(lp) else float(lp)), neg_inf=bool(np.isneginf(lp))))
    for df, loc, sc in [(1.0, 0.0, 1.0), (3.5, -1.0, 2.0), (30.0, 2.0, 0.1)]:
        for x in [loc - 6 * sc, loc - sc, loc, loc + 0.4 * sc, loc + 20 * sc]:
            x, df_, l_, s_ = f32(x), f32(df), f32(loc), f32(sc)
            add4("student_t", x, df_, l_, s_, 0, st.t.logpdf(x, df_, l_, s_))
    for loc, sc, lo, hi in [(0.0, 1.0, -1.0, 2.0), (1.0, 2.0, 2.0, 7.0), (-3.0, 0.5, -10.0, -3.5), (0.0, 1.0, 3.0, 5.0)]:
        for x in [lo - 0.1, lo, 0.5 * (lo + hi), lo + 0.9 * (hi - lo), hi, hi + 0.1]:
            x, l_, s_, lo_, hi_ = f32(x), f32(loc), f32(sc), f32(lo), f32(hi)
            add4("truncated_normal", x, l_, s_, lo_, hi_, st.truncnorm.logpdf(x, (lo_ - l_) / s_, (hi_ - l_) / s_, l_, s_))
    for rate in [0.3, 4.0, 60.0]:
        for x in [-1.0, 0.0, 1.0, 5.0, 70.0, 2.5]:
            lp = st.poisson.logpmf(x, f32(rate)) if x == int(x) else -np.inf
            add("poisson", x, f32(rate), 0, lp)
    for pr in [0.05, 0.5, 0.9]:
        for x in [-1.0, 0.0, 1.0, 7.0, 40.0]:
            add("geometric", x, f32(pr), 0, st.geom.logpmf(x + 1, f32(pr)))        # scipy counts trials, TFP failures
    for loc, sc in [(0.0, 1.0), (2.0, 0.3)]:
        for x in [loc - 2 * sc, loc, loc + 5 * sc]:
            add("gumbel", f32(x), f32(loc), f32(sc), st.gumbel_r.logpdf(f32(x), f32(loc), f32(sc)))
    for loc, sc in [(0.0, 1.0), (1.0, 4.0)]:
        for x in [loc - 0.5, loc, loc + 0.2 * sc, loc + 30 * sc]:
            add("half_cauchy", f32(x), f32(loc), f32(sc), st.halfcauchy.logpdf(f32(x), f32(loc), f32(sc)))
    for a, b in [(0.7, 1.0), (3.0, 2.0), (12.0, 0.5)]:
        for x in [0.02, 0.5, 3.0]:
            add("inverse_gamma", f32(x), f32(a), f32(b), st.invgamma.logpdf(f32(x), f32(a), scale=f32(b)))
    for k, lam in [(0.8, 1.0), (1.0, 2.0), (3.5, 0.5)]:
        for x in [-0.1, 0.05, 0.7, 4.0]:
            add("weibull", f32(x), f32(k), f32(lam), st.weibull_min.logpdf(f32(x), f32(k), scale=f32(lam)))
    for mu, sd in [(0.0, 1.0), (1.5, 0.4)]:
        for x in [0.001, 0.3, 0.5, 0.97]:
            x_ = float(f32(x))
            lg = math.log(x_) - math.log1p(-x_)
            add("logit_normal", x_, f32(mu), f32(sd), st.norm.logpdf(lg, f32(mu), f32(sd)) - math.log(x_) - math.log1p(-x_))
    for df in [1.0, 2.0, 7.5, 40.0]:
        for x in [0.05, 1.0, 9.0, 60.0]:
            add("chi2", f32(x), f32(df), 0, st.chi2.logpdf(f32(x), f32(df)))
    dump("logpdf_table.json", rows)

    dr = []
    rng_d = np.random.default_rng(5)
    for alpha in [[1.0, 1.0, 1.0], [0.5, 2.0, 3.0, 0.7], [10.0, 20.0], [2.0] * 8]:
        al = np.asarray(alpha, np.float32).astype(np.float64)
        for _ in range(3):
            x = rng_d.dirichlet(al)
            x = x.astype(np.float32).astype(np.float64)
            x = x / x.sum()                                 # scipy insists on the simplex; the kernels see float32(x)
            dr.append(dict(alpha=al.tolist(), x=x.tolist(), lp=float(st.dirichlet.logpdf(x, al))))
    dump("dirichlet_table.json", dr)

    cat = []
    rng = np.random.default_rng(0)
    for n in [2, 3, 8, 17]:
        logits = rng.standard_normal(n) * 2
        lsm = logits - sp.logsumexp(logits)
        cat.append(dict(logits=logits.tolist(), log_softmax=lsm.tolist()))
    dump("categorical_table.json", cat)

    xs = np.concatenate([np.linspace(-0.999999, 0.999999, 41), [1e-8, -1e-8, 0.5, 0.9, 0.99, 0.9999, 0.999999]])
    xs = xs.astype(np.float32).astype(np.float64)   # exactly representable float32 inputs
    dump("erfinv_table.json", dict(x=xs.tolist(), y=sp.erfinv(xs).tolist()))

    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import closed_form as cf
    g = cf.gmm_problem()
    s = cf.ssm_problem()
    kl, incs, means = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"])
    dump("closed_form.json", dict(
        flip_flip_trivial=math.log(0.7),                       # test_smc.py:32-57
        flip_flip=math.log(0.5 * 0.9 + 0.5 * 0.3),             # test_smc.py:59-87
        beta_bernoulli_true=math.log(0.5), beta_bernoulli_false=math.log(0.5),   # README.md:89-102
        beta_bernoulli_post_mean_true=0.6, beta_bernoulli_post_mean_false=0.4,
        readme_printed=[0.6039314, 0.3679334],                  # README.md:121-123 (stream dependent; informational)
        gmm_c8_d16_seed0=cf.gmm_log_ml(**g),
        ssm_dx8_T256_seed0=kl,
        ssm_increments_head=incs[:8].tolist(),
        ssm_filtered_mean_last=means[-1].tolist(),
    ))
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()

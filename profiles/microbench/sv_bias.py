"""Stochastic-volatility filter: is the device's log-ML estimate biased against the ideal float64 filter?  (two single
runs sat at z = +2.8 and +3.7.)  N device seeds x {FLAT, JAX32} x {0, 2 Metropolis moves per step}, and the multinomial resampler, against the 256-seed float64
fixture (tests/golden/sv_pf_float64.json): mean difference +- standard error, and the ratio of the spreads.
usage: python profiles/microbench/sv_bias.py [n_seeds=64]"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import genjax_amd as genjax                      # noqa: E402
from genjax_amd import C                         # noqa: E402
from genjax_amd import _abi as A                 # noqa: E402
from genjax_amd.inference import BootstrapFilter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "sv_pf_float64.json")))
phi, sigma, ys = fx["phi"], fx["sigma"], np.asarray(fx["y"], np.float32)
ref = np.asarray(fx["log_ml"], np.float64)
T, K = len(ys), 1 << 18


@genjax.gen
def step(x_prev, _):
    x = genjax.normal(phi * x_prev, sigma) @ "x"
    genjax.normal(0.0, genjax.exp(0.5 * x)) @ "y"
    return x, None


print(f"float64 filter: {ref.size} seeds, mean {ref.mean():.5f}, sd {ref.std(ddof=1):.5f}, se {ref.std(ddof=1) / math.sqrt(ref.size):.5f}")
out = {}
for rng, rn in ((A.RNG_FLAT, "flat"), (A.RNG_JAX32, "jax32")):
    for moves in (0, 2):
        bf = BootstrapFilter(step.scan(n=T), K, rng_mode=rng, rejuvenate=dict(n_moves=moves, scale=0.5) if moves else None)
        est = np.array([float(bf.run(genjax.key(1000 + i), C["y"].set(ys), (0.0, None))["log_ml"]) for i in range(n)])
        d = est.mean() - ref.mean()
        se = math.sqrt(est.var(ddof=1) / est.size + ref.var(ddof=1) / ref.size)
        out[f"{rn}_moves{moves}"] = dict(mean=float(est.mean()), sd=float(est.std(ddof=1)), bias=float(d), se=float(se), bias_over_se=float(d / se),
                                         spread_ratio=float(est.std(ddof=1) / ref.std(ddof=1)), form=bf.last_info["form_name"])
        print(f"{rn:6s} moves={moves}: mean {est.mean():.5f} sd {est.std(ddof=1):.5f}  bias {d:+.5f} +- {se:.5f} ({d / se:+.2f} SE)  spread ratio {est.std(ddof=1) / ref.std(ddof=1):.2f}  [{bf.last_info['form_name']}]")
for rng, rn in ((A.RNG_FLAT, "flat"), (A.RNG_JAX32, "jax32")):
    bf = BootstrapFilter(step.scan(n=T), K, rng_mode=rng, resampler="multinomial")
    est = np.array([float(bf.run(genjax.key(1000 + i), C["y"].set(ys), (0.0, None))["log_ml"]) for i in range(n)])
    d = est.mean() - ref.mean()
    se = math.sqrt(est.var(ddof=1) / est.size + ref.var(ddof=1) / ref.size)
    out[f"{rn}_multinomial"] = dict(mean=float(est.mean()), sd=float(est.std(ddof=1)), bias=float(d), se=float(se), bias_over_se=float(d / se),
                                    spread_ratio=float(est.std(ddof=1) / ref.std(ddof=1)), form=bf.last_info["form_name"])
    print(f"{rn:6s} multinomial: mean {est.mean():.5f} sd {est.std(ddof=1):.5f}  bias {d:+.5f} +- {se:.5f} ({d / se:+.2f} SE)  spread ratio {est.std(ddof=1) / ref.std(ddof=1):.2f}  [{bf.last_info['form_name']}]")
print(json.dumps(out))

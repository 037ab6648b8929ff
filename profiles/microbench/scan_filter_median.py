"""the generic filter (one-launch kernel) timed run by run, median per step: config 3's model at K = 2^18 / 2^20, the stochastic-volatility model with and
without moves; prints the log-ML estimates too (an emitter change must not move them).  Used for the A/Bs of DESIGN.md §9."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import genjax_amd as genjax
from genjax_amd import C, workloads
from genjax_amd.inference import BootstrapFilter
def med(bf, chm, args, n):
    for i in range(2): bf.run(genjax.key(i), chm, args)
    ts = []
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = bf.run(genjax.key(10 + i), chm, args)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), o
res = {}
scan, carry0, s = workloads.lgssm_scan(8, 256)
ys = np.asarray(s["y"], np.float32)
for K in (1 << 18, 1 << 20):
    bf = BootstrapFilter(scan, K); bf.alias_outputs = True
    dt, o = med(bf, C["y"].set(ys), (carry0, None), 9 if K == 1 << 18 else 5)
    res[f"lgssm_K{K}"] = (round(dt / 256 * 1e6, 2), float(o["log_ml"]), o["info"].get("form"))
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "sv_pf_float64.json")))
phi, sigma, ysv = fx["phi"], fx["sigma"], np.asarray(fx["y"], np.float32)
@genjax.gen
def sv_step(x_prev, _):
    x = genjax.normal(phi * x_prev, sigma) @ "x"
    genjax.normal(0.0, genjax.exp(0.5 * x)) @ "y"
    return x, None
for mv in (0, 2):
    bf = BootstrapFilter(sv_step.scan(n=len(ysv)), 1 << 18, **({"rejuvenate": dict(n_moves=2, scale=0.3)} if mv else {})); bf.alias_outputs = True
    dt, o = med(bf, C["y"].set(ysv), (0.0, None), 7)
    res[f"sv_moves{mv}"] = (round(dt / len(ysv) * 1e6, 2), float(o["log_ml"]), o["info"].get("form"))
print(json.dumps(res))

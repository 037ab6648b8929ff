"""round 6: the generic filter with the MULTINOMIAL resampler inside the one-launch kernel (config 3's model, K = 2^18, T = 256) and with
an HMC move behind every resampling inside the library's step loop (stochastic volatility, K = 2^16, T = 256), a few runs each
(for rocprofv3 --kernel-trace --stats: profiles/r06_moves_kernel_stats.csv)"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import genjax_amd as genjax
from genjax_amd import C, S, workloads
from genjax_amd.inference import BootstrapFilter, HMC

scan, carry0, s = workloads.lgssm_scan(8, 256)
ys = np.asarray(s["y"], np.float32)
fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "sv_pf_float64.json")))
phi, sigma, ysv = fx["phi"], fx["sigma"], np.asarray(fx["y"], np.float32)


@genjax.gen
def sv(x_prev, _):
    x = genjax.normal(phi * x_prev, sigma) @ "x"
    genjax.normal(0.0, genjax.exp(0.5 * x)) @ "y"
    return x, None


for name, bf, chm, args, Tn in (("multinomial in the kernel, K = 2^18", BootstrapFilter(scan, 1 << 18, resampler="multinomial"), C["y"].set(ys), (carry0, None), 256),
                                ("HMC move in the step loop, K = 2^16", BootstrapFilter(sv.scan(n=len(ysv)), 1 << 16, moves=[HMC(S["x"], 0.25, 3)]), C["y"].set(ysv), (0.0, None), len(ysv))):
    bf.run(genjax.key(1), chm, args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4):
        o = bf.run(genjax.key(2 + i), chm, args)
    torch.cuda.synchronize()
    print(name, ": us per step", (time.perf_counter() - t0) / (4 * Tn) * 1e6, "log_ml", float(o["log_ml"]), o["info"]["form_name"])

"""Odd sizes through the main entry points (crash / hang / parity smoke; run under `timeout`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import helpers as H
from genjax_amd import kernels as K_, core, workloads, _abi as A
from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
from oracle import cpu as oracle
oracle.build()
t0 = time.time()
def say(*a): print("%6.1fs" % (time.time() - t0), *a, flush=True)
# filter at tiny and beyond-fused sizes
s = workloads.ssm_problem(T=6)
for K in (1, 5, 257, (1 << 22) + 3):
    bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), K)
    a = bf.run(core.key(1), s["y"]); b = bf.run(core.key(1), s["y"], step_by_step=True)
    torch.cuda.synchronize()
    assert torch.equal(a["x"], b["x"]), K
    say("filter K", K, float(a["log_ml"]))
# generic program, many slots, K edge sizes
prog = H.zoo2()
for K in (1, 63, 65, 1 << 21):
    g = K_.run_program(prog, (1, 2), K); torch.cuda.synchronize()
    if K <= 65:
        o = oracle.run_program(prog, (1, 2), K)
        bad = (~np.isclose(g["score"].cpu().numpy(), o["score"], rtol=1e-3, atol=1e-3)).sum()
        assert bad <= 1, (K, bad)
    say("zoo2 K", K, float(g["lse"][3]))
# fused gmm at sizes around the vector width and with an offset near 2^32
gp, _ = workloads.gmm_program()
for K, off in ((1, 0), (3, 0), (4, 0), (1023, 7), (1 << 20, (1 << 32) - 1000), (1 << 20, (1 << 33) + 5)):
    g = K_.run_program(gp, (3, 4), K, offset=off, K_total=max(K, 1)); torch.cuda.synchronize()
    say("gmm K", K, "off", off, float(g["lse"][3]))
# pick / logsumexp on big arrays
x = torch.randn(1 << 25, device="cuda")
l = K_.logsumexp(x); p = K_.categorical_pick(x, l, (1, 2)); torch.cuda.synchronize()
say("lse/pick 2^25", float(l[2]), int(p[1]))
# hmc generic single chain
lp, pr = H.logreg(N=20, P=4)
ch = torch.zeros((5, 1), device="cuda")
out = K_.hmc(lp, (1, 2), ch, 0.01, 5, False, True); torch.cuda.synchronize()
say("hmc n=1", float(out["alpha"][0]))
say("all ok")

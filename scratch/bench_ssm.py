import os, sys, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from genjax_amd import core, kernels, _abi as A
from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
from oracle import closed_form as cf
s = cf.ssm_problem()
exact, _, _ = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"])
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), K)
for mode in ("native", "python"):
    kw = dict(step_by_step=(mode == "python"))
    out = bf.run(core.key(1), s["y"], **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 5
    for i in range(n):
        out = bf.run(core.key(2 + i), s["y"], **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    lml = float(out["log_ml"])
    print(f"{mode}: K=2^{int(np.log2(K))} T=256: {dt*1e3:.2f} ms/run  {K*256/dt:.3e} particle-steps/s  {dt/256*1e6:.1f} us/step  log-ML {lml:.3f} exact {exact:.3f} rel {abs(lml-exact)/abs(exact):.2e}")

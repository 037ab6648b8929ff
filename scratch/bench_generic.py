import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from genjax_amd import kernels
import helpers as H
os.environ["GJX_FORCE_GENERIC"] = "1"
K = 1 << 20
for name, prog in (("gmm", H.gmm()[0]), ("zoo", H.zoo()), ("betab", H.beta_bernoulli(True))):
    for nolds in ("1", "0"):
        os.environ["GJX_GENERIC_NO_LDS"] = nolds
        out = kernels.run_program(prog, (0, 1), K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10):
            kernels.run_program(prog, (0, 1 + i), K, out=out, ws=out["_ws"])
        e1.record(); torch.cuda.synchronize()
        print(name, "no_lds" if nolds == "1" else "lds", round(e0.elapsed_time(e1) / 10 * 1e3, 1), "us")

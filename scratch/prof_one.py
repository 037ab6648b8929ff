import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genjax_amd import _abi as A, kernels as Kn
from genjax_amd.program import SiteList, Param, PackedProgram
from oracle import closed_form as cf
g = cf.gmm_problem()
sl = SiteList()
sl.add("z", A.CATEGORICAL_LOGITS, [g["logits"]])
sl.add("x", A.MVNORMAL_DIAG, [Param.gather(g["mu"], "z"), Param.gather(g["sigma"], "z")])
sl.add("y", A.MVNORMAL_DIAG, [Param.value("x", 16), Param.const(g["r"])])
prog = PackedProgram(sl, {"y": A.MODE_OBS_TAB}, {"y": g["y"]})
K = 1 << 20
out = Kn.run_program(prog, (0, 1), K)
for _ in range(int(os.environ.get("N", "5"))):
    Kn.run_program(prog, (0, 1), K, out=out, ws=out["_ws"])
torch.cuda.synchronize()

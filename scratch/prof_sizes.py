import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from genjax_amd import _abi as A, kernels
import helpers as H
prog, g = H.gmm()
for K in (1 << 10, 1 << 14, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21):
    for lse in (True, False):
        out = kernels.run_program(prog, (0, 1), K, want_lse=lse, want_weight=False)
        for i in range(10):
            kernels.run_program(prog, (0, 1 + i), K, out=out, ws=out["_ws"], want_lse=lse, want_weight=False)
        torch.cuda.synchronize()

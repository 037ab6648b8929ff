"""Kernel durations of the native filter at several K (run under rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genjax_amd import core, workloads
from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
s = workloads.ssm_problem(T=64)
for lg in (14, 16, 18, 20):
    bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), 1 << lg)
    for rep in range(3):
        out = bf.run(core.key(rep), torch.as_tensor(s["y"], device="cuda"))
    torch.cuda.synchronize()
    print(lg, float(out["log_ml"]))

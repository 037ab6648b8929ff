import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genjax_amd import core, workloads
from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
s = workloads.ssm_problem()
bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), int(os.environ.get("SSM_K", 1 << 18)))
ys = torch.as_tensor(s["y"]).cuda()
bf.run(core.key(1), ys)
tl = torch.zeros((4096, 8), dtype=torch.int64, device="cuda")
from genjax_amd._lib import load as _load
import ctypes as _C
_load().gjx_debug_timeline(_C.c_void_p(tl.data_ptr()), tl.numel() * tl.element_size())
bf.run(core.key(2), ys)
torch.cuda.synchronize()
t = tl.cpu().numpy().astype(np.float64)
t = t[t[:, 0] > 0]          # blocks that ran (K / 256)
t0 = t[:, 0].min()
names = [(0,"start"), (2,"tile total published"), (3,"totals gathered+prefix"), (4,"tile list known"), (5,"ancestors known"), (6,"x_prev gathered+A x"), (7,"noise drawn, x stored"), (1,"end")]
for j, n in names:
    c = (t[:, j] - t0) * 0.01
    print(f"{n:24s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")

import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genjax_amd import core, workloads
from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
s = workloads.ssm_problem()
bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), int(os.environ.get("SSM_K", 1 << 18)))
ys = torch.as_tensor(s["y"]).cuda()
bf.run(core.key(1), ys)
tl = torch.zeros((4096, 8), dtype=torch.int64, device="cuda")
os.environ["GJX_STEP_TIMELINE_PTR"] = hex(tl.data_ptr())
bf.run(core.key(2), ys)
torch.cuda.synchronize()
t = tl.cpu().numpy().astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print("persistent filter, step T/2, %d blocks" % len(t))
for j, n in [(0, "step start"), (1, "block max published"), (2, "global max known"), (3, "tile total published"), (4, "totals gathered + prefix"), (5, "ancestors known"), (6, "step end (stores issued)")]:
    c = (t[:, j] - t0) * 0.01
    print(f"{n:26s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")

import os, sys, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from genjax_amd import kernels
import helpers as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
prog, pr = H.logreg(N=1024, P=16)
print("engine", kernels.hmc_engine(prog))
ch = torch.as_tensor((np.random.default_rng(0).standard_normal((17, n)) * 0.1).astype(np.float32)).cuda()
out = kernels.hmc(prog, (1, 1), ch.clone(), 0.01, 10, False, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = kernels.hmc(prog, (1, 2), ch.clone(), 0.01, L, False, True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fl = 2 * 2 * 1024 * 16 + 10 * 1024
print(f"chains {n} L {L}: {dt*1e3:.2f} ms  {n*L/dt:.3e} chain-leapfrogs/s  {n*L*fl/dt/1e12:.1f} TFLOP/s  accept {float(out['accepted'].mean()):.3f} mean|alpha| {float(out['alpha'].abs().mean()):.4f}")

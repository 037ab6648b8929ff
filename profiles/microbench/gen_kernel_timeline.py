"""per-block phase stamps (s_memrealtime, 100 MHz) of ONE launch of a generated propagate + reweight kernel (gjx_gen) — the mixture of
config 2 at K = 2^20 by default: start / table + derived constants in LDS / sites of the block's last tile done / end — min, median,
max over the blocks, us from the first block's start.  ENG=gen|auto is ignored: the stamps exist in generated kernels only."""
import ctypes as C_, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from genjax_amd import kernels, workloads, _abi as A
from genjax_amd._lib import load
K = int(os.environ.get("KK", 1 << 20))
gmm, _ = workloads.gmm_program(D=16, C=8)
os.environ["GJX_ENGINE"] = "gen"
ws = kernels.workspace(A.OP_RUN, K, "cuda")
out = kernels.run_program(gmm, (0, 1), K, ws=ws, want_weight=False)
for i in range(20):
    kernels.run_program(gmm, (0, 2 + i), K, ws=ws, out=out, want_weight=False)
nb = 4096
tl = torch.zeros((nb, 16), dtype=torch.int64, device="cuda")
load().gjx_debug_timeline(C_.c_void_p(tl.data_ptr()), tl.numel() * tl.element_size())
kernels.run_program(gmm, (0, 99), K, ws=ws, out=out, want_weight=False)
torch.cuda.synchronize()
load().gjx_debug_timeline(None, 0)
t = tl.cpu().numpy().astype(np.float64)
used = t[:, 0] > 0
t = t[used]
t0 = t[:, 0].min()
print(f"generated mixture kernel, K = {K}, {used.sum()} blocks")
for j, n in [(0, "block start"), (1, "table + derived constants in LDS"), (3, "sites of the last tile done"), (7, "block end (LSE pair out)")]:
    c = (t[:, j] - t0) * 0.01
    print(f"{n:36s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
print("prologue per block (median): %.2f us; sites (median, all tiles of a block): %.2f us" % (np.median(t[:, 1] - t[:, 0]) * 0.01, np.median(t[:, 3] - t[:, 1]) * 0.01))

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genjax_amd import kernels as K_, distributed as D
K, R, G, u = 50_001, 4, 4, 0.37
rs = np.random.default_rng(11)
lw = rs.standard_normal(K).astype(np.float32); lw[K // 7:] -= 60.0
rows = rs.standard_normal((R, K)).astype(np.float32)
lwd = torch.as_tensor(lw).cuda()
lse = K_.logsumexp(lwd, K)
anc_ref = K_.resample_indices(lwd, u, K, lse=lse).cpu().numpy()
shards = [D.shard(K, r, G) for r in range(G)]
cums, tots = [], []
for off, k in shards:
    cum, bt = K_.weight_cumsum(lwd[off:off + k].contiguous(), True, lse)
    cums.append(cum.clone()); tots.append(int(bt.cpu()[1]))
totals = torch.tensor(tots, dtype=torch.int64).cuda()
print("tots", tots)
for r, (off, k) in enumerate(shards[:1]):
    plan = K_.ShardPlan("cuda").build(totals, r, u, K)
    p = plan.wait()
    print("plan", p.slot0, p.n_valid, p.keep_lo, p.keep_hi, list(p.bounds[:G+1]))
    src = torch.as_tensor(rows[:, off:off + k].copy()).cuda()
    for trial in range(3):
        anc = torch.full((K,), -7, dtype=torch.int32, device="cuda")
        anc, kept = K_.shard_resample(cums[r], plan, u, K, src, p.own_n, anc=anc)
        a = anc.cpu().numpy()[: p.n_valid]
        want = anc_ref[p.slot0: p.slot0 + p.n_valid] - off
        bad = np.nonzero(a != want)[0]
        print("trial", trial, "bad", bad.size, bad[:5], bad[-5:], a[bad[:5]] if bad.size else None, want[bad[:5]] if bad.size else None)
    counts = np.bincount(want, minlength=k)
    print("max children", counts.max(), "argmax", counts.argmax(), "heavy(>8):", (counts > 8).sum())

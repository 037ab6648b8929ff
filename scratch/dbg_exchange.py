import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.multiprocessing as mp, torch.distributed as dist

def worker(rank, world, port, K, R, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GJX_DIST_BACKEND="gloo")
    from genjax_amd import distributed as D, kernels
    D.init_from_env("gloo")
    rs = np.random.default_rng(11)
    lw = rs.standard_normal(K).astype(np.float32); lw[K // 7:] -= 60.0
    rows = rs.standard_normal((R, K)).astype(np.float32)
    off, k = D.shard(K, rank, world)
    lw_d = torch.as_tensor(lw[off:off + k]).cuda()
    local = kernels.logsumexp(lw_d, K)
    new_rows, info = D.resample_exchange(torch.as_tensor(rows[:, off:off + k].copy()).cuda(), lw_d, None, 0.37, K, pairs=D.gather_lse_pairs(local))
    q.put((rank, new_rows.cpu().numpy(), info["sent"], info["bounds"], info["ancestors"].cpu().numpy()[:8]))
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    K, R, world = 50_001, 4, 4
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, 29911, K, R, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    [p.join() for p in ps]
    from genjax_amd import kernels, distributed as D
    rs = np.random.default_rng(11)
    lw = rs.standard_normal(K).astype(np.float32); lw[K // 7:] -= 60.0
    rows = rs.standard_normal((R, K)).astype(np.float32)
    lw_d = torch.as_tensor(lw).cuda()
    anc = kernels.resample_indices(lw_d, 0.37, K, lse=kernels.logsumexp(lw_d, K))
    want = kernels.gather_rows(torch.as_tensor(rows).cuda(), anc).cpu().numpy()
    got = np.concatenate([r[1] for r in res], axis=1)
    bad = np.nonzero((got != want).any(axis=0))[0]
    print("bounds", res[0][3], "sent", [r[2] for r in res])
    print("n bad", bad.size, "first/last bad", bad[:5], bad[-5:])
    for r in range(world):
        lo, k = D.shard(K, r, world)
        b = bad[(bad >= lo) & (bad < lo + k)]
        print("rank", r, "owns", lo, lo + k, "bad", b.size, (b.min(), b.max()) if b.size else None)
    j = bad[0] if bad.size else 0
    print("slot", j, "got", got[:, j], "want", want[:, j], "anc", int(anc[j]))

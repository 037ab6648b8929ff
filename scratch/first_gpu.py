import os, sys, time, numpy as np, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genjax_amd import _abi as A, kernels as Kn
from genjax_amd.program import SiteList, Param, PackedProgram
from oracle import cpu, closed_form as cf

dev = "cuda"
# 1. threefry
t = Kn.threefry2x32((0x13198a2e, 0x03707344), 4, ctr_lo0=0x85a308d3, ctr_hi=0x243f6a88).cpu().numpy().view(np.uint32)
print("threefry", [hex(x) for x in t[0]], "expect c4923a9c 483df7a0")
ref = np.array([cpu.threefry2x32(0x13198a2e, 0x03707344, 0x243f6a88, 0x85a308d3 + i) for i in range(4)], np.uint32)
assert (t == ref).all()

def gmm_prog(D=16, C=8, rng=A.RNG_FLAT):
    g = cf.gmm_problem(C=C, D=D)
    sl = SiteList()
    sl.add("z", A.CATEGORICAL_LOGITS, [g["logits"]])
    sl.add("x", A.MVNORMAL_DIAG, [Param.gather(g["mu"], "z"), Param.gather(g["sigma"], "z")])
    sl.add("y", A.MVNORMAL_DIAG, [Param.value("x", D), Param.const(g["r"])])
    return PackedProgram(sl, {"y": A.MODE_OBS_TAB}, {"y": g["y"]}, rng_mode=rng), g

def cmp(name, a, b, rtol=1e-4, atol=1e-4):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b) / (atol + rtol * np.abs(b))
    print(f"  {name}: max err ratio {err.max():.3g}  maxabs {np.abs(a-b).max():.3g}")
    return err.max()

for rng in (A.RNG_FLAT, A.RNG_JAX32):
  for D in (16, 1):
    prog, g = gmm_prog(D=D, rng=rng)
    K = 10007
    o = cpu.run_program(prog, (0, 1), K)
    for force in ("1", "0"):
        os.environ["GJX_FORCE_GENERIC"] = force
        print("rng", rng, "D", D, "engine", Kn.program_engine(prog))
        r = Kn.run_program(prog, (0, 1), K)
        torch.cuda.synchronize()
        zc = r["choices"][0].cpu().numpy(); zo = o["choices"][0]
        print("  z mismatches", int((zc != zo).sum()))
        same = zc == zo
        cmp("x", r["choices"][1:].cpu().numpy()[:, same], o["choices"][1:][:, same])
        cmp("score", r["score"].cpu().numpy()[same], o["score"][same])
        cmp("weight", r["weight"].cpu().numpy()[same], o["weight"][same])
        print("  lse", r["lse"].cpu().numpy(), o["lse"])

# big run + timing
os.environ["GJX_FORCE_GENERIC"] = "0"
prog, g = gmm_prog()
K = 1 << 20
exact = cf.gmm_log_ml(**g)
for ppt in ("1", "2", "4"):
    os.environ["GJX_GMM_PPT"] = ppt
    out = Kn.run_program(prog, (0, 1), K)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50):
        Kn.run_program(prog, (0, 1), K, out=out, ws=out["_ws"])
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 50
    print(f"fused ppt={ppt}: {ms*1e3:.1f} us/call  {76*K/ms/1e9:.2f} TB/s  logML {out['lse'][3].item():.5f} exact {exact:.5f} rel {(out['lse'][3].item()-exact)/abs(exact):.2e}")
os.environ["GJX_FORCE_GENERIC"] = "1"
out = Kn.run_program(prog, (0, 1), K)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    Kn.run_program(prog, (0, 1), K, out=out, ws=out["_ws"])
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 20
print(f"generic: {ms*1e3:.1f} us/call  {76*K/ms/1e9:.2f} TB/s  logML {out['lse'][3].item():.5f}")
os.environ["GJX_FORCE_GENERIC"] = "0"

# lse / pick / resample
lw = out["logw"]
l4 = Kn.logsumexp(lw); torch.cuda.synchronize()
lwc = lw.cpu().numpy()
print("lse gpu", l4.cpu().numpy(), "oracle", cpu.logsumexp(lwc))
pk = Kn.categorical_pick(lw, l4, (5, 6)); torch.cuda.synchronize()
pv = pk.cpu().numpy()
print("pick gpu", pv.view(np.float32)[0], pv[1], "oracle", cpu.categorical_pick(lwc, cpu.logsumexp(lwc), (5, 6)))
w = torch.exp(lw - l4[0])
cum, tot = Kn.weight_cumsum(w)
torch.cuda.synchronize()
cum_o, tot_o = cpu.weight_cumsum(w.cpu().numpy())
print("cumsum exact:", bool((cum.cpu().numpy().view(np.uint64) == cum_o).all()), int(tot.item()), tot_o)
bt = torch.tensor([0, tot_o], dtype=torch.int64, device=dev)
anc = Kn.resample_systematic(cum, bt, 0.37, K); torch.cuda.synchronize()
anc_o = cpu.resample_systematic(cum_o, 0.37, K)
print("systematic exact:", bool((anc.cpu().numpy() == anc_o).all()), anc_o[:8], anc_o[-3:])
anc2 = Kn.resample_multinomial(cum, bt, (7, 8), K); torch.cuda.synchronize()
anc2_o = cpu.resample_multinomial(cum_o, (7, 8), K)
print("multinomial exact:", bool((anc2.cpu().numpy() == anc2_o).all()))
gx = Kn.gather_rows(out["choices"], anc); torch.cuda.synchronize()
print("gather exact:", bool((gx.cpu().numpy() == cpu.gather_rows(out["choices"].cpu().numpy(), anc_o)).all()))
# timing resample pipeline
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    cum, tot = Kn.weight_cumsum(lw, True, l4)
    anc = Kn.resample_systematic(cum, bt, 0.37, K)
    gx = Kn.gather_rows(out["choices"], anc, gx)
ev[1].record(); torch.cuda.synchronize()
print("resample pipeline us", ev[0].elapsed_time(ev[1]) / 20 * 1e3)

# ssm
s = cf.ssm_problem()
Ad = torch.tensor(s["A"], device=dev); yd = torch.tensor(s["y"], device=dev)
ssm = A.GjxSsm(); ssm.dx = 8; ssm.dy = 8; ssm.A_dev = Ad.data_ptr(); ssm.H_dev = None; ssm.q = s["q"]; ssm.r = s["r"]; ssm.q0 = 1.0
K2 = 5000
x0, lw0, l0 = Kn.ssm_step(ssm, (1, 2), 0, 0, K2, None, None, yd[0]); torch.cuda.synchronize()
xo, lwo, lo = cpu.ssm_step(s["A"], None, s["q"], s["r"], 1.0, (1, 2), 0, 0, K2, None, None, s["y"][0])
cmp("ssm x0", x0.cpu().numpy(), xo); cmp("ssm lw0", lw0.cpu().numpy(), lwo); print(l0.cpu().numpy(), lo)
ancr = torch.randint(0, K2, (K2,), dtype=torch.int32, device=dev)
x1, lw1, l1 = Kn.ssm_step(ssm, (3, 4), 0, 1, K2, x0, ancr, yd[1]); torch.cuda.synchronize()
x1o, lw1o, l1o = cpu.ssm_step(s["A"], None, s["q"], s["r"], 1.0, (3, 4), 0, 1, K2, x0.cpu().numpy(), ancr.cpu().numpy(), s["y"][1])
cmp("ssm x1", x1.cpu().numpy(), x1o); cmp("ssm lw1", lw1.cpu().numpy(), lw1o); print(l1.cpu().numpy(), l1o)

"""Static ISA histogram of the headline kernel by issue class (run anywhere hipcc is: no GPU needed).
usage: python profiles/microbench/isa_histogram.py > profiles/r02_gmm_isa_histogram.txt
Classes and their measured issue costs (cycles per wave-instruction with >= 4 waves per SIMD, whole-grid span,
profiles/r02_valu_issue_microbench.txt): fast 2.3 | slow 4.2 | transcendental 8.3."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(ROOT, "genjax_amd", "csrc")
asm = os.path.join(tempfile.mkdtemp(), "run.s")
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", csrc, "-I", os.path.join(ROOT, "include"),
                       "--cuda-device-only", "-S", "-o", asm, os.path.join(csrc, "gjx_run.hip")], stderr=subprocess.DEVNULL)
txt = open(asm).read()
name = [n for n in re.findall(r"^(_ZN3gjx14k_run_gmm_flat\w+):", txt, re.M) if "ILi16ELi4ELi256ELb0ELb0" in n or n.endswith("ILi16ELi4ELi256ELb0EEEvNS_7GmmArgsE")][0]
i = txt.index(name + ":"); j = txt.index("s_endpgm", i)
ops = [l.split()[0] for l in (x.strip() for x in txt[i:j].splitlines()) if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
FAST = {"v_add_u32_e32", "v_xor_b32_e32", "v_sub_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_lshrrev_b32_e32", "v_mov_b32_e32",
        "v_add_f32_e32", "v_sub_f32_e32", "v_mul_f32_e32", "v_fmac_f32_e32", "v_fma_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mov_b64_e32"}
TRANS = {"v_log_f32_e32", "v_exp_f32_e32", "v_sqrt_f32_e32", "v_rcp_f32_e32", "v_sin_f32_e32", "v_cos_f32_e32", "v_rsq_f32_e32"}
cls = collections.Counter(); per = collections.Counter()
for o in ops:
    if o.startswith("v_"):
        c = "fast" if o in FAST else "trans" if o in TRANS else "slow"
        cls[c] += 1; per[(c, o)] += 1
    elif o.startswith("ds_"): cls["lds"] += 1
    elif o.startswith(("global_", "buffer_", "flat_")): cls["vmem"] += 1
    else: cls["salu+ctl"] += 1
PPT = 4
print("kernel: gjx::k_run_gmm_flat<16, 4, 256, false>  (4 particles per lane; counts are per lane = per 4 particles, static,")
print("        whole kernel incl. prologue/epilogue and the never-taken 2^32-particle key branch)")
for c in ("fast", "slow", "trans", "lds", "vmem", "salu+ctl"):
    print(f"  {c:10s} {cls[c]:5d}   per particle {cls[c] / PPT:7.1f}")
valu = cls["fast"] + cls["slow"] + cls["trans"]
cyc_sep = (2.3 * cls["fast"] + 4.2 * cls["slow"] + 8.3 * cls["trans"]) / PPT
cyc_mix = (4.2 * (cls["fast"] + cls["slow"]) + 8.3 * cls["trans"]) / PPT
print(f"  VALU total {valu} = {valu / PPT:.0f} per particle (PMC, dynamic: SQ_INSTS_VALU / SQ_WAVES / 4 in profiles/r02_valu_utilisation.json)")
print(f"  issue cycles per wave-particle if classes issued back to back: {cyc_sep:.0f}; with the measured mixing penalty")
print(f"  (a fast op next to a slow one issues at the slow rate, t_mix_add_align 4.0 vs t_mix_add_xor 2.3): {cyc_mix:.0f}")
print("per opcode:")
for (c, o), n in sorted(per.items(), key=lambda kv: -kv[1])[:24]:
    print(f"  {o:26s} {c:6s} {n:5d}")

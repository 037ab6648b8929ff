"""quick numbers: generated HMC kernel (matrix-core flavour) vs the hand-written one; generic Scan filter one- vs two-launch steps"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
dev = torch.device("cuda:0")
out = {}
for bt in (None, "256", "512", "1024"):
    if bt: os.environ["GJX_HMC_GEN_BT"] = bt
    else: os.environ.pop("GJX_HMC_GEN_BT", None)
    try:
        r = bench.run_hmc_generated(dev)["hier_logreg_N1024_P16_L100"]
        out[f"hmc_gen_bt_{bt}"] = {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items()}) for k, v in r.items()}
    except Exception as e:
        out[f"hmc_gen_bt_{bt}"] = repr(e)
os.environ.pop("GJX_HMC_GEN_BT", None)
os.environ["GJX_HMC_GEN_NO_MFMA"] = "1"
out["hmc_gen_scalar"] = bench.run_hmc_generated(dev)["hier_logreg_N1024_P16_L100"]["generated"]
os.environ.pop("GJX_HMC_GEN_NO_MFMA")
r4 = bench.run_round4(dev)
out["scan_filter_one_launch"] = {k: r4[k] for k in r4 if k.startswith("scan_filter")}
os.environ["GJX_SCAN_FILTER_TWO_LAUNCH"] = "1"
r4 = bench.run_round4(dev)
out["scan_filter_two_launch"] = {k: r4[k] for k in r4 if k.startswith("scan_filter")}
print(json.dumps(out, indent=1, default=str))

import sys, json, os
sys.path.insert(0, os.getcwd())
import torch, bench
r = bench.run_round5(torch.device("cuda:0"))
for k in ("scan_filter_lgssm_multinomial_T256_K2e18", "scan_filter_lgssm_optimal_proposal_T256_K2e18"):
    v = dict(r[k]); v.pop("roofline", None); print(k, json.dumps(v))

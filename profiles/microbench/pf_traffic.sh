#!/bin/bash
# HBM traffic per launch of the filter kernels ("gjx_gen_pf's counter traffic is 1.19x the algorithmic bytes, the
# hand-written instance reads 0.89x: nobody has said where the extra 1.1 GB per run comes from").  FETCH_SIZE and WRITE_SIZE in separate
# passes (MI355X_MICROARCH.md: KiB units, FETCH_SIZE x 2 on gfx950), generic filter (gjx_gen_pf) and hand-written filter
# (k_pf_persistent) at config 3's size.   usage: bash profiles/microbench/pf_traffic.sh <tag>
TAG=${1:-r06}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
cat > /tmp/pf_hand_run.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch
from genjax_amd import core, workloads
from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
s = workloads.ssm_problem()
bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), 1 << 18)
ys = torch.as_tensor(s["y"]).cuda()
for i in range(6):
    bf.run(core.key(1 + i), ys)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_pf_gen_$c -o pmc -- python $R/profiles/microbench/scan_filter_run.py > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_pf_hand_$c -o pmc -- python /tmp/pf_hand_run.py > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for who in ("gen", "hand"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$OUT/pmc_pf_%s_%s/**/*counter_collection.csv" % (who, c), recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    acc[r["Kernel_Name"].split("(")[0][:48]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                if "pf" in k or "gjx_gen" in k:
                    out.setdefault(who + ":" + k, {})[c] = dict(mean_KiB=sum(v) / len(v), launches=len(v))
algo = (8 * 8 + 24) * (1 << 18) * 255
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rd, wr = 2.0 * v["FETCH_SIZE"]["mean_KiB"] * 1024, v["WRITE_SIZE"]["mean_KiB"] * 1024
        v["read_bytes"], v["write_bytes"], v["total_over_algorithmic_255_steps"] = rd, wr, (rd + wr) / algo
        v["read_B_per_particle_step"], v["write_B_per_particle_step"] = rd / ((1 << 18) * 255), wr / ((1 << 18) * 255)
json.dump(out, open("$OUT/${TAG}_pf_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/pmc_pf_*

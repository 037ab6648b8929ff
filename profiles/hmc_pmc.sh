#!/bin/bash
# MFMA / VALU pipe utilisation of the config-5 HMC kernel (counters only), and the MFMA / vector-issue microbenchmark
# usage: bash profiles/hmc_pmc.sh <tag>     output: gpurun_out/<tag>/<tag>_hmc_pmc.txt, <tag>_mfma_valu_overlap_microbench.txt
TAG=${1:-r03}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG/hmcpmc; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|SQ_BUSY_CU|SQ_ACTIVE_INST|SQ_INST_CYCLES|GRBM_GUI" | head -60 > $OUT/counters.txt
CMD="python $R/bench.py --workload hmc --no-cpu-baseline --steps 2 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/p1 -o pmc -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o pmc -- $CMD > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
for sub in ("p1","p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        g = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "hmc" in r["Kernel_Name"]:
                g[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open("$OUT/summary.txt","a") as o:
            for k, cs in g.items():
                for c, v in cs.items():
                    print(k, c, len(v), sum(v)/len(v), file=o)
PY
tail -3 $OUT/p1.log $OUT/p2.log >> $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2
grep "k_hmc" $OUT/summary.txt > $R/gpurun_out/$TAG/${TAG}_hmc_pmc.txt
hipcc -O3 -std=c++17 --offload-arch=gfx950 -o $OUT/ub_mfma $R/profiles/microbench/ub_mfma.hip 2>/dev/null && $OUT/ub_mfma > $R/gpurun_out/$TAG/${TAG}_mfma_valu_overlap_microbench.txt
rm -rf $OUT

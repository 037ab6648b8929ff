#!/bin/bash
# Profile collection on the GPU box (run through gpurun): bench JSON lines, rocprofv3 kernel stats, PMC passes.
# usage: bash profiles/collect.sh r03 [gmm|ssm|hmc ...]      output: gpurun_out/<tag>/, copy what is to be judged into profiles/
TAG=${1:-r06}; shift; WL=${@:-gmm ssm hmc}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for w in $WL; do
  # the compact line the driver parses -> <tag>_bench_<w>.json; the full record (incl. --extra's measurements) -> <tag>_bench_<w>_full.json
  xa=; [ $w = gmm ] && xa=--extra
  python bench.py --workload $w $xa --extra-file $OUT/${TAG}_bench_${w}_full.json 2>/dev/null | tail -1 > $OUT/${TAG}_bench_$w.json
done
cd /tmp; export TMPDIR=/tmp
for w in $WL; do
  # gmm: the headline command WITHOUT the extra workloads (they launch the same kernels at other sizes, which would mix into
  # the per-kernel averages); the extras get their own trace below
  st=100; xf=--no-extra; [ $w != gmm ] && st=3 && xf=
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python $R/bench.py --workload $w --no-cpu-baseline $xf --steps $st --warmup 1 > $OUT/prof_$w.log 2>&1
  cp $(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${w}_kernel_stats.csv
done
if echo $WL | grep -q gmm; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_extra -o extra -- python $R/bench.py --no-cpu-baseline --extra --extra-file /tmp/bench_extra_prof.json --steps 20 --warmup 1 > $OUT/prof_extra.log 2>&1
  cp $(find $OUT/prof_extra -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_gmm_with_extras_kernel_stats.csv
fi
# PMC passes for the default bench (counters only: no tracing domains alongside --pmc); FETCH_SIZE and WRITE_SIZE in separate passes.
# Every pass under its own timeout: in the round-6 re-collection one counter pass died inside the tool at its first dispatch (SIGSEGV in
# rocprofv3's dispatch interception) and the next one hung for 58 minutes with "1838 incomplete dispatches" until the call's limit.
if echo $WL | grep -q gmm; then
  CMD="python $R/bench.py --no-cpu-baseline --extra --extra-file /tmp/bench_extra_pmc.json --steps 20 --warmup 3 --event-samples 2"
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
  timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
  timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/pmc_sq2.log 2>&1
  python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for sub in ("fetch", "write", "sq", "sq2"):
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % sub, recursive=True):
        # the extra workloads launch some kernels at several sizes: a kernel's launches are grouped by grid size and the
        # most frequent grid is the one reported (the headline kernel: 1024 blocks = K 2^20)
        g = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
        for r in csv.DictReader(open(f)):
            g[r["Kernel_Name"].split("(")[0][:60]][r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d = {}
        for k, grids in g.items():
            best = max(grids, key=lambda gs: max(len(v) for v in grids[gs].values()))
            d[k] = grids[best]
            if len(grids) > 1:
                res[k]["grid_size_reported"] = best
        for k, cs in d.items():
            for c, v in cs.items():
                res[k][c] = sum(v) / len(v)
                res[k]["launches_" + sub] = len(v)
res = {k: v for k, v in res.items() if "gjx" in k}
json.dump(res, open("$OUT/${TAG}_pmc_summary.json", "w"), indent=1)
# HBM bytes per launch as MI355X_MICROARCH.md (HBM) prescribes: counters are KiB, FETCH_SIZE x 2 on gfx950, WRITE_SIZE as is
traffic = {k.replace("void ", "").replace(" ", ""): (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
           for k, v in res.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
json.dump(traffic, open("$OUT/${TAG}_pmc_traffic.json", "w"), indent=1)
# VALU issue utilisation = cycles the VALU pipes were executing / cycles the shader engines were busy.
# SQ_ACTIVE_INST_VALU is summed over the 1024 SIMDs in units of 4 cycles (MI355X_MICROARCH.md, PMC units); SQ_BUSY_CYCLES is
# summed over the 32 shader engines in cycles (cross-check: SQ_BUSY_CYCLES / 32 / kernel duration = the shader clock).
util = {}
for k, v in res.items():
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_BUSY_CYCLES" in v and v["SQ_BUSY_CYCLES"] > 0:
        valu_cyc = 4.0 * v["SQ_ACTIVE_INST_VALU"] / 1024.0
        busy_cyc = v["SQ_BUSY_CYCLES"] / 32.0
        util[k.replace("void ", "").replace(" ", "")] = dict(
            valu_busy_cycles_per_simd=valu_cyc, se_busy_cycles=busy_cyc, valu_issue_utilisation=valu_cyc / busy_cyc,
            valu_instr_per_wave=v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_WAVES", 1), 1),
            cycles_per_valu_instr=4.0 * v["SQ_ACTIVE_INST_VALU"] / max(v.get("SQ_INSTS_VALU", 1), 1))
json.dump(util, open("$OUT/${TAG}_valu_utilisation.json", "w"), indent=1)
PY
fi
if echo $WL | grep -q hmc; then bash $R/profiles/hmc_pmc.sh $TAG > /dev/null 2>&1; fi
# the generic Scan filter alone (config 3's model as @gen + .scan, K = 2^18, T = 256): its step IS one launch of gjx_gen
python $R/profiles/microbench/scan_filter_run.py 2>/dev/null | grep "us per step" > $OUT/${TAG}_scan_filter_kernel_stats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_scanf -o scanf -- python $R/profiles/microbench/scan_filter_run.py > /dev/null 2>&1
grep -E "Name|gjx" $(find $OUT/prof_scanf -name "*kernel_stats.csv" | head -1) >> $OUT/${TAG}_scan_filter_kernel_stats.txt
GJX_SSM_PERSISTENT=0 python $R/profiles/microbench/ssm_timeline.py 2>/dev/null | grep " us" > $OUT/${TAG}_ssm_step_timeline.txt
python $R/profiles/microbench/ssm_persistent_timeline.py 2>/dev/null | grep -E " us|blocks" > $OUT/${TAG}_ssm_persistent_timeline.txt
SSM_WEIGHTS=tile_scaled python $R/profiles/microbench/ssm_persistent_timeline.py 2>/dev/null | grep -E " us|blocks" >> $OUT/${TAG}_ssm_persistent_timeline.txt
python $R/profiles/microbench/gather_timeline.py 2>/dev/null | grep " us" > $OUT/${TAG}_resample_gather_timeline.txt
python $R/profiles/microbench/gather_tiled_timeline.py 2>/dev/null | grep -E " us" > $OUT/${TAG}_resample_gather_tiled_timeline.txt
GJX_SCAN_FILTER_WIDE=0 python $R/profiles/microbench/scan_steps_timeline.py 2>/dev/null | grep -E " us|blocks" > $OUT/${TAG}_scan_steps_timeline.txt
# the filter kernel generated for the step program on the shared skeleton (gjx_gen_pf): phases of one step at 2^18 and 2^20 particles
python $R/profiles/microbench/pf_gen_timeline.py 2>/dev/null | grep -E " us|blocks" > $OUT/${TAG}_pf_gen_timeline.txt
KK=1048576 python $R/profiles/microbench/pf_gen_timeline.py 2>/dev/null | grep -E " us|blocks" >> $OUT/${TAG}_pf_gen_timeline.txt
python $R/profiles/microbench/pf_hand_timeline.py 2>/dev/null | grep -E " us|blocks" > $OUT/${TAG}_pf_hand_timeline.txt
# one launch of the generated mixture kernel: prologue / sites / end per block
python $R/profiles/microbench/gen_kernel_timeline.py 2>/dev/null | grep -E " us|blocks" > $OUT/${TAG}_gen_kernel_timeline.txt
bash $R/profiles/microbench/pf_traffic.sh $TAG > /dev/null 2>&1     # filter kernels' HBM traffic, generated and hand-written, one size per pass
python $R/profiles/microbench/sv_bias.py 256 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_sv_bias.txt
bash $R/profiles/gputests.sh $TAG > /dev/null 2>&1; cp $OUT/gputest_summary.txt $OUT/${TAG}_gputest_summary.txt; rm -f $OUT/gputest_test_*.txt $OUT/gputest_summary.txt
rm -rf $OUT/prof_*/ $OUT/pmc_*/
ls $OUT

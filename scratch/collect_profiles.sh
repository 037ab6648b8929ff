#!/bin/bash
# Round profile collection on the GPU box: bench JSON lines, rocprofv3 kernel stats, PMC passes.  Output: gpurun_out/r01/
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r01; mkdir -p $OUT
cd $R
python bench.py 2>/dev/null | tail -1 > $OUT/r01_bench_gmm.json
python bench.py --workload ssm 2>/dev/null | tail -1 > $OUT/r01_bench_ssm.json
python bench.py --workload hmc 2>/dev/null | tail -1 > $OUT/r01_bench_hmc.json
GJX_FORCE_DIST=1 python bench.py --no-cpu-baseline 2>/dev/null | grep metric > $OUT/r01_bench_gmm_sharded_path_1rank.json
cd /tmp; export TMPDIR=/tmp
for w in gmm ssm hmc; do
  st=100; [ $w != gmm ] && st=3
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python $R/bench.py --workload $w --no-cpu-baseline --steps $st --warmup 1 > $OUT/prof_$w.log 2>&1
  cp $(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/r01_${w}_kernel_stats.csv
done
GJX_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_gmm_sharded -o gmm -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 1 > $OUT/prof_gmm_sharded.log 2>&1
cp $(find $OUT/prof_gmm_sharded -name "*kernel_stats.csv" | head -1) $OUT/r01_gmm_sharded_path_1rank_kernel_stats.csv
bash $R/scratch/pmc_bench.sh > $OUT/pmc.log 2>&1
cp $R/gpurun_out/pmc_bench/summary.json $OUT/r01_pmc_summary.json
rm -rf $OUT/prof_*/ 
ls -la $OUT; head -c 600 $OUT/r01_bench_gmm.json; echo; head -12 $OUT/r01_gmm_kernel_stats.csv; tail -5 $OUT/pmc.log | cut -c1-400

#!/bin/bash
# Wider differential campaigns than the default suite runs: random programs (propagate engines, gradients, HMC emitter) and random
# resampling shapes against the oracle, several seeds.  usage: bash profiles/fuzz.sh <trials> <seed> ...   output: gpurun_out/fuzz/
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/fuzz; mkdir -p $OUT; cd $R
TR=${1:-150}; shift; SEEDS=${@:-1001 1002}
for sd in $SEEDS; do
  GJX_FUZZ_TRIALS=$TR GJX_FUZZ_SEED=$sd timeout 1500 python -m pytest -m gpu -q \
    "tests/test_gpu_parity.py::test_random_programs_against_oracle" "tests/test_gpu_parity.py::test_resampling_fuzz_against_oracle" \
    "tests/test_gpu_hmcgen.py::test_random_programs_generated_hmc_vs_interpreter_and_oracle" \
    "tests/test_gpu_expr.py::test_random_expression_programs_against_oracle" > $OUT/seed_$sd.txt 2>&1
  echo "seed $sd trials $TR: $(grep -E "passed|failed" $OUT/seed_$sd.txt | tail -1)" >> $OUT/summary.txt
  grep -E "^E  " $OUT/seed_$sd.txt | head -12 >> $OUT/summary.txt
done
cat $OUT/summary.txt

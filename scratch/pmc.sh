#!/bin/bash
# usage: pmc.sh <outdir> <script> ; collects PMC passes separately (no tracing domains)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; SCRIPT=$R/scratch/$2
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_WR" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- python $SCRIPT > $OUT.p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in d.items():
        print(k, {c: sum(v) / len(v) for c, v in cs.items()}, "n", len(next(iter(cs.values()))))
PY

#!/bin/bash
# PMC passes for bench.py (counters only: no tracing domains alongside --pmc).  FETCH_SIZE and WRITE_SIZE need
# separate passes (TCC slots).  Output: gpurun_out/pmc_bench/{fetch,write,sq}/ + summary JSON.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_bench
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 --event-samples 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- $CMD > $OUT.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- $CMD > $OUT.write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -o pmc -- $CMD > $OUT.sq.log 2>&1
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for sub in ("fetch", "write", "sq"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in d.items():
            for c, v in cs.items():
                res[k][c] = sum(v) / len(v)
                res[k]["launches_" + sub] = len(v)
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, v in res.items():
    if "gjx" in k: print(k, v)
PY

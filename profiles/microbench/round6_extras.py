"""python profiles/microbench/round6_extras.py [substring]: the round-6 extras of bench.py on their own (bench.run_round6), printed one per line"""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch      # noqa: E402
import bench      # noqa: E402

r = bench.run_round6(torch.device("cuda:0"))
for k, v in r.items():
    if len(sys.argv) < 2 or sys.argv[1] in k:
        v = dict(v) if isinstance(v, dict) else v
        if isinstance(v, dict):
            v.pop("roofline", None)
        print(k, json.dumps(v)[:900])

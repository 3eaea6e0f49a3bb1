"""Sums the rocprofv3 --pmc counter CSVs of scripts/gpu_pmc.sh per kernel (one launch per pass) and attaches the
bench line of each pass:  python tools/pmc_summary.py gpurun_out/pmc_r01d > profiles/r01/pmc_<tag>.json"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for pas in ("sq", "sq2", "fetch", "write"):
    files = glob.glob(os.path.join(root, pas, "**", "*counter_collection.csv"), recursive=True)
    agg, ids = {}, {}
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "extend" not in name and "seed_p16" not in name:
                continue
            short = name.split("(")[0].replace("void ", "").replace("gact::", "")
            agg.setdefault(short, {}).setdefault(r["Counter_Name"], 0.0)
            agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
            ids.setdefault(short, set()).add(r.get("Dispatch_Id"))
    bench = {}
    try:
        for line in open(os.path.join(root, pas + ".json")):
            if line.startswith("{"):
                d = json.loads(line)
                rf = d["roofline"]
                bench = {"value": d["value"], "kernel": rf["kernel"], "kernel_ms": rf["kernel_ms"],
                         "seed_kernel": rf.get("seed_kernel"), "seed_kernel_ms": rf["seed_kernel_ms"],
                         "kernel_cells": rf["kernel_cells"], "seed_kernel_cells": rf["seed_kernel_cells"]}
    except OSError:
        pass
    # counters are AVERAGES PER LAUNCH: a pass runs bench.py's timed step and the steps of its roofline leg, every one of them
    # one launch of each kernel in the plain sequence (GACT_HIP_NO_OVERLAP=1)
    for short, c in agg.items():
        n = max(len(ids.get(short, ())), 1)
        for k in list(c):
            c[k] = c[k] / n
        c["_launches_averaged"] = n
    agg["_bench"] = bench
    out[pas] = agg
print(json.dumps(out, indent=1))

"""Summary of a rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE pass of the DEFAULT bench command (steps in flight):
counters summed over every kernel dispatch of the process, divided by the steps the command ran (warm-up + timed +
the one-at-a-time leg; every step is the same work).  python tools/pmc_default_summary.py <dir with the csv> <bench line json>
 > profiles/rNN/pmc_default_<workload>.json"""
import csv
import glob
import json
import os
import sys

root, bench_path = sys.argv[1], sys.argv[2]
bench = None
for line in open(bench_path):
    if line.startswith("{"):
        bench = json.loads(line)
steps = bench["config"].get("passes_in_this_process")
if steps is None:                     # (a line of before the key existed: warm-up + timed + one-at-a-time leg + roofline leg of 1 + 4)
    steps = bench["steps"] + bench["warmup"] + (bench["single_slot"]["steps"] if bench["config"]["slots_in_flight"] > 1 else 0) + 1 + min(4, max(2, bench["steps"]))
kernels = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gact::", "")
        if name.startswith(("pack_kernel", "valu_probe", "revcomp", "poison")):
            continue                                  # set-up and the issue-rate probe: not part of a step
        k = kernels.setdefault(name, {"dispatches": 0, "SQ_INSTS_VALU": 0.0, "GRBM_GUI_ACTIVE": 0.0})
        if r["Counter_Name"] in k:
            k[r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            k["dispatches"] += 1
total = sum(k["SQ_INSTS_VALU"] for k in kernels.values())
print(json.dumps({"command": "bench.py --workload %s --no-others --no-cpu (default slots, steps, warm-up) under rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE"
                             % bench["config"]["workload"].replace("_self_overlap", ""),
                  "workload": bench["config"]["workload"], "cells_per_step": bench["config"]["cells_per_step"],
                  "steps_profiled": steps, "slots_in_flight": bench["config"]["slots_in_flight"],
                  "insts_valu_total": total, "insts_valu_per_step": total / steps,
                  "value_under_profiler": bench["value"], "ms_per_step_under_profiler": bench["ms_per_step"],
                  "kernels": kernels}, indent=1))

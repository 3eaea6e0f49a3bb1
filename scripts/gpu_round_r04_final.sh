# the build the round ends with, once more after the last changes: smoke, the whole -m gpu suite, the default bench line
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-r04_final}
mkdir -p $OUT
timeout -k 10 300 python __graft_entry__.py --smoke 2>&1 | tail -2
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=12 > $OUT/pytest_gpu_final.log 2>&1 || { tail -30 $OUT/pytest_gpu_final.log; exit 1; }
tail -16 $OUT/pytest_gpu_final.log
timeout -k 10 600 python bench.py > $OUT/bench_final_ecoli10x_n1.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python - "$OUT/bench_final_ecoli10x_n1.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "| single", d["single_slot"], "| feeders", d.get("feeder_threads", {}).get("value"), d.get("feeder_threads", {}).get("callers_merged_in_last_launch"))
print("roofline kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "| pipelined", d["roofline"]["pipelined"]["frac"], d["roofline"]["pipelined"]["valu_issue_utilisation"], d["roofline"]["pipelined"]["source"][:50])
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("parity"))
for o in d.get("other_configs", []):
    print(o["workload"], o["value"], o["single_slot"], o["kernel_layout"])
for v in d.get("variants", []):
    print(v["variant"][:50], v["value"], v["single_slot"]["value"], v["kernel_layout"])
PY

# round 4, round-end evidence part A: the whole -m gpu suite (slowest tests listed), the default bench line, the reference's
# caller at speed.  Results under gpurun_out/$TAG; what is to be judged is copied into profiles/r04/ afterwards.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-r04_end_a}
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=25 > $OUT/pytest_gpu_round_end.log 2>&1 || { tail -30 $OUT/pytest_gpu_round_end.log; exit 1; }
tail -32 $OUT/pytest_gpu_round_end.log
timeout -k 10 600 python bench.py > $OUT/bench_round_end_ecoli10x_n1.json 2> $OUT/bench_ecoli10x.err || { tail -5 $OUT/bench_ecoli10x.err; exit 1; }
python - "$OUT/bench_round_end_ecoli10x_n1.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "| single", d["single_slot"], "| feeders", d.get("feeder_threads", {}).get("value"), d.get("feeder_threads", {}).get("callers_merged_in_last_launch"))
print("roofline kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "| pipelined", d["roofline"]["pipelined"]["frac"], d["roofline"]["pipelined"]["valu_issue_utilisation"])
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("parity"))
for o in d.get("other_configs", []):
    print(o["workload"], o["value"], o["single_slot"], o["kernel_layout"])
for v in d.get("variants", []):
    print(v["variant"][:50], v["value"], v["single_slot"]["value"], v["kernel_layout"])
PY
for i in 1 2; do
timeout -k 10 600 python tools/darwin_on_hip_timing.py ecoli10x 8 > $OUT/darwin_on_hip_8_threads_round_end_$i.json 2> $OUT/darwin_on_hip.err || { tail -5 $OUT/darwin_on_hip.err; exit 1; }
python - "$OUT/darwin_on_hip_8_threads_round_end_$i.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for r in d["runs"]:
    cs = r["shim_split_per_call_us"]
    print(r["mode"][:12], r["gact_calling_ms_max_over_threads"], "ms; launches ms:", sorted(set(c["launch_ms"] for c in cs)), "merged:", sorted(set(c["merged"] for c in cs)), "GCUPS of the stage", r["gcups_of_the_gact_stage"])
PY
done

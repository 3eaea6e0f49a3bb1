# round 4, evidence part A: team-walker A/B, the default bench line, the reference's caller at speed
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-r04_round_a}
mkdir -p $OUT
L=darwin-gpu_amd/libgact_hip.so
CASES="team|$L|;noteam|$L|GACT_HIP_NO_TEAM_WHEN_SHARED=1" WORKLOADS="ecoli10x" REPS=3 TAG=${TAG:-r04_round_a}/ab_team bash scripts/gpu_env_ab.sh | tee $OUT/ab_team_walker.txt
timeout -k 10 600 python bench.py > $OUT/bench_ecoli10x_n1.json 2> $OUT/bench_ecoli10x.err || tail -5 $OUT/bench_ecoli10x.err
python - "$OUT/bench_ecoli10x_n1.json" <<'PY'
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
timeout -k 10 900 python tools/darwin_on_hip_timing.py ecoli10x 8 > $OUT/darwin_on_hip_8_threads.json 2> $OUT/darwin_on_hip.err || tail -5 $OUT/darwin_on_hip.err
cat $OUT/darwin_on_hip_8_threads.json

# C++ RCCL gather tests + dist tests; then the default bench with and without SDMA copies (HSA_ENABLE_SDMA), interleaved
set -e
OUT=gpurun_out/${TAG:-r04h}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_shim.py tests/test_gpu_dist.py tests/test_gpu_slots.py -x -q -m gpu 2>&1 | tail -15
for rep in 1 2; do
for v in default:X=1 nosdma:HSA_ENABLE_SDMA=0; do
  name=${v%%:*}; e=${v#*:}
  env $e timeout -k 10 400 python bench.py --no-cpu > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err || { tail -5 $OUT/bench_${name}_$rep.err; exit 1; }
  python - "$OUT/bench_${name}_$rep.json" $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "value", d["value"], "| single", d["single_slot"]["value"], d["single_slot"]["ms_per_step"], "| feeders", d.get("feeder_threads", {}).get("value"),
      "| others", [(o["workload"][:4], o["value"], o["single_slot"]["value"]) for o in d.get("other_configs", [])],
      "| variants", [(v["value"], v["single_slot"]["value"]) for v in d.get("variants", [])])
PY
done
done

# the critical lane: parity first (small tests, then every candidate of ecoli10x), then A/B against GACT_HIP_NO_CRIT_LANE=1
set -e
OUT=gpurun_out/${TAG:-r04p}
mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_scheduling.py tests/test_gpu_routing.py tests/test_gpu_slots.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 400 python -m pytest tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_poison.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 400 python -m pytest "tests/test_gpu_configs.py::test_config2_ecoli10x_every_candidate" "tests/test_gpu_configs.py::test_config2_eight_feeder_slots" -x -q -m gpu -s 2>&1 | tail -6
for rep in 1 2 3; do
for v in lane:X=1 nolane:GACT_HIP_NO_CRIT_LANE=1; do
  name=${v%%:*}; e=${v#*:}
  env $e timeout -k 10 300 python bench.py --no-cpu --no-others --steps 10 > $OUT/b_${name}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python - "$OUT/b_${name}_$rep.json" $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "value", d["value"], d["ms_per_step"], "| single", d["single_slot"]["value"], d["single_slot"]["ms_per_step"], "| plain sequence main", d["roofline"]["kernel_ms"], "seed", d["roofline"]["seed_kernel_ms"])
PY
done
done
for v in "X=1" "GACT_HIP_NO_CRIT_LANE=1"; do
  echo "== reference caller, $v"
  env $v timeout -k 10 300 python tools/darwin_on_hip_timing.py ecoli10x 8 > $OUT/d.json 2> $OUT/d.err || { tail -5 $OUT/d.err; exit 1; }
  python - <<'PY'
import json, os
d = json.load(open("gpurun_out/%s/d.json" % os.environ.get("TAG", "r04p")))
for r in d["runs"]:
    cs = r["shim_split_per_call_us"]
    print(r["mode"][:12], r["gact_calling_ms_max_over_threads"], "ms; launches ms:", sorted(set(c["launch_ms"] for c in cs)), "merged:", sorted(set(c["merged"] for c in cs)))
PY
  cp $OUT/d.json $OUT/darwin_on_hip_$(echo $v | tr '=' '_').json
done

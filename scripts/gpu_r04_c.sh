set -e
OUT=gpurun_out/${TAG:-r04c}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_scheduling.py tests/test_gpu_shim.py tests/test_reference_caller.py tests/test_gpu_slots.py -x -q -m gpu --durations=8 2>&1 | tail -16 | tee $OUT/pytest.log
timeout -k 10 300 python -m pytest "tests/test_gpu_configs.py::test_config2_eight_feeder_slots" -x -q -m gpu -s 2>&1 | tail -5 | tee $OUT/feeders.log
timeout -k 10 300 python bench.py --no-cpu --steps 10 > $OUT/bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/%s/bench.json" % __import__("os").environ.get("TAG", "r04c")))
print("value", d["value"], "ms/step", d["ms_per_step"], "single", d["single_slot"], "feeders", d.get("feeder_threads"))
print("roofline mode:", d["roofline"]["mode"], "kernel_ms", d["roofline"]["kernel_ms"], "pipelined:", d["roofline"]["pipelined"])
for o in d.get("other_configs", []):
    print(o["workload"], o["value"], o["single_slot"], o["kernel_layout"])
for v in d.get("variants", []):
    print(v["variant"][:40], v["value"], v["single_slot"]["value"])
PY

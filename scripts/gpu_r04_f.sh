# the reference's caller after the pinned upload staging; slot / scheduling / shim tests; ONT shape in flight with the caller's hint
set -e
OUT=gpurun_out/${TAG:-r04f}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_slots.py tests/test_gpu_scheduling.py tests/test_gpu_shim.py tests/test_reference_caller.py -x -q -m gpu 2>&1 | tail -5
GACT_HIP_TRACE_UPLOAD=1 timeout -k 10 600 python tools/darwin_on_hip_timing.py ecoli10x 8 > $OUT/darwin_on_hip_8_threads.json 2> $OUT/darwin_on_hip.err || { tail -5 $OUT/darwin_on_hip.err; exit 1; }
python - <<'PY'
import json, os
d = json.load(open("gpurun_out/%s/darwin_on_hip_8_threads.json" % os.environ.get("TAG", "r04f")))
for r in d["runs"]:
    print(r["mode"], r["gact_calling_ms_max_over_threads"], "ms")
    for c in r["shim_split_per_call_us"]: print("  ", c)
    for t in r.get("engine_trace", []): print("  ", t)
PY
for i in 1 2; do
timeout -k 10 300 python bench.py --workload ont --no-others --no-cpu --steps 8 --warmup 4 > $OUT/bench_ont_$i.json 2> $OUT/bench_ont.err || { tail -5 $OUT/bench_ont.err; exit 1; }
python -c "
import json; d=json.load(open('$OUT/bench_ont_$i.json')); print('ont', d['value'], d['ms_per_step'], d['single_slot'])"
done

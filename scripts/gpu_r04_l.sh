set -e
OUT=gpurun_out/${TAG:-r04l}
mkdir -p $OUT
echo "== stream pool: pacbio50mb, ont, ont in one process (default runtime settings)"
timeout -k 10 500 python tools/side_probe.py pacbio50mb ont ont | cut -c1-210
timeout -k 10 400 python -m pytest tests/test_gpu_slots.py tests/test_gpu_scheduling.py tests/test_gpu_shim.py tests/test_gpu_poison.py -x -q -m gpu 2>&1 | tail -3
GACT_HIP_TRACE_UPLOAD=1 timeout -k 10 300 python tools/darwin_on_hip_timing.py ecoli10x 8 > $OUT/d.json 2> $OUT/d.err || { tail -5 $OUT/d.err; exit 1; }
python - <<'PY'
import json, os
d = json.load(open("gpurun_out/%s/d.json" % os.environ.get("TAG", "r04l")))
for r in d["runs"]:
    cs = r["shim_split_per_call_us"]
    print(r["mode"][:12], r["gact_calling_ms_max_over_threads"], "ms; merged:", [c["merged"] for c in cs], "wait-launch ms:", [round(c["wait_fetch"] / 1e3 - c["launch_ms"], 1) for c in cs])
    for t in r.get("engine_trace", []):
        if "fetch" in t: print("   ", t)
PY

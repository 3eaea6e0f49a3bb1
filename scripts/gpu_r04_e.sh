# where the first call of a feeder thread spends its time (GACT_HIP_TRACE_UPLOAD) in the reference's caller
set -e
OUT=gpurun_out/${TAG:-r04e}
mkdir -p $OUT
GACT_HIP_TRACE_UPLOAD=1 timeout -k 10 600 python tools/darwin_on_hip_timing.py ecoli10x 8 > $OUT/darwin_on_hip_8_threads.json 2> $OUT/darwin_on_hip.err || { tail -5 $OUT/darwin_on_hip.err; exit 1; }
python - <<'PY'
import json, os
d = json.load(open("gpurun_out/%s/darwin_on_hip_8_threads.json" % os.environ.get("TAG", "r04e")))
for r in d["runs"]:
    print(r["mode"], r["gact_calling_ms_max_over_threads"], "ms")
    for c in r["shim_split_per_call_us"]: print("  ", c)
    for t in r.get("engine_trace", []): print("  ", t)
PY

# which runtime setting makes the stalled hipMemcpyAsync of a feeder thread's first call go away?
OUT=gpurun_out/${TAG:-r04g}
mkdir -p $OUT
for v in "X=1" "GPU_MAX_HW_QUEUES=16" "AMD_DIRECT_DISPATCH=0" "HSA_ENABLE_SDMA=0"; do
  echo "== $v"
  env $v GACT_HIP_TRACE_UPLOAD=1 timeout -k 10 300 python tools/darwin_on_hip_timing.py ecoli10x 8 > $OUT/d.json 2> $OUT/d.err || { tail -5 $OUT/d.err; continue; }
  python - <<'PY'
import json, os
d = json.load(open("gpurun_out/%s/d.json" % os.environ.get("TAG", "r04g")))
for r in d["runs"]:
    ups = [c["upload"] for c in r["shim_split_per_call_us"]]
    print(r["mode"][:12], r["gact_calling_ms_max_over_threads"], "ms; uploads > 1 ms:", [u for u in ups if u > 1000], "merged:", sorted(set(c["merged"] for c in r["shim_split_per_call_us"])))
PY
done

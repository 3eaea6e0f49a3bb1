# how often do eight feeder threads end up in two groups?  the FEEDERS test five times, the reference's caller twice, bench's feeder leg
set -e
OUT=gpurun_out/${TAG:-r04z}
mkdir -p $OUT
for i in 1 2 3 4 5; do
timeout -k 10 300 python -m pytest "tests/test_gpu_configs.py::test_config2_eight_feeder_slots" -x -q -m gpu -s 2>&1 | grep -E "FEEDERS|passed|failed" | cut -c1-420
done
timeout -k 10 300 python -m pytest tests/test_gpu_scheduling.py tests/test_gpu_shim.py tests/test_reference_caller.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 600 python tools/darwin_on_hip_timing.py ecoli10x 8 > $OUT/d_$i.json 2> $OUT/d.err || { tail -5 $OUT/d.err; exit 1; }
python - "$OUT/d_$i.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for r in d["runs"]:
    cs = r["shim_split_per_call_us"]
    print(r["mode"][:12], r["gact_calling_ms_max_over_threads"], "ms; launches ms:", sorted(set(c["launch_ms"] for c in cs)), "merged:", sorted(set(c["merged"] for c in cs)))
PY
done

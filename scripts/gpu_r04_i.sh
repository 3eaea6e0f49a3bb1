# (1) the reference's caller with the engine's own bus copies (no SDMA on the small transfers)
# (2) ONT shape: bench.py's side configuration in a fresh process vs the main path
set -e
OUT=gpurun_out/${TAG:-r04i}
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_shim.py tests/test_gpu_slots.py tests/test_gpu_scheduling.py -x -q -m gpu 2>&1 | tail -3
for v in "X=1" "GACT_HIP_SDMA_COPIES=1"; do
  echo "== $v"
  env $v GACT_HIP_TRACE_UPLOAD=1 timeout -k 10 300 python tools/darwin_on_hip_timing.py ecoli10x 8 > $OUT/d.json 2> $OUT/d.err || { tail -5 $OUT/d.err; exit 1; }
  python - <<'PY'
import json, os
d = json.load(open("gpurun_out/%s/d.json" % os.environ.get("TAG", "r04i")))
for r in d["runs"]:
    cs = r["shim_split_per_call_us"]
    print(r["mode"][:12], r["gact_calling_ms_max_over_threads"], "ms; uploads > 1 ms:", [c["upload"] for c in cs if c["upload"] > 1000], "merged:", [c["merged"] for c in cs],
          "wait-launch ms:", [round(c["wait_fetch"] / 1e3 - c["launch_ms"], 1) for c in cs])
PY
  cp $OUT/d.json $OUT/darwin_on_hip_$(echo $v | tr '=' '_').json
done
echo "== ont, side configuration in a fresh process"
timeout -k 10 300 python tools/side_probe.py ont
echo "== ont, main path"
timeout -k 10 300 python bench.py --workload ont --no-others --no-cpu --steps 8 --warmup 4 > $OUT/bench_ont.json 2> $OUT/bench_ont.err
python -c "
import json; d=json.load(open('$OUT/bench_ont.json')); print('ont main path', d['value'], d['ms_per_step'], d['single_slot'])"

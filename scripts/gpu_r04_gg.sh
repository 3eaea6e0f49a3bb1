# size of the critical lane for the merged half-size calls of the reference's caller
set -e
OUT=gpurun_out/${TAG:-r04gg}
mkdir -p $OUT
for rep in 1 2; do
for b in 256 128 384 192; do
  GACT_HIP_LANE_BLOCKS=$b timeout -k 10 300 python tools/darwin_on_hip_timing.py ecoli10x 8 > $OUT/d_${b}_$rep.json 2> $OUT/d.err || { tail -5 $OUT/d.err; exit 1; }
  python - "$OUT/d_${b}_$rep.json" $b <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["runs"][0]
cs = r["shim_split_per_call_us"]
print("lane blocks", sys.argv[2], r["mode"][:12], r["gact_calling_ms_max_over_threads"], "ms; launches ms:", sorted(set(c["launch_ms"] for c in cs)), "merged:", sorted(set(c["merged"] for c in cs)))
PY
done
done

# width of the stored pointer band once more on the round-end build: 32, 40, 48 (default), 56 columns; interleaved
set -e
OUT=gpurun_out/${TAG:-r04bb}
mkdir -p $OUT
for rep in 1 2; do
for b in 48 32 40 56; do
  GACT_HIP_BAND=$b timeout -k 10 300 python bench.py --no-cpu --no-others --steps 10 > $OUT/b_${b}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python - "$OUT/b_${b}_$rep.json" $b <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("band", sys.argv[2], "value", d["value"], d["ms_per_step"], "| single", d["single_slot"]["value"], d["single_slot"]["ms_per_step"], "| plain sequence main", d["roofline"]["kernel_ms"])
PY
done
done

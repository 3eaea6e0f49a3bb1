# ONT shape alone: resident blocks per CU of the wide launch (2 is the default), interleaved
set -e
OUT=gpurun_out/${TAG:-r04cc}
mkdir -p $OUT
for rep in 1 2; do
for b in 2 3 1; do
  GACT_HIP_WIDE_BLOCKS_PER_CU=$b timeout -k 10 300 python bench.py --workload ont --no-cpu --no-others --slots 1 --steps 4 --warmup 1 > $OUT/b_${b}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python - "$OUT/b_${b}_$rep.json" $b <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("wide blocks per CU", sys.argv[2], "one at a time", d["value"], d["ms_per_step"], "| main", d["roofline"]["kernel_ms"], "seed", d["roofline"]["seed_kernel_ms"])
PY
done
done

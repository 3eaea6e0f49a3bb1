# ONT shape alone: the critical lane (one wide block per CU for the longest chains) + the split launch for the rest, against the
# all-wide launch at one and two blocks per CU; bench.py --workload ont --slots 1 (with its parity gate), interleaved
set -e
OUT=gpurun_out/${TAG:-r04dd}
mkdir -p $OUT
for rep in 1 2; do
for v in "wide2:X=1" "wide1:GACT_HIP_WIDE_BLOCKS_PER_CU=1" "lane:GACT_HIP_LANE_SMALL=3" "lane128:GACT_HIP_LANE_SMALL=3 GACT_HIP_LANE_BLOCKS=128" "lane384:GACT_HIP_LANE_SMALL=3 GACT_HIP_LANE_BLOCKS=384"; do
  name=${v%%:*}; e=${v#*:}
  env $e timeout -k 10 300 python bench.py --workload ont --no-cpu --no-others --slots 1 --steps 4 --warmup 1 > $OUT/b_${name}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python - "$OUT/b_${name}_$rep.json" $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-8s" % sys.argv[2], "one at a time", d["value"], d["ms_per_step"], "| main", d["roofline"]["kernel_ms"], "seed", d["roofline"]["seed_kernel_ms"], "lane", d["single_slot"].get("critical_lane"))
PY
done
done

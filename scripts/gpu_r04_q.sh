# a SMALL critical lane beside the one main launch of a full-size run (plain sequence, no overlapped seeding): does taking only
# the very longest chains off the split launch shorten it?  one-at-a-time figures, interleaved
set -e
OUT=gpurun_out/${TAG:-r04q}
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_scheduling.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
for v in "plain:GACT_HIP_NO_OVERLAP=1 GACT_HIP_NO_CRIT_LANE=1" "lane16:GACT_HIP_NO_OVERLAP=1 GACT_HIP_CRIT_LANE_ALWAYS=1 GACT_HIP_LANE_BLOCKS=16" "lane48:GACT_HIP_NO_OVERLAP=1 GACT_HIP_CRIT_LANE_ALWAYS=1 GACT_HIP_LANE_BLOCKS=48" "lane128:GACT_HIP_NO_OVERLAP=1 GACT_HIP_CRIT_LANE_ALWAYS=1 GACT_HIP_LANE_BLOCKS=128" "overlapped:X=1"; do
  name=${v%%:*}; e=${v#*:}
  env $e timeout -k 10 300 python bench.py --no-cpu --no-others --slots 1 --steps 6 --warmup 2 > $OUT/b_${name}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python - "$OUT/b_${name}_$rep.json" $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-10s" % sys.argv[2], "one at a time", d["value"], d["ms_per_step"], "| main", d["roofline"]["kernel_ms"], "seed", d["roofline"]["seed_kernel_ms"], "lane", d["single_slot"].get("critical_lane"))
PY
done
done

# does a hardware queue of its own for the aux stream (overlapped seeding) change the one-at-a-time figure?  interleaved
set -e
OUT=gpurun_out/${TAG:-r04m}
mkdir -p $OUT
for rep in 1 2 3; do
for v in default:X=1 hwq8:GPU_MAX_HW_QUEUES=8; do
  name=${v%%:*}; e=${v#*:}
  env $e timeout -k 10 300 python bench.py --no-cpu --no-others --steps 10 > $OUT/b_${name}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python - "$OUT/b_${name}_$rep.json" $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "value", d["value"], d["ms_per_step"], "| single", d["single_slot"]["value"], d["single_slot"]["ms_per_step"], "| plain sequence main", d["roofline"]["kernel_ms"], "seed", d["roofline"]["seed_kernel_ms"])
PY
done
done

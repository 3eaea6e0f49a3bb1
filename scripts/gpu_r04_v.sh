# four blocks per CU for the linear-gap split launch (128 registers, 30 spilled outside the step loops; LDS 4 x 40,704 B fits):
# build/libgact_b4.so against the default library, bench.py (with its parity gate), interleaved
set -e
OUT=gpurun_out/${TAG:-r04v}
mkdir -p $OUT
for rep in 1 2 3; do
for v in b3:darwin-gpu_amd/libgact_hip.so b4:build/libgact_b4.so; do
  name=${v%%:*}; lib=${v#*:}
  GACT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python bench.py --no-cpu --no-others --steps 10 > $OUT/b_${name}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python - "$OUT/b_${name}_$rep.json" $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "value", d["value"], d["ms_per_step"], "| single", d["single_slot"]["value"], d["single_slot"]["ms_per_step"], "| plain sequence main", d["roofline"]["kernel_ms"], "seed", d["roofline"]["seed_kernel_ms"], "| ws GB", d["config"].get("workspace_gb"))
PY
done
done

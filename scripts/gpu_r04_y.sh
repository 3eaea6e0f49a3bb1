# the one-lane linear-gap walker with a refill per sixteen moves (default build) against one per eight (build/libgact_span8.so):
# parity first, then bench.py interleaved
set -e
OUT=gpurun_out/${TAG:-r04y}
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_scheduling.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 400 python -m pytest "tests/test_gpu_configs.py::test_config2_ecoli10x_every_candidate" "tests/test_gpu_configs.py::test_config3_pacbio50mb" -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
for v in span16:darwin-gpu_amd/libgact_hip.so span8:build/libgact_span8.so; do
  name=${v%%:*}; lib=${v#*:}
  GACT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python bench.py --no-cpu --no-others --steps 10 > $OUT/b_${name}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python - "$OUT/b_${name}_$rep.json" $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "value", d["value"], d["ms_per_step"], "| single", d["single_slot"]["value"], d["single_slot"]["ms_per_step"], "| plain sequence main", d["roofline"]["kernel_ms"], "seed", d["roofline"]["seed_kernel_ms"])
PY
done
done

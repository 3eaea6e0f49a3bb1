set -e
OUT=gpurun_out/${TAG:-r04aff}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py -x -q -m gpu -k "other_scoring or geometr or kernel_family" 2>&1 | tail -5 | tee $OUT/pytest.log
for lib in ${LIBS:-darwin-gpu_amd/libgact_hip.so}; do
GACT_HIP_LIB_PATH=$PWD/$lib timeout -k 10 400 python bench.py --only-variants --steps 4 --warmup 1 --no-cpu > $OUT/bench_variants.json 2> $OUT/bench_variants.err || tail -5 $OUT/bench_variants.err
python - <<PY
import json
d = json.load(open("$OUT/bench_variants.json"))
for v in d.get("variants", []):
    print("$lib", v["variant"][:50], "| in flight", v["value"], "| single", v["single_slot"]["value"], "| main ms", v["kernel_ms"], v["kernel_layout"])
PY
done

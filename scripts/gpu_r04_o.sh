set -e
OUT=gpurun_out/${TAG:-r04o}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_routing.py tests/test_gpu_slots.py tests/test_gpu_guards.py tests/test_gpu_dsoft.py tests/test_gpu_scheduling.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
timeout -k 10 400 python bench.py --only-variants --no-cpu > $OUT/variants_$i.json 2> $OUT/v.err || { tail -5 $OUT/v.err; exit 1; }
python -c "
import json; d=json.load(open('$OUT/variants_$i.json'))
for v in d['variants']: print(v['variant'][:40], v['value'], v['single_slot']['value'], v['kernel_layout'], v['raw_byte_candidates'])"
done

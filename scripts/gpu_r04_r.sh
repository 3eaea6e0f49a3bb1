set -e
OUT=gpurun_out/${TAG:-r04r}
mkdir -p $OUT
for i in 1 2; do
timeout -k 10 400 python bench.py --only-variants --no-cpu > $OUT/variants_$i.json 2> $OUT/v.err || { tail -5 $OUT/v.err; exit 1; }
python -c "
import json; d=json.load(open('$OUT/variants_$i.json'))
for v in d['variants'][:1]: print(v['variant'][:40], v['value'], v['single_slot'], v['kernel_layout'], v['raw_byte_candidates'], v['kernel_ms'], v['seed_kernel_ms'])"
done

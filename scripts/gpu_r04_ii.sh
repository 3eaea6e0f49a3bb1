set -e
OUT=gpurun_out/${TAG:-r04ii}
mkdir -p $OUT
for rep in 1 2; do
for s in 4 6 8; do
  GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python bench.py --no-cpu --no-others --slots $s --steps 16 --warmup 8 > $OUT/b_${s}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python -c "
import json; d=json.load(open('$OUT/b_${s}_$rep.json')); print('8 hardware queues, slots $s value', d['value'], d['ms_per_step'])"
done
done

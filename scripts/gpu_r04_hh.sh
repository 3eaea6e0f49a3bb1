# steps in flight: 3, 4 (default), 5, 6 slots; interleaved
set -e
OUT=gpurun_out/${TAG:-r04hh}
mkdir -p $OUT
for rep in 1 2; do
for s in 4 3 5 6; do
  timeout -k 10 300 python bench.py --no-cpu --no-others --slots $s --steps 12 --warmup 6 > $OUT/b_${s}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python -c "
import json; d=json.load(open('$OUT/b_${s}_$rep.json')); print('slots $s value', d['value'], d['ms_per_step'])"
done
done

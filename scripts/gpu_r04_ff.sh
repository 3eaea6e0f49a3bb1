set -e
OUT=gpurun_out/${TAG:-r04ff}
mkdir -p $OUT
for rep in 1 2; do
for v in "default:X=1" "two:GACT_HIP_WIDE_BLOCKS_PER_CU=2"; do
name=${v%%:*}; e=${v#*:}
env $e timeout -k 10 300 python bench.py --workload ont --no-cpu --no-others --steps 8 --warmup 4 > $OUT/b_${name}_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
python -c "
import json; d=json.load(open('$OUT/b_${name}_$rep.json')); print('$name ont', d['value'], d['ms_per_step'], d['single_slot']['value'], d['single_slot']['ms_per_step'], d['roofline']['kernel_ms'])"
done
done
timeout -k 10 300 python tools/side_probe.py pacbio50mb ont | cut -c1-200

set -e
OUT=gpurun_out/${TAG:-r04ee}
mkdir -p $OUT
timeout -k 10 400 python -m pytest "tests/test_gpu_configs.py::test_config5_ont_every_candidate" tests/test_gpu_slots.py tests/test_gpu_scheduling.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
timeout -k 10 300 python bench.py --workload ont --no-cpu --no-others --steps 8 --warmup 4 > $OUT/b_$rep.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
python -c "
import json; d=json.load(open('$OUT/b_$rep.json')); print('ont', d['value'], d['ms_per_step'], d['single_slot'], d['roofline']['kernel_ms'])"
done

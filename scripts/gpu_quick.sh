# quick check of a build: chain / golden / property tests, then the three workloads without the CPU baseline
set -e
mkdir -p gpurun_out/quick
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -4
for w in ecoli10x ont pacbio50mb; do
timeout -k 10 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu > gpurun_out/quick/$w.json 2> gpurun_out/quick/$w.err; python -c "
import json;d=json.load(open('gpurun_out/quick/$w.json'));r=d['roofline'];print('$w',d['value'],d['ms_per_step'],r['kernel_ms'],r['seed_kernel_ms'])"
done

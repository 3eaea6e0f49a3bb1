set -e
mkdir -p gpurun_out/r02e
for b in 0 2 1; do
GACT_HIP_WIDE_BLOCKS_PER_CU=$b timeout -k 10 300 python bench.py --workload ont --steps 4 --warmup 1 --no-cpu > gpurun_out/r02e/bench_ont_b$b.json 2> gpurun_out/r02e/bench_ont_b$b.err; cut -c1-400 gpurun_out/r02e/bench_ont_b$b.json
done
GACT_HIP_NO_WIDE=1 timeout -k 10 300 python bench.py --workload ont --steps 4 --warmup 1 --no-cpu > gpurun_out/r02e/bench_ont_narrow.json 2> gpurun_out/r02e/bench_ont_narrow.err; cut -c1-400 gpurun_out/r02e/bench_ont_narrow.json
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err; cut -c1-400 gpurun_out/r02e/bench.json

set -e
mkdir -p gpurun_out/r02c
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_properties.py tests/test_gpu_tiles.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02c/pytest.log; cat gpurun_out/r02c/pytest.log
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err; cat gpurun_out/r02c/bench.json
timeout -k 10 300 python bench.py --workload ont --steps 5 --warmup 2 > gpurun_out/r02c/bench_ont.json 2> gpurun_out/r02c/bench_ont.err; cat gpurun_out/r02c/bench_ont.json
timeout -k 10 300 python bench.py --workload pacbio50mb --steps 5 --warmup 2 > gpurun_out/r02c/bench_pacbio50mb.json 2> gpurun_out/r02c/bench_pacbio50mb.err; cat gpurun_out/r02c/bench_pacbio50mb.json

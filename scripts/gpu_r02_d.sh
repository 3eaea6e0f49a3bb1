set -e
mkdir -p gpurun_out/r02d
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r02d/pytest.log; cat gpurun_out/r02d/pytest.log
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err; cat gpurun_out/r02d/bench.json
timeout -k 10 300 python bench.py --workload ont --steps 5 --warmup 2 --no-cpu > gpurun_out/r02d/bench_ont.json 2> gpurun_out/r02d/bench_ont.err; cat gpurun_out/r02d/bench_ont.json
timeout -k 10 300 python bench.py --workload pacbio50mb --steps 5 --warmup 2 --no-cpu > gpurun_out/r02d/bench_pacbio50mb.json 2> gpurun_out/r02d/bench_pacbio50mb.err; cat gpurun_out/r02d/bench_pacbio50mb.json

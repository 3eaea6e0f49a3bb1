# round 2: parity of the linear-gap pass, then the bench
set -e
mkdir -p gpurun_out/r02b
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02b/pytest.log; cat gpurun_out/r02b/pytest.log
timeout -k 10 300 python bench.py > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err; cat gpurun_out/r02b/bench.json
GACT_HIP_NO_LIN=1 timeout -k 10 300 python bench.py --no-cpu > gpurun_out/r02b/bench_nolin.json 2> gpurun_out/r02b/bench_nolin.err; cat gpurun_out/r02b/bench_nolin.json

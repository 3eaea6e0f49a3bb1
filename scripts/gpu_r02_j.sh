set -e
mkdir -p gpurun_out/r02j
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -4
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/r02j/bench.json 2> gpurun_out/r02j/bench.err; cut -c60-140 gpurun_out/r02j/bench.json
timeout -k 10 300 python bench.py --workload ont --steps 4 --warmup 1 --no-cpu > gpurun_out/r02j/ont.json 2> gpurun_out/r02j/ont.err; cut -c60-140 gpurun_out/r02j/ont.json
timeout -k 10 300 python bench.py --workload pacbio50mb --steps 4 --warmup 1 --no-cpu > gpurun_out/r02j/pb.json 2> gpurun_out/r02j/pb.err; cut -c60-140 gpurun_out/r02j/pb.json
GACT_HIP_NO_LIN=1 timeout -k 10 300 python bench.py --no-cpu > gpurun_out/r02j/bench_nolin.json 2> gpurun_out/r02j/bench_nolin.err; cut -c60-140 gpurun_out/r02j/bench_nolin.json
timeout -k 10 600 python tools/stamps.py ecoli10x > gpurun_out/r02j/stamps_ecoli.txt 2>&1; head -12 gpurun_out/r02j/stamps_ecoli.txt

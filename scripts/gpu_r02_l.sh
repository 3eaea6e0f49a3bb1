set -e
mkdir -p gpurun_out/r02l
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -4
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/r02l/bench.json 2> gpurun_out/r02l/bench.err; cut -c60-140 gpurun_out/r02l/bench.json
timeout -k 10 300 python bench.py --workload ont --steps 4 --warmup 1 --no-cpu > gpurun_out/r02l/ont.json 2> gpurun_out/r02l/ont.err; cut -c60-140 gpurun_out/r02l/ont.json
timeout -k 10 300 python bench.py --workload pacbio50mb --steps 4 --warmup 1 --no-cpu > gpurun_out/r02l/pb.json 2> gpurun_out/r02l/pb.err; cut -c60-140 gpurun_out/r02l/pb.json
timeout -k 10 600 python tools/stamps.py ecoli10x > gpurun_out/r02l/stamps_ecoli.txt 2>&1; head -9 gpurun_out/r02l/stamps_ecoli.txt

set -e
mkdir -p gpurun_out/r02i
timeout -k 10 600 python tools/stamps.py ecoli10x > gpurun_out/r02i/stamps_ecoli.txt 2>&1; cat gpurun_out/r02i/stamps_ecoli.txt
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -5

set -e
mkdir -p gpurun_out/r02k
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py -x -q -m gpu -k "geometry" 2>&1 | tail -4
timeout -k 10 600 python tools/stamps.py ecoli10x > gpurun_out/r02k/stamps_ecoli.txt 2>&1; head -9 gpurun_out/r02k/stamps_ecoli.txt

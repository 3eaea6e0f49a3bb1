set -e
mkdir -p gpurun_out/r02g
timeout -k 10 300 python bench.py --no-cpu --steps 8 > gpurun_out/r02g/b.json 2> gpurun_out/r02g/b.err; cut -c60-140 gpurun_out/r02g/b.json
GACT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/ab/libgact_b32.so timeout -k 10 300 python bench.py --no-cpu --steps 8 > gpurun_out/r02g/b32.json 2> gpurun_out/r02g/b32.err; cut -c60-140 gpurun_out/r02g/b32.json
timeout -k 10 300 python bench.py --workload ont --steps 4 --warmup 1 --no-cpu > gpurun_out/r02g/ont.json 2> gpurun_out/r02g/ont.err; cut -c60-140 gpurun_out/r02g/ont.json
timeout -k 10 300 python bench.py --workload pacbio50mb --steps 4 --warmup 1 --no-cpu > gpurun_out/r02g/pb.json 2> gpurun_out/r02g/pb.err; cut -c60-140 gpurun_out/r02g/pb.json
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -3

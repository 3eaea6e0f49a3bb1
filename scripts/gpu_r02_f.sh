set -e
mkdir -p gpurun_out/r02f
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu --steps 6 > gpurun_out/r02f/b64_$i.json 2> gpurun_out/r02f/b64_$i.err; cut -c60-140 gpurun_out/r02f/b64_$i.json
GACT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/ab/libgact_b32.so timeout -k 10 300 python bench.py --no-cpu --steps 6 > gpurun_out/r02f/b32_$i.json 2> gpurun_out/r02f/b32_$i.err; cut -c60-140 gpurun_out/r02f/b32_$i.json
done

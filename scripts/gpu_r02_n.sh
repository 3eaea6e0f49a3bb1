set -e
mkdir -p gpurun_out/r02n
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu --steps 8 > gpurun_out/r02n/b3_$i.json 2> gpurun_out/r02n/b3_$i.err; python -c "
import json;d=json.load(open('gpurun_out/r02n/b3_$i.json'));print('3 blocks/CU',d['value'],d['roofline']['kernel_ms'],d['roofline']['seed_kernel_ms'],d['roofline']['waves_per_cu'])"
GACT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/ab/libgact_lin4.so timeout -k 10 300 python bench.py --no-cpu --steps 8 > gpurun_out/r02n/b4_$i.json 2> gpurun_out/r02n/b4_$i.err; python -c "
import json;d=json.load(open('gpurun_out/r02n/b4_$i.json'));print('4 blocks/CU',d['value'],d['roofline']['kernel_ms'],d['roofline']['seed_kernel_ms'])"
done
GACT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/ab/libgact_lin4.so timeout -k 10 300 python bench.py --workload pacbio50mb --steps 4 --warmup 1 --no-cpu > gpurun_out/r02n/pb4.json 2> gpurun_out/r02n/pb4.err; cut -c60-140 gpurun_out/r02n/pb4.json
GACT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/ab/libgact_lin4.so timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -3

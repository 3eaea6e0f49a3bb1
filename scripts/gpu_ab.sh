# A/B of two builds of the library on the three workloads: GACT_HIP_LIB_PATH=<other .so> for the B leg
set -e
mkdir -p gpurun_out/ab
B=${1:-darwin-gpu_amd/libgact_hip_b4.so}
for w in ${WORKLOADS:-ecoli10x ont pacbio50mb}; do
for leg in A B A B; do
if [ $leg = B ]; then export GACT_HIP_LIB_PATH=$PWD/$B; else unset GACT_HIP_LIB_PATH; fi
timeout -k 10 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu > gpurun_out/ab/$w.$leg.json 2> gpurun_out/ab/$w.$leg.err; python -c "
import json;d=json.load(open('gpurun_out/ab/$w.$leg.json'));r=d['roofline'];print('$w $leg',d['value'],d['ms_per_step'],r['kernel_ms'],r['seed_kernel_ms'])"
done
done

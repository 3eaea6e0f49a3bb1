# round 4: banded pointer stores -- parity of the chain / golden / property / poison files (default band and a narrow one that
# forces second runs), then bench.py one step at a time per band width
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-band}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_properties.py -x -q -m gpu --durations=5 2>&1 | tail -12 | tee $OUT/pytest_default.log
GACT_HIP_BAND=24 timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_poison.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_band24.log
for b in ${BANDS:-40 0 32 24 56}; do
  for w in ${WORKLOADS:-ecoli10x}; do
    GACT_HIP_BAND=$b timeout -k 10 300 python bench.py --workload $w --steps 8 --warmup 2 --no-cpu --no-others --slots 1 > $OUT/bench_${w}_band$b.json 2> $OUT/bench_${w}_band$b.err
    python -c "
import json;d=json.load(open('$OUT/bench_${w}_band$b.json'));r=d['roofline'];print('$w band $b: GCUPS',d['value'],'ms/step',d['ms_per_step'],'main',r['kernel_ms'],'seed',r['seed_kernel_ms'])"
  done
done

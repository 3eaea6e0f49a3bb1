# round 2, first GPU pass: the new tests, then configs 3 and 5 through bench.py with kernel traces
set -e
mkdir -p gpurun_out/r02a
timeout -k 10 900 python -m pytest tests/test_gpu_guards.py tests/test_gpu_shim.py tests/test_gpu_dsoft.py tests/test_gpu_configs.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r02a/pytest.log; cat gpurun_out/r02a/pytest.log
for w in ont pacbio50mb; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 > gpurun_out/r02a/bench_${w}_n1.json 2> gpurun_out/r02a/bench_${w}.err; cat gpurun_out/r02a/bench_${w}_n1.json
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in ont pacbio50mb; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02a/prof_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/r02a/prof_$w.json 2> $R/gpurun_out/r02a/prof_$w.err
done
find $R/gpurun_out/r02a -name "*kernel_stats*"

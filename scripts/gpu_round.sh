# one GPU round: the whole -m gpu suite, the default bench, rocprofv3 kernel traces of the three single-GPU
# workloads (BASELINE configs 2, 3, 5); results under gpurun_out/$TAG (copy what is to be judged into profiles/)
set -e
TAG=${TAG:-round}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -18 $OUT/pytest_gpu.log; fi
timeout -k 10 420 python bench.py > $OUT/bench_ecoli10x_n1.json 2> $OUT/bench_ecoli10x.err; cat $OUT/bench_ecoli10x_n1.json
cd /tmp && export TMPDIR=/tmp
# (--no-others: one workload per profile, so that a kernel's average duration in the stats is that workload's;
#  --slots 1: one step at a time, so that it is the kernel's own duration -- what roofline.kernel_ms reports; the default
#  run keeps four steps in flight and its kernels overlap)
for w in ecoli10x pacbio50mb ont; do
  rm -rf $OUT/prof_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python3 $R/bench.py --workload $w --no-others --no-config4 --no-reference-caller --slots 1 --steps 5 --warmup 2 --cpu-seconds 6 > $OUT/bench_${w}_profiled.json 2> $OUT/prof_$w.err
  cp $(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$w.csv
done
# the default command itself (four steps in flight): the kernels' durations here include the time they share the machine
rm -rf $OUT/prof_default
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 $R/bench.py --no-others --no-config4 --no-reference-caller --cpu-seconds 6 > $OUT/bench_default_profiled.json 2> $OUT/prof_default.err
cp $(find $OUT/prof_default -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_default_four_in_flight.csv
# the headline workload off its fastest kernels (bench.py's `variants`), all in one profile: the kernels differ by name
rm -rf $OUT/prof_variants
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_variants -- python3 $R/bench.py --only-variants --no-config4 --no-reference-caller --slots 1 --steps 1 --warmup 1 --no-cpu > $OUT/bench_variants_profiled.json 2> $OUT/prof_variants.err
cp $(find $OUT/prof_variants -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_variants.csv
ls $OUT

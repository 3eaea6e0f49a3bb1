# round 4, evidence part B: rocprofv3 kernel traces and PMC passes (results under gpurun_out/$TAG; copy what is to be judged into profiles/r04)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-r04_round_b}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (1) one step at a time in the plain sequence (seed launch, ONE main launch): a kernel's average duration is its own -- what roofline.kernel_ms reports
for w in ecoli10x pacbio50mb ont; do
  rm -rf $OUT/prof_$w
  GACT_HIP_NO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python3 $R/bench.py --workload $w --no-others --slots 1 --steps 5 --warmup 2 --no-cpu > $OUT/bench_${w}_one_at_a_time_profiled.json 2> $OUT/prof_$w.err
  cp $(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_${w}_one_at_a_time_plain_sequence.csv
done
# (2) the default command (four steps in flight; its one-at-a-time legs included)
rm -rf $OUT/prof_default
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 $R/bench.py --no-others --no-cpu > $OUT/bench_default_profiled.json 2> $OUT/prof_default.err
cp $(find $OUT/prof_default -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_ecoli10x_default_four_in_flight.csv
# (3) the variants (affine scorings, dirty reads, int32)
rm -rf $OUT/prof_variants
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_variants -- python3 $R/bench.py --only-variants --slots 1 --steps 1 --warmup 1 --no-cpu > $OUT/bench_variants_profiled.json 2> $OUT/prof_variants.err
cp $(find $OUT/prof_variants -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_variants.csv
# (4) PMC: per kernel (plain sequence, one launch of each kernel per pass) and of the default command
export GACT_HIP_NO_OVERLAP=1
for w in ecoli10x pacbio50mb ont; do
  PMC_OUT=${TAG:-r04_round_b}/pmc_$w WORKLOAD=$w bash $R/scripts/gpu_pmc.sh > $OUT/pmc_$w.log 2>&1 || tail -3 $OUT/pmc_$w.log
  python3 $R/tools/pmc_summary.py $OUT/pmc_$w > $OUT/pmc_$w.json
done
unset GACT_HIP_NO_OVERLAP
PMC_OUT=${TAG:-r04_round_b}/pmc_default WORKLOADS=ecoli10x bash $R/scripts/gpu_pmc_default.sh | tee $OUT/pmc_default.log
ls $OUT

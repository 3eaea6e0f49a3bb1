# the ONT-shape one-at-a-time profile and PMC passes again after the wide launch went to one block per CU
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-r04_ont_prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
w=ont
rm -rf $OUT/prof_$w
GACT_HIP_NO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python3 $R/bench.py --workload $w --no-others --slots 1 --steps 5 --warmup 2 --no-cpu > $OUT/bench_${w}_one_at_a_time_profiled.json 2> $OUT/prof_$w.err
cp $(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_${w}_one_at_a_time_plain_sequence.csv
export GACT_HIP_NO_OVERLAP=1
PMC_OUT=${TAG:-r04_ont_prof}/pmc_$w WORKLOAD=$w bash $R/scripts/gpu_pmc.sh > $OUT/pmc_$w.log 2>&1 || tail -3 $OUT/pmc_$w.log
python3 $R/tools/pmc_summary.py $OUT/pmc_$w > $OUT/pmc_$w.json
head -c 700 $OUT/pmc_$w.json; echo
head -3 $OUT/kernel_stats_${w}_one_at_a_time_plain_sequence.csv | cut -c1-200

# every counter pass the round-end roofline quotes, on the build in the tree: the main + seed launch of three workloads and of
# the affine variant (scripts/gpu_pmc.sh -> tools/pmc_summary.py) and the default command with steps in flight
# (scripts/gpu_pmc_default.sh).  Summaries land in gpurun_out/$TAG/: copy them to profiles/rNN/pmc_round_end_*.json
R=$GRAFT_REPO_ROOT
TAG=${TAG:-pmc_round_end}
mkdir -p $R/gpurun_out/$TAG
for w in ${WORKLOADS:-ecoli10x pacbio50mb ont}; do
  WORKLOAD=$w PMC_OUT=$TAG/raw_$w bash $R/scripts/gpu_pmc.sh > $R/gpurun_out/$TAG/$w.log 2>&1 && python3 $R/tools/pmc_summary.py $R/gpurun_out/$TAG/raw_$w > $R/gpurun_out/$TAG/pmc_round_end_$w.json && echo "$w ok"
done
WORKLOAD=ecoli10x EXTRA="--scoring 2,-3,-5,-2" PMC_OUT=$TAG/raw_variant_2 bash $R/scripts/gpu_pmc.sh > $R/gpurun_out/$TAG/variant_2.log 2>&1 && python3 $R/tools/pmc_summary.py $R/gpurun_out/$TAG/raw_variant_2 > $R/gpurun_out/$TAG/pmc_round_end_ecoli10x_variant_2.json && echo "variant_2 ok"
PMC_OUT=$TAG/raw_default WORKLOADS=ecoli10x bash $R/scripts/gpu_pmc_default.sh > $R/gpurun_out/$TAG/default.log 2>&1 && cp $R/gpurun_out/$TAG/raw_default/pmc_default_ecoli10x.json $R/gpurun_out/$TAG/pmc_default_round_end_ecoli10x.json && echo "default ok"
# the raw counter CSVs are large: keep the summaries
rm -rf $R/gpurun_out/$TAG/raw_*/sq $R/gpurun_out/$TAG/raw_*/sq2 $R/gpurun_out/$TAG/raw_*/fetch $R/gpurun_out/$TAG/raw_*/write $R/gpurun_out/$TAG/raw_default/ecoli10x
ls $R/gpurun_out/$TAG

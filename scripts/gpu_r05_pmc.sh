# round 5 counter passes (each counter set in its own rocprofv3 run, no tracing domains):
#   (1) the main launch of ecoli10x as one-wave-does-all and as DP + walker waves (GACT_HIP_ROLES=1)
#   (2) the wide linear-gap launch of the ONT shape at 1 / 2 / 3 blocks per CU (experiments build: GACT_HIP_WIDE_BLOCKS_PER_CU)
# usage: TAG=r05_pmc [PART=roles|ont|all] bash scripts/gpu_r05_pmc.sh
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r05_pmc}
B="python3 $R/bench.py --slots 1 --steps 1 --warmup 0 --no-cpu --no-others --no-config4 --no-reference-caller"
if [ "${PART:-all}" != ont ]; then
  for m in old roles; do
    if [ $m = roles ]; then envv="GACT_HIP_ROLES=1"; else envv="X=1"; fi
    PASSES="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES;SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
      CMD="$B --workload ecoli10x" ENVV="$envv" TAG=$TAG/ecoli10x_$m bash $R/scripts/gpu_pmc_any.sh
  done
fi
if [ "${PART:-all}" != roles ]; then
  for n in 1 2 3; do
    PASSES="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES;SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT;TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
      CMD="$B --workload ont" ENVV="GACT_HIP_LIB_PATH=$R/build/libgact_hip_exp.so GACT_HIP_WIDE_BLOCKS_PER_CU=$n" TAG=$TAG/ont_wide_${n}_per_cu bash $R/scripts/gpu_pmc_any.sh
  done
fi

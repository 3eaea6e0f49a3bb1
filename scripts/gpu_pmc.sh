# PMC passes for the dominant kernel (each counter set in its own run, no tracing domains)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PMC_OUT:-pmc_r02}
mkdir -p $OUT
# (EXTRA: e.g. "--scoring 2,-3,-5,-2" for the affine kernels; bench.py's roofline leg runs the plain sequence, one step at a time)
B="python3 $R/bench.py --workload ${WORKLOAD:-ecoli10x} --slots 1 --steps 1 --warmup 0 --no-cpu --no-others --no-config4 --no-reference-caller ${EXTRA}"   # one launch of each kernel per pass
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $B > $OUT/sq.json 2> $OUT/sq.err
echo sq done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.json 2> $OUT/fetch.err
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.json 2> $OUT/write.err
echo write done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -- $B > $OUT/sq2.json 2> $OUT/sq2.err
echo sq2 done
find $OUT -name "*counter_collection.csv" | head

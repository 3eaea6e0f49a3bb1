set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_x
rm -rf $OUT; mkdir -p $OUT
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $B > $OUT/sq.json 2> $OUT/sq.err
echo done

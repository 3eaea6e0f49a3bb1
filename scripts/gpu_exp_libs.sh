# timing experiments: tools/exp_time.py (no parity check) with each library of $LIBS in turn, $REPS times, interleaved
# usage: LIBS="build/libfw_base.so build/libfw_band32.so" WORKLOADS="ecoli10x" REPS=2 TAG=exp bash scripts/gpu_exp_libs.sh
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-exp}
mkdir -p $OUT
for w in ${WORKLOADS:-ecoli10x}; do
  for rep in $(seq 1 ${REPS:-2}); do
    for lib in $LIBS; do
      GACT_HIP_LIB_PATH=$R/$lib timeout -k 10 200 python $R/tools/exp_time.py $w >> $OUT/times.txt 2>> $OUT/err.txt
      tail -1 $OUT/times.txt
    done
  done
done

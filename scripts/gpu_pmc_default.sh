# the PMC pass of the DEFAULT command (steps in flight), for roofline.pipelined: counters only, no tracing domains
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PMC_OUT:-pmc_default}
mkdir -p $OUT
for w in ${WORKLOADS:-ecoli10x}; do
  rm -rf $OUT/$w
  timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/$w -- python3 $R/bench.py --workload $w --no-others --no-cpu --no-config4 --no-reference-caller > $OUT/bench_$w.json 2> $OUT/$w.err
  python3 $R/tools/pmc_default_summary.py $OUT/$w $OUT/bench_$w.json > $OUT/pmc_default_$w.json
  python3 -c "
import json;d=json.load(open('$OUT/pmc_default_$w.json'));print('$w', d['steps_profiled'],'steps,', '%.4g VALU instructions per step' % d['insts_valu_per_step'], {k:v['dispatches'] for k,v in d['kernels'].items()})"
done

# steps in flight (bench.py --slots): throughput of one workload at several numbers.  WORKLOADS="ont ecoli10x" SLOTS="4 6 8" bash scripts/gpu_slots.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-slots}
mkdir -p $OUT
for w in ${WORKLOADS:-ont ecoli10x pacbio50mb}; do
  for s in ${SLOTS:-4 6 8}; do
    timeout -k 10 280 python $R/bench.py --workload $w --slots $s --steps ${STEPS:-16} --warmup 4 --no-cpu --no-others > $OUT/${w}_slots$s.json 2> $OUT/${w}_slots$s.err || { echo "FAILED $w $s: $(tail -n 2 $OUT/${w}_slots$s.err)"; continue; }
    python -c "
import json;d=json.load(open('$OUT/${w}_slots$s.json'));print('$w slots=$s', d['value'], 'GCUPS', d['ms_per_step'], 'ms/step', 'single', (d.get('single_slot') or d['config'].get('single_slot') or {}).get('value'))"
  done
done

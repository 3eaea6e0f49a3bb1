# one workload under several settings of one environment switch: VAR=name VALS="a b c" WORKLOAD=ont
set -e
mkdir -p gpurun_out/sweep
w=${WORKLOAD:-ont}
for v in $VALS; do
export $VAR=$v
timeout -k 10 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu > gpurun_out/sweep/$w.$v.json 2> gpurun_out/sweep/$w.$v.err; python -c "
import json;d=json.load(open('gpurun_out/sweep/$w.$v.json'));r=d['roofline'];print('$w $VAR=$v',d['value'],d['ms_per_step'],r['kernel_ms'],r['seed_kernel_ms'])"
done

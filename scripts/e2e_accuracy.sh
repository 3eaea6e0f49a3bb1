# whole pipeline from FASTA on the config-2 read set, host filter and device filter:
# D-SOFT -> HIP GACT -> darwin.<t>.out; identical sorted lines; accuracy vs simulator truth
set -e
D=$(mktemp -d)
python - "$D" <<'PY'
import sys
sys.path.insert(0, "darwin-gpu_amd")
from gact_amd import synth, workload
cfg = dict(workload.CONFIGS["ecoli10x"]); seed = cfg.pop("seed")
rs = synth.simulate_reads(seed=seed, **cfg)
rs.write_fasta(sys.argv[1] + "/reads.fasta")
open(sys.argv[1] + "/params.cfg", "w").write(workload.PARAMS_CFG)
PY
cd $D
t0=$(date +%s%N)
$GRAFT_REPO_ROOT/darwin-gpu_amd/host/darwin_hip reads.fasta reads.fasta 8 | tail -3
t1=$(date +%s%N); echo "pipeline wall, host D-SOFT: $(( (t1 - t0) / 1000000 )) ms"
cat darwin.*.out | sort > host.sorted; rm darwin.*.out
t0=$(date +%s%N)
$GRAFT_REPO_ROOT/darwin-gpu_amd/host/darwin_hip reads.fasta reads.fasta 8 --device-dsoft | tail -3
t1=$(date +%s%N); echo "pipeline wall, device D-SOFT: $(( (t1 - t0) / 1000000 )) ms"
cat darwin.*.out | sort > device.sorted
cmp host.sorted device.sorted && echo "host-filter and device-filter outputs identical: $(wc -l < device.sorted) lines, $(uniq device.sorted | wc -l) unique"
python $GRAFT_REPO_ROOT/tools/measure_sensitivity.py reads.fasta darwin.*.out

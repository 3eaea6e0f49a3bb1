# whole pipeline from FASTA on the config-2 read set: D-SOFT restatement -> HIP GACT -> accuracy vs simulator truth
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
cd $D; SECONDS=0
$GRAFT_REPO_ROOT/darwin-gpu_amd/host/darwin_hip reads.fasta reads.fasta 8 | tail -4; echo "pipeline wall: $SECONDS s"
cat darwin.*.out | sort | uniq | wc -l
python $GRAFT_REPO_ROOT/tools/measure_sensitivity.py reads.fasta darwin.*.out

set -e
cat > /tmp/dirty_probe.py <<'PY'
import sys, os, json
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "darwin-gpu_amd"), os.path.join(ROOT, "oracle")]
import bench
from gact_amd import workload
blk = workload.make_block("ecoli10x", candidates="dsoft")
bench.SIDE_SLOTS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
o = bench.variant_config(bench.VARIANTS[0], "ecoli10x", blk.rs.reads, blk.cf, blk.cr)
print(json.dumps({k: o[k] for k in ("value", "ms_per_step", "single_slot", "kernel_ms", "seed_kernel_ms", "raw_byte_candidates")}))
PY
timeout -k 10 300 python -m pytest tests/test_gpu_routing.py tests/test_gpu_slots.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do timeout -k 10 300 python /tmp/dirty_probe.py 4; done

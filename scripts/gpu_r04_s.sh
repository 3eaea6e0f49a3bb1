set -e
OUT=gpurun_out/${TAG:-r04s}
mkdir -p $OUT
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
for v in "X=1" "GPU_MAX_HW_QUEUES=8" "GACT_HIP_NO_SIDE_LANE=1"; do
  for s in 4 1; do
  echo "== $v, engine of $s slot(s)"
  env $v timeout -k 10 300 python /tmp/dirty_probe.py $s
  done
done

# A/B of two (or more) builds of the engine in one GPU call: bench.py per workload with each library in turn
# (GACT_HIP_LIB_PATH), REPS times, interleaved (run-to-run noise on one box is +-3-4 %: compare medians and minima).
# usage: LIBS="darwin-gpu_amd/libgact_hip.so build/libB.so" WORKLOADS="ecoli10x ont" REPS=5 TAG=ab bash scripts/gpu_ab.sh
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-ab}
mkdir -p $OUT
for w in ${WORKLOADS:-ecoli10x ont}; do
  for rep in $(seq 1 ${REPS:-3}); do
    for lib in ${LIBS:-darwin-gpu_amd/libgact_hip.so}; do
      name=$(basename $lib .so)
      GACT_HIP_LIB_PATH=$R/$lib timeout -k 10 280 python $R/bench.py --workload $w --steps ${STEPS:-10} --warmup 3 --no-cpu --no-others > $OUT/${w}_${name}_$rep.json 2> $OUT/${w}_${name}_$rep.err
    done
  done
done
python - "$OUT" <<'PY'
import glob, json, os, sys
from collections import defaultdict
rows = defaultdict(list)
for p in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.load(open(p))
    except Exception:
        continue
    w, rest = os.path.basename(p)[:-5].split("_", 1)
    name = rest.rsplit("_", 1)[0]
    rows[(w, name)].append((d["roofline"]["kernel_ms"], d["roofline"]["seed_kernel_ms"], d["ms_per_step"], d["value"]))
for (w, name), v in sorted(rows.items()):
    main = sorted(x[0] for x in v); step = sorted(x[2] for x in v); g = sorted(x[3] for x in v)
    print("%-10s %-30s n=%d  main ms min %.2f med %.2f max %.2f | step ms min %.2f med %.2f | GCUPS med %.0f max %.0f | seed %.2f" % (
        w, name, len(v), main[0], main[len(main) // 2], main[-1], step[0], step[len(step) // 2], g[len(g) // 2], g[-1], sorted(x[1] for x in v)[len(v) // 2]))
PY

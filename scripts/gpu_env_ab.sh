# bench.py under several (library, environment) settings, REPS times interleaved.
# usage: CASES="name1|lib1|VAR=1 VAR2=2;name2|lib2|" WORKLOADS="ecoli10x" REPS=3 TAG=x bash scripts/gpu_env_ab.sh
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-envab}
mkdir -p $OUT
IFS=';' read -ra CS <<< "$CASES"
for w in ${WORKLOADS:-ecoli10x}; do
  for rep in $(seq 1 ${REPS:-3}); do
    for cs in "${CS[@]}"; do
      IFS='|' read -r name lib envs <<< "$cs"
      env GACT_HIP_LIB_PATH=$R/$lib $envs timeout -k 10 280 python $R/bench.py --workload $w --steps ${STEPS:-10} --warmup 3 --no-cpu --no-others > $OUT/${w}_${name}_$rep.json 2> $OUT/${w}_${name}_$rep.err || echo "FAILED $w $name $rep: $(tail -n 2 $OUT/${w}_${name}_$rep.err)"
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
    rows[(w, name)].append((d["roofline"]["kernel_ms"], d["roofline"]["seed_kernel_ms"], d["ms_per_step"], d["value"], d["roofline"]["kernel"]))
for (w, name), v in sorted(rows.items()):
    main = sorted(x[0] for x in v); step = sorted(x[2] for x in v); g = sorted(x[3] for x in v)
    print("%-10s %-24s n=%d  main ms min %.2f med %.2f max %.2f | step ms med %.2f | GCUPS med %.0f max %.0f | seed %.2f | %s" % (
        w, name, len(v), main[0], main[len(main) // 2], main[-1], step[len(step) // 2], g[len(g) // 2], g[-1], sorted(x[1] for x in v)[len(v) // 2], v[0][4]))
PY

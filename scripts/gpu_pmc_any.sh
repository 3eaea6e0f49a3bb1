# rocprofv3 --pmc passes of an arbitrary command; per pass a counter set (separate runs, no tracing domains).
# usage: PASSES="A B C;D E" CMD="python3 tools/exp_time.py ecoli10x" TAG=x [ENVV="GACT_HIP_LIB_PATH=..."] bash scripts/gpu_pmc_any.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-pmc_any}
mkdir -p $OUT
IFS=';' read -ra PS <<< "$PASSES"
n=0
for p in "${PS[@]}"; do
  n=$((n+1))
  rm -rf $OUT/pass$n
  (cd $R && env $ENVV timeout -k 10 ${PASS_TIMEOUT:-300} rocprofv3 --pmc $p --output-format csv -d $OUT/pass$n -- $CMD > $OUT/pass$n.log 2> $OUT/pass$n.err) || echo "pass $n failed: $(tail -n 3 $OUT/pass$n.err)"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for d in sorted(glob.glob(os.path.join(sys.argv[1], "pass*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
    for k, v in acc.items():
        if "extend" in k or "seed" in k:
            print(os.path.basename(d), k, {c: "%.4g per launch" % (x / max(cnt[(k, c)], 1)) for c, x in v.items()})
PY

# what the three layouts of a small run cost, for the time model of gact_policy.hpp (layout_times): the sweep of
# tools/sweep_policy.py at counts below the tile slots, once per forced layout (experiments build: GACT_HIP_WIDE_BLOCKS_PER_CU)
# usage: TAG=r05_calib bash scripts/gpu_layout_calibration.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-r05_calib}
mkdir -p $OUT
C="--lengths ${LENGTHS:-5000,10000,30000,100000} --counts ${COUNTS:-3000,6000,10000,15000,20000,24000}"
GACT_HIP_NO_WIDE=1 timeout -k 10 300 python $R/tools/sweep_policy.py $C --out $OUT/split.json > $OUT/split.log 2>&1; echo "split rc=$?"
for n in 1 2; do
  GACT_HIP_LIB_PATH=$R/build/libgact_hip_exp.so GACT_HIP_FORCE_WIDE=1 GACT_HIP_WIDE_BLOCKS_PER_CU=$n timeout -k 10 300 python $R/tools/sweep_policy.py $C --out $OUT/wide$n.json > $OUT/wide$n.log 2>&1; echo "wide$n rc=$?"
done
timeout -k 10 300 python $R/tools/sweep_policy.py $C --out $OUT/auto.json > $OUT/auto.log 2>&1; echo "auto rc=$?"
python3 - "$OUT" <<'PY'
import json, sys, os
o = sys.argv[1]
d = {m: {(r["read_length"], r["candidates"]): r for r in json.load(open(os.path.join(o, m + ".json")))["rows"]} for m in ("split", "wide1", "wide2", "auto")}
print("%7s %7s %6s | %8s %8s %8s | %8s %s" % ("readlen", "cands", "t/ch", "split", "wide1", "wide2", "auto", "auto layout"))
for k in sorted(d["split"]):
    r = d["split"][k]
    print("%7d %7d %6.1f | %8.2f %8.2f %8.2f | %8.2f %s" % (k[0], k[1], r["tiles_per_chain"], r["ms"], d["wide1"].get(k, {}).get("ms", 0), d["wide2"].get(k, {}).get("ms", 0),
                                                       d["auto"].get(k, {}).get("ms", 0), d["auto"].get(k, {}).get("layout", "")))
PY

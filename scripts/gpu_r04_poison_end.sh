# the default sweep with the traceback workspace poisoned before every launch (GACT_HIP_POISON_WS), round-end build
set -e
OUT=gpurun_out/${TAG:-r04_poison_end}
mkdir -p $OUT
GACT_HIP_POISON_WS=424242 timeout -k 10 900 python tools/stress_parity.py 250 8088 > $OUT/stress_parity_round_end_poison_250_8088.log 2>&1 || { tail -5 $OUT/stress_parity_round_end_poison_250_8088.log; exit 1; }
tail -n 1 $OUT/stress_parity_round_end_poison_250_8088.log | cut -c1-160

# random-configuration parity sweeps on the round-end build (tools/stress_parity.py): default, mixed (dirty reads, big tiles,
# two runs in flight), every read dirty
set -e
OUT=gpurun_out/${TAG:-r04_stress_end}
mkdir -p $OUT
timeout -k 10 700 python tools/stress_parity.py 300 3031 > $OUT/stress_parity_round_end_300_3031.log 2>&1 || { tail -5 $OUT/stress_parity_round_end_300_3031.log; exit 1; }
tail -n 1 $OUT/stress_parity_round_end_300_3031.log | cut -c1-160
STRESS_MIXED=1 timeout -k 10 500 python tools/stress_parity.py 200 515 > $OUT/stress_parity_round_end_mixed_200_515.log 2>&1 || { tail -5 $OUT/stress_parity_round_end_mixed_200_515.log; exit 1; }
tail -n 1 $OUT/stress_parity_round_end_mixed_200_515.log | cut -c1-160
STRESS_RAW_ONLY=1 timeout -k 10 400 python tools/stress_parity.py 120 77 > $OUT/stress_parity_round_end_raw_120_77.log 2>&1 || { tail -5 $OUT/stress_parity_round_end_raw_120_77.log; exit 1; }
tail -n 1 $OUT/stress_parity_round_end_raw_120_77.log | cut -c1-160

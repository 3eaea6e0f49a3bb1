set -e
OUT=gpurun_out/${TAG:-r04b}
mkdir -p $OUT
timeout -k 10 300 python tools/overlap_probe.py ecoli10x 8 | tee $OUT/overlap_ecoli.txt
timeout -k 10 300 python tools/overlap_probe.py pacbio50mb 3 | tee $OUT/overlap_pacbio.txt
GACT_HIP_TRACE=1 timeout -k 10 600 python -m pytest "tests/test_gpu_configs.py::test_config2_eight_feeder_slots" -x -q -m gpu -s 2>&1 | grep -v "^\[gact_hip\] \(pass\|  \|seed\|main\)" | tail -8 | tee $OUT/feeders.log

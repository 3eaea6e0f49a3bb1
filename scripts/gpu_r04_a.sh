set -e
OUT=gpurun_out/${TAG:-r04a}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_slots.py tests/test_gpu_routing.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.log
timeout -k 10 300 python tools/overlap_probe.py ecoli10x 8 | tee $OUT/overlap_ecoli.txt
timeout -k 10 300 python tools/overlap_probe.py pacbio50mb 3 | tee $OUT/overlap_pacbio.txt
timeout -k 10 600 python -m pytest "tests/test_gpu_configs.py::test_config2_eight_feeder_slots" -x -q -m gpu -s 2>&1 | tail -5 | tee $OUT/feeders.log

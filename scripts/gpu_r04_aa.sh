set -e
timeout -k 10 400 python -m pytest tests/test_gpu_scheduling.py -x -q -m gpu 2>&1 | tail -12
for i in 1 2 3; do
timeout -k 10 300 python -m pytest "tests/test_gpu_scheduling.py::test_a_group_of_feeder_threads_that_came_apart_joins_again" "tests/test_gpu_configs.py::test_config2_eight_feeder_slots" -x -q -m gpu 2>&1 | tail -2
done
timeout -k 10 300 python -m pytest tests/test_gpu_shim.py tests/test_reference_caller.py tests/test_gpu_slots.py -x -q -m gpu 2>&1 | tail -2

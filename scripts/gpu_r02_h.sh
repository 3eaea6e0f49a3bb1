set -e
mkdir -p gpurun_out/r02h
timeout -k 10 600 tools/issue_probe > gpurun_out/r02h/issue_rate_probe.json 2> gpurun_out/r02h/probe.err; tail -c 600 gpurun_out/r02h/issue_rate_probe.json
timeout -k 10 300 python bench.py --steps 5 --cpu-seconds 5 > gpurun_out/r02h/b.json 2> gpurun_out/r02h/b.err; cat gpurun_out/r02h/b.json
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -15

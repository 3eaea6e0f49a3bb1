# one GPU round: tests, bench, rocprofv3 kernel trace of the same bench command
set -e
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout -k 10 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; cat gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r01
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err
find $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -name "*stats*" | head

set -e
for v in "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=16"; do
echo "== $v: pacbio50mb, ont in one process"
env $v timeout -k 10 500 python tools/side_probe.py pacbio50mb ont | cut -c1-200
done

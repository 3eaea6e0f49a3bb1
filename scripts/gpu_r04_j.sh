set -e
echo "== pacbio50mb, ont, ont in one process"
timeout -k 10 500 python tools/side_probe.py pacbio50mb ont ont
echo "== pacbio50mb, 20 s idle, ont"
timeout -k 10 500 python tools/side_probe.py pacbio50mb sleep20 ont

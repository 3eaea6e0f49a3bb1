#!/bin/bash
# registers, spills and scratch of every kernel of the engine (hipcc -S, device side): tools/kernel_regs.sh [extra -D flags]
cd "$(dirname "$0")/.." && mkdir -p build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -S --cuda-device-only "$@" -o build/engine.s darwin-gpu_amd/csrc/gact_engine.hip 2>/dev/null
python3 - <<'PY'
import re
txt = open("build/engine.s").read()
for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
    name = m.group(1)
    import subprocess
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("gact::", "").replace("void ", "")
    print("%-75s vgpr %3s spill %2s | sgpr %3s spill %3s | scratch %4s B" % (short[:75], m.group(5), m.group(6), m.group(3), m.group(4), m.group(2)))
PY

"""bench.py's side configurations in a fresh process, one after the other: is a later one slower because of what ran before
it in the process (allocations, engine state) or because the chip is warm?
python tools/side_probe.py ont | python tools/side_probe.py pacbio50mb sleep20 ont"""
import json
import os
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "Power", "Temperature (Sensor junction)", "Temperature (Sensor edge)"))]
        return keep[:6]
    except Exception as err:
        return [str(err)]


args = types.SimpleNamespace(candidates="dsoft")
for name in sys.argv[1:] or ["ont"]:
    if name.startswith("sleep"):
        time.sleep(float(name[5:]))
        print(json.dumps({"slept": float(name[5:]), "clocks": clocks()}))
        continue
    o = bench.side_config(name, args)
    print(json.dumps({k: o[k] for k in ("workload", "value", "ms_per_step", "single_slot", "kernel_layout", "kernel_ms")} | {"clocks_after": clocks()}))
    sys.stdout.flush()

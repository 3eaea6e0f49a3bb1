"""Rewrites INTEGRATION.md section 7 from the library's own option table (gact_hip_options_describe); tests/test_cabi.py
checks that the two agree.  python tools/options_table_md.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
from gact_amd import engine

rows = engine.options_table()
md = ["\n## 7. Switches\n",
      "Every switch `libgact_hip.so` reads is in one table in `csrc/gact_engine.hip` (`kOptions`); `gact_hip_options_describe`\n"
      "returns it without an engine or a device, `tests/test_cabi.py` checks that this section and the library agree, and that no\n"
      "`getenv` of a `GACT_HIP_*` name exists in the library outside it.  None of them changes a record.  *When*: `create` = read\n"
      "once from the environment variable in `gact_hip_create`; `live` = `gact_hip_set_option(e, name, value)` on a running engine\n"
      "(where a variable is named too it gives the initial value).  *Class*: `kernels` = which kernels run (how the tests reach every\n"
      "kernel family); `scheduling`; `diagnostic` = tuning and tracing, read only by builds with `-DGACT_EXPERIMENTS` (which say so on\n"
      "stderr in `gact_hip_create`) and ignored by the default build.  (This table is written by `tools/options_table_md.py`.)\n",
      "| name | environment variable | when | class | what it does |", "|---|---|---|---|---|"]
for name, env, when, klass, doc in rows:
    md.append("| `%s` | %s | %s | %s | %s |" % (name, "`%s`" % env if env else "—", "live" if when.startswith("live") else "create",
                                          klass.split(" ")[0], doc.replace("|", "\\|")))
md.append("")
md.append("Outside the library: the C++ shim (`host/gact_shim.cpp`) reads `GACT_HIP_DEVICE` (the GPU `GPU_init` takes, default 0),\n"
          "`GACT_HIP_PAIR_STRANDS=1` (a feeder thread's two `GACT_Batch` calls as one run, §2) and `GACT_HIP_TIME=1` (the reference's\n"
          "`-D TIME` line, `gact.cpp:554-558`); the Python binding used by tests and `bench.py` reads `GACT_HIP_LIB_PATH` (another build of\n"
          "the library, for A/B runs).  Compile-time, measurements only: `-DGACT_EXPERIMENTS`, `-DGACT_STAMPS`, `-DGACT_ROLE_DP_WAVES=<n>`,\n"
          "`-DGACT_LIN_MAX3=0`, `-DGACT_WALK_TEAM=0/1`, `-DGACT_EXP_FAKE_WALK` (DESIGN.md, \"Diagnostic switches\").\n")
path = os.path.join(ROOT, "INTEGRATION.md")
s = open(path).read()
if "\n## 7. Switches" in s:
    s = s[:s.index("\n## 7. Switches")]
open(path, "w").write(s.rstrip("\n") + "\n" + "\n".join(md))
print("INTEGRATION.md 7:", len(rows), "switches")

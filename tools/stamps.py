"""Diagnostic: per-phase shader-clock shares of extend_p16_kernel (build with -DGACT_STAMPS).
   hipcc ... -DGACT_STAMPS -o /tmp/libgact_hip_stamps.so ; python tools/stamps.py"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
import numpy as np
from gact_amd import engine, workload

lib = os.path.join(ROOT, "gpurun_out", "libgact_hip_stamps.so")
if not os.path.exists(lib):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGACT_STAMPS",
                           "-I" + os.path.join(ROOT, "include"), "-o", lib,
                           os.path.join(ROOT, "darwin-gpu_amd", "csrc", "gact_engine.hip")])
engine.LIB_PATH = lib
blk = workload.make_block(sys.argv[1] if len(sys.argv) > 1 else "ecoli10x")
eng = engine.Engine()
cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
nf, nr = len(blk.cf), len(blk.cr)
eng.candidates_upload(np.concatenate([blk.cf, blk.cr]))
out = (C.c_ulonglong * 8)()
eng.L.gact_hip_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
for rep in range(2):
    eng.candidates_run_mixed(nf + nr, rc_from=nf)
    rec = eng.candidates_fetch(nf + nr)
    eng.L.gact_hip_debug_stamps(eng.h, out)
    st = eng.last_run_stats()
    v = [int(x) for x in out]
    tot = sum(v[:6])
    names = ["pick", "load", "dp_pass", "store_wait", "traceback", "consume"]
    print("main %.1f ms; wave-iterations %d, pointer steps/iter %.1f" % (st["main_ms"], v[6], v[7] / max(v[6], 1)))
    for n, x in zip(names, v[:6]):
        print("  %-10s %5.1f %%   %8.0f clocks/iter" % (n, 100.0 * x / tot, x / max(v[6], 1)))

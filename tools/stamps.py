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

lib = os.environ.get("GACT_STAMPS_LIB") or os.path.join(ROOT, "darwin-gpu_amd", "libgact_hip_stamps.so")
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

# ---- timeline of the last main launch: when the queues ran dry, when the waves ended
nw = 3072
tl = (C.c_ulonglong * (4 * nw))()
eng.L.gact_hip_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
eng.L.gact_hip_debug_timeline(eng.h, tl, nw)
t = np.array(list(tl), dtype=np.float64).reshape(nw, 4)
cyc = (C.c_ulonglong * nw)()
eng.L.gact_hip_debug_wave_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
eng.L.gact_hip_debug_wave_cycles(eng.h, cyc, nw)
cyc = np.array(list(cyc), dtype=np.float64)
ok = (t[:, 2] > t[:, 0]) & (cyc > 0)
print("in-kernel clock (wave lifetimes: s_memtime cycles / s_memrealtime at 100 MHz): median %.0f MHz (p10 %.0f, p90 %.0f)" %
      tuple(np.percentile(cyc[ok] / (t[ok, 2] - t[ok, 0]) * 100.0, [50, 10, 90]).tolist()))
t = t[t[:, 2] > 0]
t0 = t[:, 0].min()
ms = lambda x: (x - t0) / 1e5          # 100 MHz
end = ms(t[:, 2]); empty = ms(t[t[:, 1] > 0][:, 1])
print("waves %d; launch span %.1f ms; queues first seen empty at %.1f ms (median %.1f)" %
      (len(t), end.max(), empty.min(), np.median(empty)))
print("wave end times: p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f ms" %
      tuple(np.percentile(end, [10, 50, 90, 99]).tolist() + [end.max()]))
edges = np.linspace(0, end.max(), 21)
alive = [(end > e).sum() for e in edges[:-1]]
print("waves still running at 0,5,..,95 % of the span:", alive)
print("iterations per wave: mean %.1f max %d" % (t[:, 3].mean(), t[:, 3].max()))

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "darwin-gpu_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
from gact_amd import engine, synth
import test_gpu_routing as t
dirty = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 and sys.argv[1] else ()
rs = t._dirty_reads(501, dirty)
cf, cr = synth.synth_candidates(rs, seed=502, min_overlap=300, false_frac=0.15)
eng = engine.Engine()
cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
cands = np.concatenate([cf, cr])
eng.candidates_upload(cands)
eng.candidates_run_mixed(len(cands), rc_from=len(cf))
got = eng.candidates_fetch(len(cands))
print("ok", dirty, eng.last_run_stats()["raw_candidates"], int(got["cells"].sum()))

"""GPU: the N>1 plumbing of bench.py on the one GPU a test box has -- torch (its bundled HIP runtime) and
libgact_hip.so in one fresh process, the nccl (= RCCL) process group, the record gather straight from the
engine's device memory.  The 8-GPU run itself is the driver's; the same code at world sizes 2 and 3 runs on
CPU under gloo (tests/test_dist_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


@pytest.mark.parametrize("slots,steps", [(4, 6), (4, 2), (1, 2)], ids=["four-in-flight", "fewer-steps-than-slots", "one-at-a-time"])
def test_bench_force_dist_as_a_child_process(slots, steps):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--workload", "ecoli10x_small",
                          "--steps", str(steps), "--warmup", "1", "--cpu-seconds", "2", "--slots", str(slots)], capture_output=True,
                         text=True, cwd=ROOT, env=_env(), timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["steps"] == steps and res["config"]["slots_in_flight"] == slots
    assert res["config"]["gathered_records"] == res["config"]["candidates"] > 2000
    assert res["parity"]["bit_exact"] is True and res["parity"]["checked_candidates"] > 100
    # the C-ABI's own RCCL gather (gact_hip_comm_*), made once more behind the timed region: the same lines
    cg = res["config"]["gather"]["c_abi_rccl_gather"]
    assert cg["ok"] is True and cg["equals_torch_distributed_gather"] is True, cg
    assert cg["records_per_rank"] == [res["config"]["candidates"]]


def _two_ranks_on_one_device(extra, timeout=900):
    """bench.py as two processes (world 2) on the one GPU of a test box: gloo on host tensors stands in for RCCL, which
    refuses two ranks on one device; everything else -- block building per rank, exchange, deal, engine per rank, steps in
    flight, the gather in launch order, the checksums -- is the code the driver's 8-GPU run executes"""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   LOCAL_WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-device"] + extra,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    outs = [p.communicate(timeout=timeout) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-1500:] + se[-3000:]
    return json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])


def test_two_ranks_on_one_device_weak_scaling_path():
    res = _two_ranks_on_one_device(["--workload", "ecoli10x_small", "--steps", "5", "--warmup", "1", "--cpu-seconds", "2", "--slots", "4"])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["slots_in_flight"] == 4
    assert "rehearsal" in res["config"]
    g = res["config"]["gather"]
    assert g["ranks"] == 2 and len(g["records_per_rank"]) == 2 and min(g["records_per_rank"]) > 1000
    assert res["config"]["gathered_records"] == res["config"]["candidates"] == sum(g["records_per_rank"])
    assert len(g["crc32_per_rank"]) == 2 and res["parity"]["bit_exact"] is True


def test_two_ranks_on_one_device_strong_scaling_path():
    """--scaling strong: a FIXED job of eight blocks (small ones here), rank r builds blocks r, r + 2, ..., four exchange rounds,
    the merged list dealt over the two ranks"""
    res = _two_ranks_on_one_device(["--scaling", "strong", "--strong-blocks-of", "ecoli10x_small", "--steps", "3", "--warmup", "1", "--slots", "2"])
    assert res["n_gpus"] == 2 and res["scaling"] == "strong"
    c4 = res["config4_strong"]
    assert c4["n_gpus"] == 2 and len(c4["per_rank_ms_per_step"]["all"]) == 2 and c4["gather_ms_per_step_rank0"] is not None
    assert c4["candidates"] > 8 * 2000 and c4["cells_per_step"] > 0 and len(c4["build_seconds_per_rank"]) == 2


SCRIPT = r"""
import sys
sys.path[:0] = [%r, %r]
import torch                      # first: one HIP runtime for torch and the engine
import torch.distributed as dist
import numpy as np
from gact_amd import dist as gdist, engine, synth
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
rs = synth.simulate_reads(20000, n_reads=12, seed=5, mean_len=4000, sd_len=800, min_len=800, max_len=7000)
cf, cr = synth.synth_candidates(rs, seed=6, min_overlap=300)
blocks = gdist.exchange_blocks(dist, (rs.reads, cf, cr), 1)
assert len(blocks) == 1
# the tensor form of the exchange (what ranks > 1 use), looped back through the nccl group
payload, sizes = gdist._pack_block((rs.reads, cf, cr))
t = torch.from_numpy(payload).cuda(); outs = [torch.empty_like(t)]
dist.all_gather(outs, t)
reads2, cf2, cr2 = gdist._unpack_block(outs[0].cpu().numpy(), sizes, cf.dtype)
assert all(np.array_equal(a, b) for a, b in zip(reads2, rs.reads)) and cf2.tobytes() == cf.tobytes() and cr2.tobytes() == cr.tobytes()
eng = engine.Engine()
cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
n = len(cf) + len(cr)
eng.candidates_upload(np.concatenate([cf, cr]))
g = gdist.RecordGather(torch, dist, n, engine.OVERLAP_DTYPE.itemsize, 0, 1, "cuda")
gl = gdist.RecordGather(torch, dist, n, gdist.LINE_BYTES, 0, 1, "cuda")          # what bench.py gathers: 32-byte lines
dev = gdist.DeviceRecords(eng.device_overlaps_ptr(0), n, engine.OVERLAP_DTYPE.itemsize)
for _ in range(3):
    eng.candidates_run_mixed(n, rc_from=len(cf)); eng.sync(0)
    got = g.to_host(g(dev), engine.OVERLAP_DTYPE)[0]
    lines = gl.to_host(gl(dev), gdist.LINE_DTYPE)[0]
    want = eng.candidates_fetch(n)
    assert got.tobytes() == want.tobytes() and want["n_tiles"].sum() > n
    assert lines.tobytes() == gdist.lines_from_overlaps(want).tobytes()
    assert ((lines["comp_emitted"] >> 1) == want["emitted"]).all() and want["emitted"].sum() > 0
host = gdist.gather_records(torch, dist, want, 0, 1, "cuda")[0]
assert host.tobytes() == want.tobytes()
eng.close()
dist.destroy_process_group()
print("GATHER_OK", n)
"""


def test_device_resident_gather_equals_host_fetch():
    code = SCRIPT % (os.path.join(ROOT, "darwin-gpu_amd"), os.path.join(ROOT, "oracle"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=_env(), timeout=600)
    assert out.returncode == 0 and "GATHER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]

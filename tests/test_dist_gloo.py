"""N>1 path on CPU: two gloo ranks run the same deal / exchange / gather code bench.py
uses (gact_amd/dist.py); the per-rank compute is stood in for by the oracle, and rank 0's
gathered records must equal a single-process run over the whole candidate list."""
import os
import socket
import sys

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "darwin-gpu_amd"), os.path.join(root, "oracle")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gact_amd import dist as gdist, synth, workload
    import oracle_py
    blk = workload.make_block("tiny", block=rank, candidates="synthetic")
    blocks = gdist.exchange_blocks(dist, (blk.rs.reads, blk.cf, blk.cr), world)
    reads, cf_all, cr_all = gdist.merge_blocks(blocks)
    offs = np.zeros(len(reads) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    cat = np.concatenate(reads)
    rcat = np.concatenate([synth.revcomp(r) for r in reads])
    orc = oracle_py.Oracle()
    my_cf, my_cr = gdist.deal(cf_all, rank, world), gdist.deal(cr_all, rank, world)
    rf, _ = orc.gact_many(cat, offs, cat, offs, my_cf, complement=False)
    rr, _ = orc.gact_many(cat, offs, rcat, offs, my_cr, complement=True)
    mine = np.concatenate([rf, rr])
    parts = gdist.gather_records(torch, dist, mine, rank, world, "cpu")
    # every rank's part as rank 0 received it, against that rank's own records (what bench.py checks at N > 1)
    gdist.verify_gathered(torch, dist, mine, parts, rank, world, "cpu")
    if rank == 0:
        nf = [len(cf_all[r::world]) for r in range(world)]
        got_f = gdist.undeal([p[:n] for p, n in zip(parts, nf)], len(cf_all))
        got_r = gdist.undeal([p[n:] for p, n in zip(parts, nf)], len(cr_all))
        want_f, _ = orc.gact_many(cat, offs, cat, offs, cf_all, complement=False)
        want_r, _ = orc.gact_many(cat, offs, rcat, offs, cr_all, complement=True)
        q.put((got_f.tobytes() == want_f.tobytes(), got_r.tobytes() == want_r.tobytes(),
               len(cf_all), len(cr_all), [len(p) for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_ranks_gather_equals_single_process(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok_f, ok_r, nf, nr, sizes = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok_f and ok_r and nf > 20 and nr > 20
    assert sum(sizes) == nf + nr and max(sizes) - min(sizes) <= 2


def test_deal_undeal_roundtrip():
    from gact_amd import dist as gdist
    x = np.arange(23)
    for w in (1, 2, 5, 8):
        assert np.array_equal(gdist.undeal([gdist.deal(x, r, w) for r in range(w)], len(x)), x)

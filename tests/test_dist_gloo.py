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


def _strong_worker(rank, world, port, q):
    """bench.py's strong-scaling job (config4_strong): a FIXED set of blocks, rank r builds the blocks b % world == r, the
    rounds of exchange_block_rounds put all of them on every rank, the merged list is dealt over the ranks"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "darwin-gpu_amd"))
    import zlib
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gact_amd import dist as gdist, workload
    NB = 5
    built = {}
    for b in range(rank, NB, world):
        blk = workload.make_block("tiny", block=b, candidates="synthetic")
        built[b] = (blk.rs.reads, blk.cf, blk.cr)
    blocks = gdist.exchange_block_rounds(dist, built, NB, rank, world)
    reads, cf_all, cr_all = gdist.merge_blocks(blocks)
    q.put((rank, len(reads), zlib.crc32(np.concatenate(reads).tobytes()), zlib.crc32(cf_all.tobytes()), zlib.crc32(cr_all.tobytes()),
           len(gdist.deal(cf_all, rank, world)) + len(gdist.deal(cr_all, rank, world))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_fixed_job_of_blocks_over_ranks(world):
    """every rank ends up with the job a single process builds, whatever the number of ranks (strong scaling)"""
    import zlib
    import torch.multiprocessing as mp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "darwin-gpu_amd"))
    from gact_amd import dist as gdist, workload
    single = []
    for b in range(5):
        blk = workload.make_block("tiny", block=b, candidates="synthetic")
        single.append((blk.rs.reads, blk.cf, blk.cr))
    reads, cf_all, cr_all = gdist.merge_blocks(gdist.exchange_block_rounds(None, dict(enumerate(single)), 5, 0, 1))
    want = (len(reads), zlib.crc32(np.concatenate(reads).tobytes()), zlib.crc32(cf_all.tobytes()), zlib.crc32(cr_all.tobytes()))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    rows = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in rows) == list(range(world))
    assert all(tuple(r[1:5]) == want for r in rows), (rows, want)
    assert sum(r[5] for r in rows) == len(cf_all) + len(cr_all)


def _pipeline_worker(rank, world, port, q):
    """bench.py's timed region at N > 1 with steps in flight, the engine stood in for by tagged records: every rank
    launches step k on slot k % S and completes (= gathers) the steps in launch order; rank 0 must receive, at its k-th
    gather, step k's records of EVERY rank."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "darwin-gpu_amd"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gact_amd import dist as gdist
    S, steps, n_mine = 4, 11, 50 + 7 * rank                  # ragged record counts: the gather pads
    slots = [np.zeros(n_mine, dtype=gdist.LINE_DTYPE) for _ in range(S)]
    gather = gdist.RecordGather(torch, dist, n_mine, gdist.LINE_BYTES, rank, world, "cpu")
    launched, seen = [0], []

    def launch(slot):
        k = launched[0]
        launched[0] += 1
        slots[slot]["ref_id"] = rank
        slots[slot]["score"] = 1000 * k + np.arange(n_mine)          # what step k "computed" on this rank

    def complete(slot):
        parts = gather(slots[slot])
        if rank == 0:
            got = gather.to_host(parts, gdist.LINE_DTYPE)
            seen.append([(int(g["ref_id"][0]), int(g["score"][0]) // 1000, len(g), bool((g["score"] % 1000 == np.arange(len(g))).all()))
                         for g in got])
        return slot

    gdist.run_pipelined(steps, S, launch, complete)
    if rank == 0:
        q.put(seen)
    dist.barrier()
    dist.destroy_process_group()


def test_steps_in_flight_gather_in_launch_order():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    seen = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(seen) == 11
    for k, parts in enumerate(seen):
        assert [p[0] for p in parts] == [0, 1]                       # rank r's part in place r
        assert all(p[1] == k for p in parts), (k, parts)             # ... and it is step k's, of every rank
        assert [p[2] for p in parts] == [50, 57] and all(p[3] for p in parts)


def test_run_pipelined_order():
    from gact_amd import dist as gdist
    for S in (1, 2, 4, 6):
        for n in (0, 1, 3, 4, 9):
            log = []
            gdist.run_pipelined(n, S, lambda s: log.append(("L", s)), lambda s: log.append(("C", s)))
            assert [x for x in log if x[0] == "L"] == [("L", k % S) for k in range(n)]
            assert [x for x in log if x[0] == "C"] == [("C", k % S) for k in range(n)]
            # a slot is never launched again before its step has been completed
            busy = set()
            for kind, s_ in log:
                if kind == "L":
                    assert s_ not in busy
                    busy.add(s_)
                else:
                    busy.remove(s_)


def test_deal_undeal_roundtrip():
    from gact_amd import dist as gdist
    x = np.arange(23)
    for w in (1, 2, 5, 8):
        assert np.array_equal(gdist.undeal([gdist.deal(x, r, w) for r in range(w)], len(x)), x)

"""Multi-GPU plumbing of the GACT stage: one process per GPU,
read sets replicated, the candidate list dealt round-robin, one gather of the
fixed-size overlap records to rank 0.  torch.distributed is only the
transport (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
import numpy as np


def deal(cands, rank, world):
    """rank's share of a candidate list, round-robin (chain lengths vary widely,
    contiguous ranges would not balance)"""
    return np.ascontiguousarray(cands[rank::world])


def undeal(parts, total):
    """inverse of deal(): parts[r] holds the records of cands[r::world]"""
    world = len(parts)
    out = np.zeros(total, dtype=parts[0].dtype)
    for r, p in enumerate(parts):
        out[r::world] = p
    return out


def merge_blocks(blocks):
    """blocks: list of (reads, cands_fwd, cands_rc) with block-local read ids.
    Returns (reads, cands_fwd, cands_rc) over the concatenated read list."""
    reads, cf_all, cr_all = [], [], []
    base = 0
    for rd, cf, cr in blocks:
        for c, dst in ((cf, cf_all), (cr, cr_all)):
            c = c.copy()
            c["ref_id"] += base
            c["query_id"] += base
            dst.append(c)
        reads.extend(rd)
        base += len(rd)
    return reads, np.concatenate(cf_all), np.concatenate(cr_all)


def exchange_blocks(dist, block, world):
    """every rank simulated one block; afterwards every rank holds all of them"""
    if world == 1:
        return [block]
    gathered = [None] * world
    dist.all_gather_object(gathered, block)
    return gathered


def gather_records(torch, dist, records, rank, world, device):
    """The one collective of the path: rank 0 receives every rank's overlap records.
    records: structured numpy array.  Returns list of arrays on rank 0, else None."""
    item = records.dtype.itemsize
    mine = np.ascontiguousarray(records).view(np.uint8).reshape(-1, item)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([mine.shape[0]], dtype=torch.int64, device=device))
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    buf = torch.zeros((mx, item), dtype=torch.uint8, device=device)
    if mine.shape[0]:
        buf[:mine.shape[0]] = torch.from_numpy(mine.copy()).to(device)
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, outs, dst=0)
    if rank != 0:
        return None
    return [o[:n].cpu().numpy().reshape(-1).view(records.dtype) for o, n in zip(outs, counts)]

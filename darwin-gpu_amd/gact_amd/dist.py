"""Multi-GPU plumbing of the GACT stage: one process per GPU,
read sets replicated, the candidate list dealt round-robin, one gather of the
fixed-size overlap records to rank 0.  torch.distributed is only the
transport (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
import numpy as np


def run_pipelined(n_steps, in_flight, launch, complete):
    """n_steps passes over a rank's candidates with `in_flight` of them launched before the oldest is completed: step k
    runs on slot k % in_flight, complete(slot) delivers the output of the step that ran there (for N > 1 that is where
    the one gather of the path sits, so every rank must complete in this same order).  Returns the last complete()."""
    last = None
    for k in range(n_steps):
        launch(k % in_flight)
        if k >= in_flight - 1:
            last = complete((k - (in_flight - 1)) % in_flight)
    for k in range(max(n_steps - (in_flight - 1), 0), n_steps):
        last = complete(k % in_flight)
    return last


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


def _pack_block(block):
    """(reads, cands_fwd, cands_rc) -> one flat uint8 buffer + its four section sizes"""
    reads, cf, cr = block
    offs = np.zeros(len(reads) + 1, dtype=np.int64)
    if len(reads):
        offs[1:] = np.cumsum([len(r) for r in reads])
    parts = [np.concatenate(reads).astype(np.uint8) if len(reads) else np.zeros(0, np.uint8),
             offs.view(np.uint8), np.ascontiguousarray(cf).view(np.uint8).reshape(-1),
             np.ascontiguousarray(cr).view(np.uint8).reshape(-1)]
    return np.concatenate(parts), np.array([len(x) for x in parts], dtype=np.int64)


def _unpack_block(buf, sizes, cand_dtype):
    a, b, c, d = (int(x) for x in sizes)
    cat = buf[:a]
    offs = buf[a:a + b].copy().view(np.int64)
    cf = buf[a + b:a + b + c].copy().view(cand_dtype)
    cr = buf[a + b + c:a + b + c + d].copy().view(cand_dtype)
    reads = [np.ascontiguousarray(cat[offs[k]:offs[k + 1]]) for k in range(len(offs) - 1)]
    return reads, cf, cr


def exchange_blocks(dist, block, world, torch=None, device="cpu"):
    """every rank simulated one block; afterwards every rank holds all of them.  Flat byte tensors through two
    all_gathers (section sizes, then the padded payloads) -- nothing is pickled."""
    if world == 1:
        return [block]
    if torch is None:
        import torch
    payload, sizes = _pack_block(block)
    all_sizes = [torch.zeros(4, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_sizes, torch.from_numpy(sizes).to(device))
    all_sizes = [t.cpu().numpy() for t in all_sizes]
    mx = max(int(t.sum()) for t in all_sizes)
    mine = torch.zeros(mx, dtype=torch.uint8, device=device)
    mine[:len(payload)] = torch.from_numpy(payload).to(device)
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine)
    cand_dtype = np.asarray(block[1]).dtype
    return [_unpack_block(o.cpu().numpy(), sz, cand_dtype) for o, sz in zip(outs, all_sizes)]


def exchange_block_rounds(dist, built, n_blocks, rank, world, torch=None, device="cpu"):
    """A FIXED job of n_blocks genome blocks over `world` ranks (strong scaling): rank r built the blocks b with b % world == r
    (`built`: {b: (reads, cands_fwd, cands_rc)}); afterwards every rank holds all n_blocks, in block order.  One
    exchange_blocks round per ceil(n_blocks / world): block k * world + r travels in round k from rank r, a rank with no block
    left sends an empty one."""
    from . import synth
    empty = ([], np.zeros(0, dtype=synth.CAND_DTYPE), np.zeros(0, dtype=synth.CAND_DTYPE))
    blocks = [None] * n_blocks
    for k in range((n_blocks + world - 1) // world):
        got = exchange_blocks(dist, built.get(k * world + rank, empty), world, torch=torch, device=device)
        for r, g in enumerate(got):
            if k * world + r < n_blocks:
                blocks[k * world + r] = g
    return blocks


class DeviceRecords:
    """The engine's device-resident overlap records (gact_hip_device_overlaps) as something torch can wrap without a
    copy: the CUDA array interface, which torch.as_tensor honours on ROCm as well."""

    def __init__(self, ptr, n, itemsize):
        self.__cuda_array_interface__ = {"shape": (int(n), int(itemsize)), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


# What travels in the gather: the eight numbers of an output line (gact.cpp:214-224), 32 bytes, with the engine's
# `emitted` flag (gact.cpp:213: is the line printed at all) folded into bit 1 of `comp`.  The engine's own record
# (gact_overlap, 56 bytes) carries bookkeeping (first-tile score, tile and cell counts) that rank 0 has no use for.
LINE_DTYPE = np.dtype([(n, "<i4") for n in ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp_emitted")])
LINE_BYTES = 32


def line_records(torch, src):
    """(n, 56) uint8 device tensor of gact_overlap records -> (n, 32) uint8 tensor of LINE_DTYPE records"""
    out = src[:, :LINE_BYTES].clone()
    out[:, 28] |= (src[:, 32] != 0).to(torch.uint8) << 1
    return out


def lines_from_overlaps(records):
    """the same on the host (CPU tests, single-process runs)"""
    out = np.zeros(len(records), dtype=LINE_DTYPE)
    for n in ("ref_id", "query_id", "ab", "ae", "bb", "be", "score"):
        out[n] = records[n]
    out["comp_emitted"] = records["comp"] | ((records["emitted"] != 0).astype(np.int32) << 1)
    return out


class RecordGather:
    """The one collective of the path: rank 0 receives every rank's overlap records.  The record counts
    are fixed once the candidates are dealt, so they are exchanged once; every step after that is one padded
    `gather` (RCCL over xGMI under the nccl backend, each peer's 32-56 B x count on its own link into rank 0)."""

    def __init__(self, torch, dist, n_mine, itemsize, rank, world, device):
        self.torch, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device
        self.itemsize, self.n_mine = itemsize, n_mine
        counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([n_mine], dtype=torch.int64, device=device))
        self.counts = [int(c.item()) for c in counts]
        mx = max(max(self.counts), 1)
        self.buf = torch.zeros((mx, itemsize), dtype=torch.uint8, device=device)
        self.outs = [torch.empty_like(self.buf) for _ in range(world)] if rank == 0 else None

    def __call__(self, records):
        """records: numpy structured array (host), or DeviceRecords (stays on the device: no host round trip in
        front of the collective).  Returns the per-rank record tensors on rank 0 (still on `device`), else None."""
        torch = self.torch
        if isinstance(records, DeviceRecords):
            src = torch.as_tensor(records, device=self.device)
            if self.itemsize == LINE_BYTES and src.shape[1] != LINE_BYTES:
                src = line_records(torch, src)           # engine records narrowed to the printable line, on the device
        else:
            src = torch.from_numpy(np.ascontiguousarray(records).view(np.uint8).reshape(-1, self.itemsize)).to(self.device)
        if self.n_mine:
            self.buf[:self.n_mine].copy_(src[:self.n_mine])
        self.dist.gather(self.buf, self.outs, dst=0)
        if self.device != "cpu":
            # the copy out of the engine's record array and the collective are queued on torch's stream; the engine's
            # next launch (its own non-blocking stream) rewrites that array: every rank waits here, not only rank 0
            torch.cuda.current_stream().synchronize()
        if self.rank != 0:
            return None
        return [o[:n] for o, n in zip(self.outs, self.counts)]

    def to_host(self, parts, dtype):
        return [p.cpu().numpy().reshape(-1).view(dtype) for p in parts]


def checksum_lines(lines):
    """order-sensitive checksum of LINE_DTYPE records (what a rank sent / what rank 0 received from it)"""
    import zlib
    return zlib.crc32(np.ascontiguousarray(lines).tobytes())


def verify_gathered(torch, dist, my_lines, gathered, rank, world, device):
    """every rank's own line records against what rank 0 holds of it after the gather: the per-rank checksums
    travel in one all_gather; rank 0 raises on the first rank whose part differs"""
    mine = torch.tensor([checksum_lines(my_lines)], dtype=torch.int64, device=device)
    sums = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sums, mine)
    if rank != 0:
        return True
    for r in range(world):
        if checksum_lines(gathered[r]) != int(sums[r].item()):
            raise RuntimeError("gathered records of rank %d differ from what that rank's engine holds" % r)
    return [int(x.item()) for x in sums]          # rank 0: every rank's checksum, as that rank computed it


def gather_records(torch, dist, records, rank, world, device):
    """one-shot form of RecordGather for host-side records; returns list of arrays on rank 0, else None"""
    records = np.ascontiguousarray(records)
    g = RecordGather(torch, dist, len(records), records.dtype.itemsize, rank, world, device)
    parts = g(records)
    return None if parts is None else g.to_host(parts, records.dtype)

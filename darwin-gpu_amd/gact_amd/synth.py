"""Deterministic PBSIM-shape read simulator and candidate synthesiser.

The reference's bundled 10x E.coli PBSIM ``reads.fasta`` is absent from the
mount (reference/.MISSING_LARGE_BLOBS, README:18), so every workload here is
synthetic and seeded:

* genome: iid uniform ACGT
* reads: length ~ N(mean, sd^2) clipped to [lo, hi], strand 50/50, per-base
  error rate split sub:ins:del (PBSIM CLR default 10:60:30)
* header ``>S<i>_<start>_<len>``, 70-column wrap (fasta.h:19, and the
  ``name_start_len`` convention of measure_sensitivity_PBSIM.py:20-22 /
  generateperfect.py:86-94)

``synth_candidates`` places D-SOFT-shaped seed hits (ref_id, query_id,
ref_pos, query_pos, comp) -- the tuple darwin.cpp:216-238 hands to GACT --
on true overlaps (plus a share of false hits), from the simulator's own
genome->read coordinate maps.  It stands in for the D-SOFT filter, which is
outside the hot path.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b

CAND_DTYPE = np.dtype([("ref_id", "<i4"), ("query_id", "<i4"),
                       ("ref_pos", "<i4"), ("query_pos", "<i4")])


def revcomp(seq):
    """RevComp of darwin.cpp:110-147 on a uint8 array."""
    return _COMP[np.asarray(seq, dtype=np.uint8)[::-1]]


class ReadSet:
    """reads[i] (uint8 ASCII), rc[i], plus truth for candidate synthesis."""

    def __init__(self):
        self.genome = None
        self.reads = []
        self.names = []
        self.start = []      # genome start of each read's span
        self.span = []       # genome span length
        self.strand = []     # 0 = '+', 1 = '-'
        self.gmap = []       # gmap[i][g - start] = read coordinate (in '+' orientation) of genome base g

    @property
    def n(self):
        return len(self.reads)

    def rc(self, i):
        return revcomp(self.reads[i])

    def concat(self, rc=False):
        seqs = [revcomp(r) for r in self.reads] if rc else self.reads
        offs = np.zeros(len(seqs) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(s) for s in seqs])
        return (np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)), offs

    def total_bases(self):
        return int(sum(len(r) for r in self.reads))

    def write_fasta(self, path):
        with open(path, "wb") as f:
            for name, r in zip(self.names, self.reads):
                f.write(b">" + name.encode() + b"\n")
                b = r.tobytes()
                for k in range(0, len(b), 70):
                    f.write(b[k:k + 70] + b"\n")


def simulate_reads(genome_len, n_reads=None, coverage=None, seed=1, mean_len=10000, sd_len=2000,
                   min_len=1000, max_len=25000, error=0.15, split=(10, 60, 30),
                   uniform_len=None, n_frac=0.0):
    """Returns a ReadSet.  Give n_reads or coverage.  uniform_len=(lo,hi) switches
    the length law to uniform (ONT-shape config 5).  n_frac sprinkles 'N's."""
    rng = np.random.default_rng(seed)
    rs = ReadSet()
    genome = _ACGT[rng.integers(0, 4, size=genome_len)]
    rs.genome = genome
    if n_reads is None:
        mean = mean_len if uniform_len is None else 0.5 * (uniform_len[0] + uniform_len[1])
        n_reads = max(1, int(round(coverage * genome_len / mean)))
    tot = float(sum(split))
    p_sub, p_ins, p_del = (error * s / tot for s in split)
    for i in range(n_reads):
        if uniform_len is None:
            L = int(np.clip(rng.normal(mean_len, sd_len), min_len, max_len))
        else:
            L = int(rng.integers(uniform_len[0], uniform_len[1] + 1))
        L = min(L, genome_len)
        st = int(rng.integers(0, genome_len - L + 1))
        g = genome[st:st + L]
        u = rng.random(L)
        deleted = u < p_del
        subbed = (u >= p_del) & (u < p_del + p_sub)
        base = g.copy()
        if subbed.any():
            # substitute by a different base
            idx = np.searchsorted(_ACGT_SORTED, base[subbed])
            base[subbed] = _ACGT_SORTED[(idx + rng.integers(1, 4, size=int(subbed.sum()))) % 4]
        ins = rng.random(L) < p_ins          # one inserted base before genome base k
        emit = (~deleted).astype(np.int64) + ins.astype(np.int64)
        ends = np.cumsum(emit)
        total = int(ends[-1]) if L else 0
        out = np.empty(total, dtype=np.uint8)
        begin = ends - emit
        ins_pos = begin[ins]
        out[ins_pos] = _ACGT[rng.integers(0, 4, size=len(ins_pos))]
        keep = ~deleted
        kept_pos = (begin + ins.astype(np.int64))[keep]
        out[kept_pos] = base[keep]
        # read coordinate of each genome base (deleted bases map to the next emitted one)
        gmap = (begin + ins.astype(np.int64)).astype(np.int32)
        if n_frac > 0 and total:
            out[rng.random(total) < n_frac] = ord("N")
        strand = int(rng.integers(0, 2))
        if strand:
            out = revcomp(out)
        rs.reads.append(np.ascontiguousarray(out))
        rs.names.append("S%d_%d_%d" % (i + 1, st, L))
        rs.start.append(st); rs.span.append(L); rs.strand.append(strand); rs.gmap.append(gmap)
    return rs


_ACGT_SORTED = np.frombuffer(b"ACGT", dtype=np.uint8)  # already sorted in ASCII


def _read_coord(rs, i, g, want_rc):
    """coordinate of genome base g in reads[i] (want_rc=False) or rc(reads[i])."""
    c = int(rs.gmap[i][g - rs.start[i]])
    n = len(rs.reads[i])
    c = min(c, n - 1)
    flipped = bool(rs.strand[i]) ^ bool(want_rc)
    return (n - 1 - c) if flipped else c


def synth_candidates(rs, seed=2, min_overlap=1000, per_pair=1.17, false_frac=0.03,
                     max_candidates=None, include_self=True):
    """D-SOFT-shaped candidates for the self-overlap run (same_file).

    Returns (cands_for, cands_rev): structured arrays CAND_DTYPE for the
    forward-strand queries (darwin.cpp:213-248) and the reverse-complemented
    queries (darwin.cpp:252-287), ordered by query id like AlignReads builds them.
    """
    rng = np.random.default_rng(seed)
    n = rs.n
    order = np.argsort(rs.start, kind="stable")
    starts = np.asarray(rs.start)[order]
    ends = starts + np.asarray(rs.span)[order]
    fwd, rev = [], []
    for qi in range(n):
        qs, qe = rs.start[qi], rs.start[qi] + rs.span[qi]
        hi = int(np.searchsorted(starts, qe))
        for k in range(hi):
            ri = int(order[k])
            if ends[k] <= qs:
                continue
            if ri == qi and not include_self:
                continue
            lo_g, hi_g = max(qs, rs.start[ri]), min(qe, rs.start[ri] + rs.span[ri])
            if hi_g - lo_g < min_overlap:
                continue
            comp = rs.strand[ri] != rs.strand[qi]
            reps = 1 + (rng.random() < (per_pair - 1.0))
            for _ in range(int(reps)):
                g = int(rng.integers(lo_g, hi_g))
                rp = _read_coord(rs, ri, g, False)
                qp = _read_coord(rs, qi, g, comp)
                (rev if comp else fwd).append((ri, qi, rp, qp))
        n_false = rng.random() < false_frac * 20
        if n_false:
            ri = int(rng.integers(0, n))
            rp = int(rng.integers(0, len(rs.reads[ri]) + 1))
            qp = int(rng.integers(0, len(rs.reads[qi])))
            (rev if rng.random() < 0.5 else fwd).append((ri, qi, rp, qp))
    cf = np.array(fwd, dtype=CAND_DTYPE) if fwd else np.zeros(0, CAND_DTYPE)
    cr = np.array(rev, dtype=CAND_DTYPE) if rev else np.zeros(0, CAND_DTYPE)
    if max_candidates is not None:
        cf, cr = cf[:max_candidates], cr[:max_candidates]
    return cf, cr

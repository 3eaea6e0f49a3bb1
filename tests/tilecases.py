"""Seeded tile cases shared by the CPU and GPU suites."""
import numpy as np

from gact_amd import synth

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def related_pair(rng, R, Q, err=0.15):
    """two noisy copies of one random template, cut to R and Q bases"""
    L = max(R, Q) + 40
    rs = synth.simulate_reads(L + 50, n_reads=2, seed=int(rng.integers(1 << 30)), mean_len=L, sd_len=1,
                              min_len=L, max_len=L, error=err)
    a = rs.reads[0] if not rs.strand[0] else synth.revcomp(rs.reads[0])
    b = rs.reads[1] if not rs.strand[1] else synth.revcomp(rs.reads[1])
    # both span almost the same genome stretch; align their starts roughly
    off = rs.start[0] - rs.start[1]
    if off > 0:
        b = b[off:]
    else:
        a = a[-off:]
    a, b = a[:R], b[:Q]
    return a.tobytes(), b.tobytes()


def random_tiles(seed, n, max_len=320, with_n=False):
    """list of (ref_bytes, query_bytes, reverse, first)"""
    rng = np.random.default_rng(seed)
    out = []
    special = [1, 2, 7, 8, 9, 15, 16, 17, 19, 20, 21, 39, 40, 41, 160, 199, 200, 201, 319, 320]
    for k in range(n):
        if k % 3 == 0:
            R = int(rng.choice(special)); Q = int(rng.choice(special))
        elif k % 3 == 1:
            R = Q = max_len
        else:
            R = int(rng.integers(1, max_len + 1)); Q = int(rng.integers(1, max_len + 1))
        R, Q = min(R, max_len), min(Q, max_len)
        kind = k % 7
        if kind == 6:
            a = bytes(ACGT[rng.integers(0, 4, R)]); b = bytes(ACGT[rng.integers(0, 4, Q)])
        elif kind == 5:
            a = b"A" * R; b = (b"A" * Q) if k % 2 else (b"C" * Q)
        else:
            a, b = related_pair(rng, R, Q, err=[0.15, 0.05, 0.3, 0.0, 0.12][kind])
        if with_n and len(a) > 3:
            a = bytearray(a); b = bytearray(b)
            for _ in range(1 + len(a) // 40):
                a[int(rng.integers(0, len(a)))] = ord("N")
            for _ in range(1 + len(b) // 40):
                b[int(rng.integers(0, len(b)))] = ord("N" if rng.random() < 0.7 else "a")
            a, b = bytes(a), bytes(b)
        out.append((a, b, int(k // 2 % 2), int(k % 2)))
    return out


# known-answer tiles measured from the compiled reference
# (AlignWithBT with scores 1,-1,-1,-1, early_terminate 200).  (ref, query, reverse, first, header, counts D/I/M)
KAT = [
    (b"A" * 320, b"C" * 320, 0, 1, [0, 320, 320], (0, 0, 0)),
    (b"A" * 37, b"C" * 101, 0, 1, [0, 37, 101], (0, 0, 0)),
    (b"A" * 320, b"C" * 320, 0, 0, [0], (0, 0, 0)),
    (b"N" * 50, b"N" * 50, 0, 1, [50, 50, 50], (0, 0, 50)),
    (b"a" * 50, b"A" * 50, 0, 1, [0, 50, 50], (0, 0, 0)),
    (b"A", b"A", 0, 1, [1, 1, 1], (0, 0, 1)),
    (b"A", b"C", 0, 0, [0], (0, 0, 0)),
    (b"ACGTTTTTACGT", b"ACGTCCCCACGT", 0, 1, [4, 12, 12], (0, 0, 4)),
]

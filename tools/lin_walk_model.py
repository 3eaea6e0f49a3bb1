"""Model check of the linear-gap pointer scheme (gact_lin.hpp): for gap_open == gap_extend == mismatch the
traceback of align.cpp:185-230 -- op codes plus the two open/extend flags -- visits the same states as a walk
that only follows the op code of every cell it enters (the tag of H in max(4M+3, 4(H_up+g)+2, 4(H_left+g)+1)),
with ZERO recognised from the running cell score after a diagonal move.
python tools/lin_walk_model.py [n_tiles] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle")]
import numpy as np
import oracle_py


def model(ref, query, match, g, reverse, first, early, ref_pos=None, query_pos=None):
    R, Q = len(ref), len(query)
    r = ref[::-1] if reverse else ref
    q = query[::-1] if reverse else query
    H = np.zeros((R + 1, Q + 1), dtype=np.int64)
    T = np.zeros((R + 1, Q + 1), dtype=np.int64)
    best, bi, bj = 0, 0, 0
    for i in range(1, R + 1):
        for j in range(1, Q + 1):
            s = match if r[i - 1] == q[j - 1] else g
            m = 4 * max(H[i - 1][j - 1] + s, 0) + 3
            up = 4 * (H[i - 1][j] + g) + 2
            le = 4 * (H[i][j - 1] + g) + 1
            x = max(m, up, le)
            H[i][j] = x >> 2
            T[i][j] = x & 3
            if H[i][j] >= best:
                best, bi, bj = H[i][j], i, j
    i, j = (bi, bj) if first else (R if ref_pos is None else ref_pos, Q if query_pos is None else query_pos)
    out = [best, bi, bj] if first else [H[i][j]]
    v = H[i][j]
    state = 0 if v == 0 else T[i][j]
    isteps = jsteps = 0
    while state != 0:
        if isteps >= early or jsteps >= early:
            break
        out.append(int(state))
        if state == 3:
            v -= match if r[i - 1] == q[j - 1] else g
            i -= 1; j -= 1; isteps += 1; jsteps += 1
            state = 0 if (v == 0 or i < 1 or j < 1) else T[i][j]
        elif state == 2:
            v -= g
            i -= 1; isteps += 1
            state = T[i][j] if i >= 1 else 3
        else:
            v -= g
            j -= 1; jsteps += 1
            state = T[i][j] if j >= 1 else 3
        assert i < 1 or j < 1 or v == H[i][j], "running score lost"
    return [int(x) for x in out]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    orc = oracle_py.Oracle()
    for it in range(n):
        R, Q = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        alpha = rng.choice([2, 3, 4])
        ref = bytes(rng.choice(list(b"ACGT"[:alpha]), R).tolist())
        if rng.random() < 0.6:           # a noisy copy: long paths with gaps
            qq = []
            for ch in ref:
                u = rng.random()
                if u < 0.12: continue
                if u < 0.24: qq.append(int(rng.choice(list(b"ACGT"[:alpha]))))
                qq.append(int(rng.choice(list(b"ACGT"[:alpha]))) if rng.random() < 0.1 else ch)
            query = bytes(qq[:Q]) or b"A"
        else:
            query = bytes(rng.choice(list(b"ACGT"[:alpha]), Q).tolist())
        match = int(rng.integers(0, 5)); g = -int(rng.integers(0, 4))
        reverse = bool(rng.integers(2)); first = bool(rng.integers(2))
        early = int(rng.choice([1, 5, 20, 200]))
        want = orc.align_with_bt(ref, query, scoring=(match, g, g, g), reverse=reverse, first=first, early_terminate=early)
        got = model(ref, query, match, g, reverse, first, early)
        if list(want) != got:
            print("MISMATCH", it, ref, query, match, g, reverse, first, early, "\n", list(want), "\n", got)
            sys.exit(1)
    print("linear-gap walk model: %d random tiles identical to align_with_bt" % n)


if __name__ == "__main__":
    main()

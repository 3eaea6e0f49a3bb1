"""ctypes bindings for the TEST-ONLY checkers under oracle/.

* ``Oracle``  -> oracle/liboracle.so   (C restatement, gact_oracle.c)
* ``RefLib``  -> oracle/_ref/libdarwin_ref.so (the reference's own align.cpp +
  gact.cpp compiled unchanged; exists only where /root/reference was present
  at build time, or as a prebuilt file on the GPU box)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class TileTrace(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "ref_off", "query_off", "ref_len", "query_len", "reverse", "first",
        "tile_score", "max_i", "max_j", "n_states", "i_steps", "j_steps")]


class Overlap(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp",
        "emitted", "first_tile_score", "n_tiles")] + [("cells", C.c_int64)]


class Candidate(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("ref_id", "query_id", "ref_pos", "query_pos")]


OVERLAP_DTYPE = np.dtype([(n, "<i4") for n in (
    "ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp",
    "emitted", "first_tile_score", "n_tiles")] + [("_pad", "<i4"), ("cells", "<i8")])
assert OVERLAP_DTYPE.itemsize == C.sizeof(Overlap)

CANDIDATE_DTYPE = np.dtype([(n, "<i4") for n in ("ref_id", "query_id", "ref_pos", "query_pos")])


def build(force=False):
    """make -C oracle (liboracle.so always; _ref only where the reference is mounted)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or \
            os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "gact_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode("latin-1")


class Oracle:
    def __init__(self):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        L = self.lib
        L.oracle_align_with_bt.restype = C.c_int
        L.oracle_align_with_bt.argtypes = [
            C.c_char_p, C.c_longlong, C.c_char_p, C.c_longlong,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.oracle_gact.restype = None
        L.oracle_gact.argtypes = [
            C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.POINTER(Overlap), C.POINTER(TileTrace), C.c_int]
        L.oracle_gact_many.restype = C.c_int64
        L.oracle_gact_many.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_format_line.restype = C.c_int
        L.oracle_format_line.argtypes = [C.POINTER(Overlap), C.c_char_p, C.c_char_p,
                                         C.c_char_p, C.c_int]

    def align_with_bt(self, ref, query, scoring=(1, -1, -1, -1), reverse=False, first=False,
                      early_terminate=200, ref_pos=None, query_pos=None):
        ref, query = _b(ref), _b(query)
        R, Q = len(ref), len(query)
        cap = 2 * (R + Q) + 16
        out = (C.c_int * cap)()
        n = self.lib.oracle_align_with_bt(
            ref, R, query, Q, *scoring,
            Q if query_pos is None else query_pos, R if ref_pos is None else ref_pos,
            int(reverse), int(first), early_terminate, out, cap)
        if n < 0:
            raise RuntimeError("oracle_align_with_bt failed: %d" % n)
        return list(out[:n])

    def gact(self, ref, query, ref_pos, query_pos, tile_size=320, tile_overlap=120,
             threshold=35, ref_id=0, query_id=1, complement=False,
             scoring=(1, -1, -1, -1), same_file=True, trace_cap=0):
        ref, query = _b(ref), _b(query)
        ov = Overlap()
        tr = (TileTrace * max(trace_cap, 1))()
        self.lib.oracle_gact(ref, query, len(ref), len(query), tile_size, tile_overlap,
                             ref_pos, query_pos, threshold, ref_id, query_id, int(complement),
                             *scoring, int(same_file), C.byref(ov), tr, trace_cap)
        traces = [tr[k] for k in range(min(trace_cap, ov.n_tiles))]
        return ov, traces

    def gact_many(self, ref_concat, ref_offsets, query_concat, query_offsets, cands,
                  complement=False, same_file=True, tile_size=320, tile_overlap=120,
                  threshold=35, scoring=(1, -1, -1, -1), n_threads=1):
        """cands: structured array CANDIDATE_DTYPE; returns (overlaps array, cells)."""
        ref_concat = np.ascontiguousarray(ref_concat, dtype=np.uint8)
        query_concat = np.ascontiguousarray(query_concat, dtype=np.uint8)
        ref_offsets = np.ascontiguousarray(ref_offsets, dtype=np.int64)
        query_offsets = np.ascontiguousarray(query_offsets, dtype=np.int64)
        cands = np.ascontiguousarray(cands, dtype=CANDIDATE_DTYPE)
        out = np.zeros(len(cands), dtype=OVERLAP_DTYPE)
        cells = self.lib.oracle_gact_many(
            ref_concat.ctypes.data, ref_offsets.ctypes.data,
            query_concat.ctypes.data, query_offsets.ctypes.data,
            cands.ctypes.data, len(cands), int(complement), int(same_file),
            tile_size, tile_overlap, threshold, *scoring, n_threads, out.ctypes.data)
        return out, int(cells)

    def format_line(self, ov, ref_name, query_name):
        buf = C.create_string_buffer(512)
        o = ov if isinstance(ov, Overlap) else overlap_from_record(ov)
        n = self.lib.oracle_format_line(C.byref(o), _b(ref_name), _b(query_name), buf, 512)
        return buf.raw[:n].decode()


def overlap_from_record(rec):
    o = Overlap()
    for name, _ in Overlap._fields_:
        setattr(o, name, int(rec[name]))
    return o


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libdarwin_ref.so"))


class RefLib:
    """The reference's own AlignWithBT / GACT (oracle/_ref/libdarwin_ref.so)."""

    def __init__(self):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", "libdarwin_ref.so"))
        L = self.lib
        L.ref_align_with_bt.restype = C.c_int
        L.ref_align_with_bt.argtypes = [
            C.c_char_p, C.c_longlong, C.c_char_p, C.c_longlong,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.ref_gact.restype = C.c_int
        L.ref_gact.argtypes = [
            C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]

    def dsoft_candidates(self, reference_seqs, queries, seed_size=14, bin_size=64, window_size=4, threshold=21,
                         num_seeds=800, seed_occurence_multiple=32, max_candidates=1000000):
        """SeedPosTable + DSOFT of the reference on (reference_seqs, queries); returns per query the decoded
        candidates [(ref_id, ref_pos, query_pos)] exactly as darwin.cpp:213-224 derives them."""
        L = self.lib
        L.ref_dsoft_build.restype = C.c_void_p
        L.ref_dsoft_build.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ref_dsoft_query.restype = C.c_int
        L.ref_dsoft_query.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.c_uint32]
        # darwin.cpp:532-543: pad every sequence with 'N' to whole bins
        concat, start_bin, bin_to_chr, cur = b"", [], [], 0
        for i, r in enumerate(reference_seqs):
            r = _b(r)
            start_bin.append(cur)
            concat += r
            nb = len(r) // bin_size
            bin_to_chr += [i] * nb
            cur += nb
            if len(r) % bin_size:
                concat += b"N" * (bin_size - len(r) % bin_size)
                bin_to_chr.append(i)
                cur += 1
        num_bins = 1 + (len(concat) >> (bin_size.bit_length() - 1))
        tab = L.ref_dsoft_build(concat + b"\0" * 64, len(concat), seed_size, seed_occurence_multiple, bin_size,
                                window_size)
        out = (C.c_uint64 * max_candidates)()
        res = []
        for q in queries:
            q = _b(q)
            n = L.ref_dsoft_query(tab, q + b"\0" * 64, len(q), num_seeds, threshold, out, max_candidates, num_bins)
            cands = []
            for k in range(n):
                ref_pos = out[k] >> 32
                chr_id = bin_to_chr[ref_pos // bin_size] if ref_pos // bin_size < len(bin_to_chr) else 0
                ref_pos -= start_bin[chr_id] * bin_size
                ref_pos = min(ref_pos, len(reference_seqs[chr_id]))
                cands.append((chr_id, int(ref_pos), int(out[k] & 0xffffffff)))
            res.append(cands)
        return res

    def align_with_bt(self, ref, query, scoring=(1, -1, -1, -1), reverse=False, first=False,
                      early_terminate=200, ref_pos=None, query_pos=None):
        ref, query = _b(ref), _b(query)
        R, Q = len(ref), len(query)
        cap = 2 * (R + Q) + 16
        out = (C.c_int * cap)()
        n = self.lib.ref_align_with_bt(
            ref, R, query, Q, *scoring,
            Q if query_pos is None else query_pos, R if ref_pos is None else ref_pos,
            int(reverse), int(first), early_terminate, out, cap)
        if n < 0:
            raise RuntimeError("ref_align_with_bt failed: %d" % n)
        return list(out[:n])

    def gact_line(self, ref, query, ref_pos, query_pos, tile_size=320, tile_overlap=120,
                  threshold=35, ref_id=0, query_id=1, complement=False,
                  scoring=(1, -1, -1, -1), same_file=True, ref_name="r", query_name="q"):
        ref, query = _b(ref), _b(query)
        buf = C.create_string_buffer(1024)
        n = self.lib.ref_gact(ref, query, len(ref), len(query), tile_size, tile_overlap,
                              ref_pos, query_pos, threshold, ref_id, query_id, int(complement),
                              *scoring, int(same_file), _b(ref_name), _b(query_name), buf, 1024)
        if n < 0:
            raise RuntimeError("ref_gact failed")
        return buf.raw[:n].decode()

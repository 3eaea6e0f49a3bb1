"""ctypes binding of the C-ABI in include/gact_hip.h (libgact_hip.so).

Plumbing only: numpy arrays in, numpy arrays out.  Fails loudly when the HIP
library is missing or no device is present -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # darwin-gpu_amd/
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "libgact_hip.so")
SOURCES = [os.path.join(_PKG, "csrc", f) for f in
           ("gact_engine.hip", "gact_kernels.hpp", "gact_device.hpp", "gact_chain.hpp", "gact_p16.hpp", "gact_p16s.hpp", "gact_lin.hpp",
            "gact_aff.hpp", "gact_roles.hpp", "gact_coop.hpp", "gact_policy.hpp", "gact_big.hpp", "gact_gather.hpp", "dsoft_device.hpp", "dsoft_engine.hpp")] + \
          [os.path.join(_ROOT, "include", "gact_hip.h")]

SET_REF, SET_QUERY, SET_QUERY_RC = 0, 1, 2
STATE_Z, STATE_D, STATE_I, STATE_M = 0, 1, 2, 3

TILE_DTYPE = np.dtype([("ref_id", "<i4"), ("query_id", "<i4"), ("ref_off", "<i4"), ("query_off", "<i4"),
                       ("ref_len", "<i4"), ("query_len", "<i4"),
                       ("reverse", "u1"), ("first", "u1"), ("query_set", "u1"), ("pad", "u1")])
TILE_RESULT_DTYPE = np.dtype([(n, "<i4") for n in
                              ("score", "max_i", "max_j", "ref_steps", "query_steps", "n_states")])
CAND_DTYPE = np.dtype([(n, "<i4") for n in ("ref_id", "query_id", "ref_pos", "query_pos")])
OVERLAP_DTYPE = np.dtype([(n, "<i4") for n in
                          ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp", "emitted",
                           "first_tile_score", "n_tiles", "reserved")] + [("cells", "<i8")])
assert TILE_DTYPE.itemsize == 28 and OVERLAP_DTYPE.itemsize == 56


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("tile_size", "tile_overlap", "match", "mismatch", "gap_open", "gap_extend",
                 "first_tile_score_threshold", "device_id", "n_slots", "max_blocks")]


class DeviceInfo(C.Structure):
    _fields_ = [("compute_units", C.c_int32), ("clock_mhz", C.c_int32), ("waves_per_cu", C.c_int32),
                ("wave_size", C.c_int32), ("hbm_bytes", C.c_int64), ("arch", C.c_char * 32)]


class RunStats(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("seed_ms", C.c_float), ("main_ms", C.c_float),
                ("packed16", C.c_int32), ("handed_off", C.c_int32), ("seed_packed16", C.c_int32),
                ("tagged_pointers", C.c_int32), ("linear_gap", C.c_int32), ("seed_cells", C.c_int64),
                ("raw_candidates", C.c_int32), ("band_redos", C.c_int32),
                ("merged_callers", C.c_int32), ("overlapped_seeding", C.c_int32),
                ("critical_lane", C.c_int32), ("role_waves", C.c_int32)]


class DsoftParams(C.Structure):
    """params.cfg [DSOFT_params]; the defaults are the reference's"""
    _fields_ = [(n, C.c_int32) for n in ("seed_size", "bin_size", "window_size", "threshold", "num_seeds",
                                         "seed_occurence_multiple", "max_candidates")]

    def __init__(self, seed_size=14, bin_size=64, window_size=4, threshold=21, num_seeds=800,
                 seed_occurence_multiple=32, max_candidates=1000000):
        super().__init__(seed_size, bin_size, window_size, threshold, num_seeds, seed_occurence_multiple,
                         max_candidates)


class DsoftInfo(C.Structure):
    _fields_ = [("ref_length", C.c_int64), ("n_minimizers", C.c_int64), ("table_bytes", C.c_int64),
                ("pos_bytes", C.c_int64), ("max_occurrence", C.c_int32), ("n_bins", C.c_int32),
                ("build_ms", C.c_float)]


class GactHipError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> darwin-gpu_amd/libgact_hip.so (in-tree)."""
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(s) for s in SOURCES)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(_ROOT, "include"), "-o", LIB_PATH, SOURCES[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


DRIVER_PATH = os.path.join(_PKG, "host", "darwin_hip")
DRIVER_SOURCES = [os.path.join(_PKG, "host", f) for f in
                  ("darwin_hip.cpp", "gact_shim.cpp", "dsoft.cpp", "gact.h", "align.h", "dsoft.h")]


def build_driver(force=False, verbose=False):
    """g++ -> darwin-gpu_amd/host/darwin_hip: the darwin.cpp-shaped driver + the gact.h/align.h shim"""
    build(force=force, verbose=verbose)
    if not force and os.path.exists(DRIVER_PATH):
        newest = max(os.path.getmtime(s) for s in DRIVER_SOURCES + [LIB_PATH])
        if os.path.getmtime(DRIVER_PATH) >= newest:
            return DRIVER_PATH
    cmd = ["g++", "-O2", "-std=c++14", "-pthread", "-I" + os.path.join(_ROOT, "include"),
           "-I" + os.path.join(_PKG, "host"), "-o", DRIVER_PATH, DRIVER_SOURCES[0], DRIVER_SOURCES[1], DRIVER_SOURCES[2],
           "-L" + _PKG, "-lgact_hip", "-Wl,-rpath," + _PKG]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return DRIVER_PATH


ASAN_DRIVER_PATH = DRIVER_PATH + "_asan"


def build_driver_asan(force=False, verbose=False):
    """the same host sources under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on
    the CPU build only; the CPU suite drives it through --dsoft-only, which never touches the GPU)"""
    build(verbose=verbose)
    if not force and os.path.exists(ASAN_DRIVER_PATH):
        newest = max(os.path.getmtime(s) for s in DRIVER_SOURCES + [LIB_PATH])
        if os.path.getmtime(ASAN_DRIVER_PATH) >= newest:
            return ASAN_DRIVER_PATH
    cmd = ["g++", "-O1", "-g", "-std=c++14", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_PKG, "host"),
           "-o", ASAN_DRIVER_PATH, DRIVER_SOURCES[0], DRIVER_SOURCES[1], DRIVER_SOURCES[2],
           "-L" + _PKG, "-lgact_hip", "-Wl,-rpath," + _PKG]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return ASAN_DRIVER_PATH


def driver_path():
    """the built driver; compiled here only if it does not exist yet (never re-built behind the back of
    concurrently running ranks -- __graft_entry__.build() is what refreshes stale binaries)"""
    if os.path.exists(DRIVER_PATH) and os.path.exists(LIB_PATH):
        return DRIVER_PATH
    return build_driver()


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    LIB_PATH = os.environ.get("GACT_HIP_LIB_PATH", LIB_PATH)        # A/B measurements of two builds in one GPU call
    if not os.path.exists(LIB_PATH):
        raise GactHipError("%s is missing: run __graft_entry__.build() (hipcc) first; "
                           "this engine has no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    L.gact_hip_last_error.restype = C.c_char_p
    L.gact_hip_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.gact_hip_destroy.argtypes = [vp]
    L.gact_hip_destroy.restype = None
    L.gact_hip_get_device_info.argtypes = [vp, C.POINTER(DeviceInfo)]
    L.gact_hip_upload_seqs.argtypes = [vp, C.c_int, vp, vp, i32]
    L.gact_hip_align_tiles.argtypes = [vp, C.c_int, i32, vp, vp, vp, i32]
    L.gact_hip_align_tiles_inline.argtypes = [vp, C.c_int, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32]
    L.gact_hip_extend_candidates.argtypes = [vp, C.c_int, i32, vp, C.c_int, C.c_int, vp]
    L.gact_hip_candidates_upload.argtypes = [vp, C.c_int, i32, vp]
    L.gact_hip_candidates_run.argtypes = [vp, C.c_int, i32, C.c_int, C.c_int]
    L.gact_hip_candidates_run_range.argtypes = [vp, C.c_int, i32, i32, C.c_int, C.c_int]
    L.gact_hip_candidates_run_mixed.argtypes = [vp, C.c_int, i32, i32, i32, C.c_int]
    L.gact_hip_candidates_fetch.argtypes = [vp, C.c_int, i32, vp]
    L.gact_hip_sync.argtypes = [vp, C.c_int]
    L.gact_hip_last_kernel_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.gact_hip_last_run_stats.argtypes = [vp, C.c_int, C.POINTER(RunStats)]
    L.gact_hip_device_overlaps.argtypes = [vp, C.c_int]
    L.gact_hip_device_overlaps.restype = vp
    L.gact_hip_stream.argtypes = [vp, C.c_int]
    L.gact_hip_stream.restype = vp
    L.gact_hip_measure_valu_rate.argtypes = [vp, C.POINTER(C.c_double)]
    L.gact_hip_measure_valu_rate.restype = C.c_int
    L.gact_hip_format_overlap.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, i32]
    L.gact_hip_dsoft_build.argtypes = [vp, C.POINTER(DsoftParams), C.POINTER(DsoftInfo)]
    L.gact_hip_dsoft_query.argtypes = [vp, C.c_int, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_float)]
    L.gact_hip_candidates_download.argtypes = [vp, C.c_int, i32, vp]
    L.gact_hip_derive_revcomp.argtypes = [vp]
    L.gact_hip_register_output.argtypes = [vp, C.c_int, vp, C.c_int64]
    L.gact_hip_unregister_output.argtypes = [vp, C.c_int]
    try:                                       # (an older build loaded through GACT_HIP_LIB_PATH for an A/B run has none)
        L.gact_hip_comm_create.argtypes = [vp, i32, i32, C.c_char_p, i32, C.POINTER(vp)]
        L.gact_hip_comm_gather_lines.argtypes = [vp, C.c_int, i32, vp, vp, C.c_int64]
        L.gact_hip_comm_destroy.argtypes = [vp]
        for name in ("comm_create", "comm_gather_lines", "comm_destroy"):
            getattr(L, "gact_hip_" + name).restype = C.c_int
    except AttributeError:
        pass
    try:
        L.gact_hip_set_option.argtypes = [vp, C.c_char_p, i32]
        L.gact_hip_set_option.restype = C.c_int
    except AttributeError:
        pass
    for name in ("register_output", "unregister_output", "derive_revcomp", "dsoft_build", "dsoft_query", "candidates_download", "create", "get_device_info", "upload_seqs", "align_tiles", "align_tiles_inline",
                 "extend_candidates", "candidates_upload", "candidates_run", "candidates_run_range", "candidates_run_mixed",
                 "candidates_fetch", "sync", "last_kernel_ms", "last_run_stats", "format_overlap"):
        getattr(L, "gact_hip_" + name).restype = C.c_int
    _lib = L
    return L


EXPORTS = ("gact_hip_create", "gact_hip_destroy", "gact_hip_last_error", "gact_hip_get_device_info",
           "gact_hip_upload_seqs", "gact_hip_align_tiles", "gact_hip_align_tiles_inline",
           "gact_hip_extend_candidates", "gact_hip_candidates_upload", "gact_hip_candidates_run",
           "gact_hip_candidates_run_range", "gact_hip_candidates_run_mixed", "gact_hip_candidates_fetch", "gact_hip_sync",
           "gact_hip_last_kernel_ms", "gact_hip_last_run_stats", "gact_hip_device_overlaps", "gact_hip_stream",
           "gact_hip_measure_valu_rate", "gact_hip_format_overlap", "gact_hip_dsoft_build", "gact_hip_dsoft_query",
           "gact_hip_candidates_download", "gact_hip_derive_revcomp", "gact_hip_register_output",
           "gact_hip_unregister_output", "gact_hip_set_option", "gact_hip_prepare",
           "gact_hip_comm_create", "gact_hip_comm_gather_lines", "gact_hip_comm_destroy", "gact_hip_options_describe", "gact_hip_plan_describe")


def plan(count, flags=0, compute_units=256, tile_size=320, tile_overlap=120, scoring=(1, -1, -1, -1), threshold=35):
    """the launch plan of gact_policy.hpp for one pass over `count` candidates (no device needed); flags: 1 raw bytes,
    2 the launch shares the machine, 4 role launch on, 8 cooperative launch always, 16 never"""
    import json
    lib = load()
    lib.gact_hip_plan_describe.restype = C.c_int64
    lib.gact_hip_plan_describe.argtypes = [C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_int64]
    p = Params(tile_size, tile_overlap, scoring[0], scoring[1], scoring[2], scoring[3], threshold, 0, 1, 0)
    buf = C.create_string_buffer(1024)
    if lib.gact_hip_plan_describe(C.byref(p), compute_units, count, flags, buf, 1024) < 0:
        raise GactHipError(lib.gact_hip_last_error().decode())
    return json.loads(buf.value.decode())


def options_table():
    """[(name, environment variable or None, when, class, doc)] of every switch libgact_hip.so reads (no device needed)"""
    lib = load()
    lib.gact_hip_options_describe.restype = C.c_int64
    lib.gact_hip_options_describe.argtypes = [C.c_char_p, C.c_int64]
    n = lib.gact_hip_options_describe(None, 0)
    buf = C.create_string_buffer(int(n))
    lib.gact_hip_options_describe(buf, n)
    rows = []
    for line in buf.value.decode().splitlines():
        name, env, when, klass, doc = [x.strip() for x in line.split(" | ", 4)]
        rows.append((name, None if env == "-" else env, when, klass, doc))
    return rows


class Engine:
    """One gact_hip_engine.  Mirrors GPU_init .. GPU_close of the reference."""

    def __init__(self, tile_size=320, tile_overlap=120, scoring=(1, -1, -1, -1), threshold=35,
                 device_id=0, n_slots=1, max_blocks=0):
        self.L = load()
        self.p = Params(tile_size, tile_overlap, scoring[0], scoring[1], scoring[2], scoring[3],
                        threshold, device_id, n_slots, max_blocks)
        self.h = C.c_void_p()
        self._registered = {}
        self._check(self.L.gact_hip_create(C.byref(self.p), C.byref(self.h)))
        self.tile_size = tile_size
        self.tile_overlap = tile_overlap

    def _check(self, rc):
        if rc < 0:
            raise GactHipError("gact_hip error %d: %s" % (rc, self.L.gact_hip_last_error().decode()))
        return rc

    def close(self):
        if self.h:
            for slot in list(getattr(self, "_registered", {})):
                self.L.gact_hip_unregister_output(self.h, slot)
            self._registered = {}
            self.L.gact_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        info = DeviceInfo()
        self._check(self.L.gact_hip_get_device_info(self.h, C.byref(info)))
        return {"compute_units": info.compute_units, "clock_mhz": info.clock_mhz,
                "waves_per_cu": info.waves_per_cu, "wave_size": info.wave_size,
                "hbm_bytes": info.hbm_bytes, "arch": info.arch.decode()}

    def upload(self, which, concat, offsets):
        concat = np.ascontiguousarray(concat, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self._check(self.L.gact_hip_upload_seqs(self.h, which, concat.ctypes.data, offsets.ctypes.data,
                                                len(offsets) - 1))

    def derive_revcomp(self):
        """SET_QUERY_RC := reverse complement of SET_QUERY, on the device (darwin.cpp:110-147)"""
        self._check(self.L.gact_hip_derive_revcomp(self.h))

    def upload_seqs(self, which, seqs):
        seqs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else
                np.asarray(s, dtype=np.uint8) for s in seqs]
        offs = np.zeros(len(seqs) + 1, dtype=np.int64)
        if seqs:
            offs[1:] = np.cumsum([len(s) for s in seqs])
        cat = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
        self.upload(which, cat, offs)

    def align_tiles(self, tiles, slot=0):
        tiles = np.ascontiguousarray(tiles, dtype=TILE_DTYPE)
        n = len(tiles)
        stride = 2 * self.tile_size
        res = np.zeros(n, dtype=TILE_RESULT_DTYPE)
        states = np.zeros((n, stride), dtype=np.uint8)
        self._check(self.L.gact_hip_align_tiles(self.h, slot, n, tiles.ctypes.data, res.ctypes.data,
                                                states.ctypes.data, stride))
        return res, states

    def align_tiles_inline(self, refs, queries, reverses, firsts, slot=0):
        n = len(refs)
        stride_seq = max([1] + [len(r) for r in refs] + [len(q) for q in queries])
        rb = np.zeros((n, stride_seq), dtype=np.uint8)
        qb = np.zeros((n, stride_seq), dtype=np.uint8)
        rl = np.zeros(n, dtype=np.int32)
        ql = np.zeros(n, dtype=np.int32)
        for t in range(n):
            r = np.frombuffer(refs[t], dtype=np.uint8) if isinstance(refs[t], (bytes, bytearray)) else refs[t]
            q = np.frombuffer(queries[t], dtype=np.uint8) if isinstance(queries[t], (bytes, bytearray)) else queries[t]
            rb[t, :len(r)] = r
            qb[t, :len(q)] = q
            rl[t], ql[t] = len(r), len(q)
        rv = np.ascontiguousarray(reverses, dtype=np.uint8)
        fs = np.ascontiguousarray(firsts, dtype=np.uint8)
        stride = 2 * self.tile_size
        res = np.zeros(n, dtype=TILE_RESULT_DTYPE)
        states = np.zeros((n, stride), dtype=np.uint8)
        self._check(self.L.gact_hip_align_tiles_inline(
            self.h, slot, n, rb.ctypes.data, qb.ctypes.data, stride_seq, rl.ctypes.data, ql.ctypes.data,
            rv.ctypes.data, fs.ctypes.data, res.ctypes.data, states.ctypes.data, stride))
        return res, states

    def extend(self, cands, complement=False, same_file=True, slot=0):
        cands = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        out = np.zeros(len(cands), dtype=OVERLAP_DTYPE)
        self._check(self.L.gact_hip_extend_candidates(self.h, slot, len(cands), cands.ctypes.data,
                                                      int(complement), int(same_file), out.ctypes.data))
        return out

    def candidates_upload(self, cands, slot=0):
        cands = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        self._check(self.L.gact_hip_candidates_upload(self.h, slot, len(cands), cands.ctypes.data))

    def candidates_run(self, n, complement=False, same_file=True, slot=0, first=0):
        self._check(self.L.gact_hip_candidates_run_range(self.h, slot, first, n, int(complement), int(same_file)))

    def candidates_run_mixed(self, n, rc_from, same_file=True, slot=0, first=0):
        """candidates [first, first+n) of the uploaded array; index >= rc_from => complement"""
        self._check(self.L.gact_hip_candidates_run_mixed(self.h, slot, first, n, rc_from, int(same_file)))

    def candidates_fetch(self, n, slot=0, out=None):
        """records of candidates [0, n); `out`: a caller-owned OVERLAP_DTYPE array to fill (no allocation)"""
        if out is None:
            out = np.empty(n, dtype=OVERLAP_DTYPE)
        assert out.dtype == OVERLAP_DTYPE and len(out) >= n and out.flags["C_CONTIGUOUS"]
        self._check(self.L.gact_hip_candidates_fetch(self.h, slot, n, out.ctypes.data))
        return out

    def register_output(self, out, slot=0):
        """page-locks a caller-owned record array that will be fetched into repeatedly (opt-in; it must outlive the
        registration)"""
        assert out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]
        self._check(self.L.gact_hip_register_output(self.h, slot, out.ctypes.data, out.nbytes))
        self._registered[slot] = out               # (the array must not be collected while it is page-locked)

    def unregister_output(self, slot=0):
        self._check(self.L.gact_hip_unregister_output(self.h, slot))
        self._registered.pop(slot, None)

    def sync(self, slot=0):
        self._check(self.L.gact_hip_sync(self.h, slot))

    def prepare(self, expected_candidates=0):
        """arrays, second streams and empty launches ahead of the first job (include/gact_hip.h gact_hip_prepare)"""
        if hasattr(self.L, "gact_hip_prepare"):
            self.L.gact_hip_prepare.argtypes = [C.c_void_p, C.c_int32]
            self.L.gact_hip_prepare.restype = C.c_int
            self._check(self.L.gact_hip_prepare(self.h, int(expected_candidates)))

    def set_option(self, name, value):
        """scheduling switches of a live engine: "overlap_seed", "combine", "combine_window_us" (include/gact_hip.h)"""
        if hasattr(self.L, "gact_hip_set_option"):
            self._check(self.L.gact_hip_set_option(self.h, name.encode(), int(value)))

    # ---- D-SOFT on the device
    def dsoft_build(self, params=None):
        """minimizer index over the uploaded SET_REF (seed_pos_table.cpp:46-98)"""
        params = params or DsoftParams()
        info = DsoftInfo()
        self._check(self.L.gact_hip_dsoft_build(self.h, C.byref(params), C.byref(info)))
        return {n: getattr(info, n) for n, _ in DsoftInfo._fields_}

    def dsoft_query(self, first_query, n_queries, slot=0):
        """filters queries [first, first+n) of SET_QUERY / SET_QUERY_RC; the slot's device candidate array then
        holds the forward candidates followed by the reverse-complement ones.  Returns (n_forward, n_reverse, ms)."""
        nf, nr, ms = C.c_int32(), C.c_int32(), C.c_float()
        self._check(self.L.gact_hip_dsoft_query(self.h, slot, first_query, n_queries, C.byref(nf), C.byref(nr),
                                                C.byref(ms)))
        return nf.value, nr.value, float(ms.value)

    def candidates_download(self, n, slot=0):
        out = np.zeros(n, dtype=CAND_DTYPE)
        self._check(self.L.gact_hip_candidates_download(self.h, slot, n, out.ctypes.data))
        return out

    def last_kernel_ms(self, slot=0):
        ms = C.c_float()
        self._check(self.L.gact_hip_last_kernel_ms(self.h, slot, C.byref(ms)))
        return float(ms.value)

    def last_run_stats(self, slot=0):
        st = RunStats()
        self._check(self.L.gact_hip_last_run_stats(self.h, slot, C.byref(st)))
        if st.band_redos & (1 << 30):
            raise GactHipError("the role launch's watchdog fired: a DP wave went on without its walker's results (records are wrong)")
        return {"total_ms": st.total_ms, "seed_ms": st.seed_ms, "main_ms": st.main_ms,
                "packed16": bool(st.packed16), "layout": ("int32", "packed16-uniform", "packed16-split", "packed16-wide")[st.packed16],
                "seed_layout": "packed16" if st.seed_packed16 else "int32",
                "tagged_pointers": bool(st.tagged_pointers), "linear_gap": st.linear_gap == 1, "affine_drift": st.linear_gap == 2,
                "handed_off": st.handed_off, "seed_cells": st.seed_cells, "raw_candidates": st.raw_candidates,
                "band_redos": st.band_redos, "merged_callers": st.merged_callers, "overlapped_seeding": bool(st.overlapped_seeding),
                "critical_lane": bool(st.critical_lane), "role_waves": st.role_waves == 1, "coop_walks": st.role_waves == 2}

    def measure_valu_rate(self):
        v = C.c_double()
        self._check(self.L.gact_hip_measure_valu_rate(self.h, C.byref(v)))
        return float(v.value)

    def device_overlaps_ptr(self, slot=0):
        return self.L.gact_hip_device_overlaps(self.h, slot)

    def format_overlap(self, rec, ref_name, query_name):
        rec = np.ascontiguousarray(rec, dtype=OVERLAP_DTYPE)
        buf = C.create_string_buffer(512)
        n = self._check(self.L.gact_hip_format_overlap(rec.ctypes.data, ref_name.encode(), query_name.encode(),
                                                       buf, 512))
        return buf.raw[:n].decode()


class Comm:
    """The C-ABI's own RCCL gather (include/gact_hip.h gact_hip_comm_*; csrc/gact_gather.hpp): what host/darwin_hip
    --rccl-gather and any C caller use where bench.py uses torch.distributed.  Collective calls: every rank makes them."""

    LINE_DTYPE = np.dtype([(n, "<i4") for n in ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp_emitted")])

    def __init__(self, eng, rank, world, id_path, timeout_s=120):
        self.eng, self.rank, self.world = eng, rank, world
        self.h = C.c_void_p()
        eng._check(eng.L.gact_hip_comm_create(eng.h, rank, world, id_path.encode(), int(timeout_s), C.byref(self.h)))

    def gather_lines(self, n, slot=0, room=None):
        """the first n records of `slot`, as 32-byte lines, to rank 0: (counts per rank, lines on rank 0 / None elsewhere)"""
        counts = np.zeros(self.world, dtype=np.int64)
        cap = int(room if room is not None else (n + 1) * self.world) if self.rank == 0 else 0
        lines = np.zeros(cap, dtype=self.LINE_DTYPE) if self.rank == 0 else None
        self.eng._check(self.eng.L.gact_hip_comm_gather_lines(self.h, slot, int(n), counts.ctypes.data,
                                                              lines.ctypes.data if lines is not None else None, cap))
        return counts, (lines[:int(counts.sum())] if lines is not None else None)

    def close(self):
        if self.h:
            self.eng.L.gact_hip_comm_destroy(self.h)
            self.h = C.c_void_p()


def queue_from_tile(result, states, first):
    """The std::queue<int> AlignWithBT would return (align.cpp:190-199), as a list."""
    n = int(result["n_states"])
    st = [int(x) for x in states[:n]]
    if first:
        return [int(result["score"]), int(result["max_i"]), int(result["max_j"])] + st
    return [int(result["score"])] + st

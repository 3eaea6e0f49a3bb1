"""CPU suite: the C-ABI library loads and exports every symbol include/gact_hip.h declares;
struct layouts seen from Python match the header; no compute call is made (no GPU here)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "gact_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gact_hip_[a-z_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(hip_lib_path):
    from gact_amd import engine
    lib = ctypes.CDLL(hip_lib_path)
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(engine.EXPORTS)


def test_struct_layouts(tmp_path):
    """sizes and a few offsets as a C compiler sees include/gact_hip.h == what the ctypes / numpy side declares"""
    import subprocess
    from gact_amd import engine
    assert engine.TILE_DTYPE.itemsize == 28
    assert engine.TILE_RESULT_DTYPE.itemsize == 24
    assert engine.CAND_DTYPE.itemsize == 16
    assert engine.OVERLAP_DTYPE.itemsize == 56 and engine.OVERLAP_DTYPE.fields["cells"][1] == 48
    assert ctypes.sizeof(engine.Params) == 40
    (tmp_path / "abi.c").write_text(
        "#include <stdio.h>\n#include <stddef.h>\n#include \"gact_hip.h\"\n"
        "int main(void) { printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(gact_hip_params), "
        "sizeof(gact_tile), sizeof(gact_tile_result), sizeof(gact_candidate), sizeof(gact_overlap), "
        "offsetof(gact_overlap, cells), sizeof(gact_hip_run_stats), offsetof(gact_hip_run_stats, seed_cells), "
        "sizeof(gact_hip_device_info), sizeof(gact_dsoft_params), sizeof(gact_dsoft_info), "
        "offsetof(gact_dsoft_info, build_ms)); return 0; }\n")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), "-o", str(tmp_path / "abi"),
                           str(tmp_path / "abi.c")])
    got = [int(v) for v in subprocess.check_output([str(tmp_path / "abi")]).split()]
    want = [ctypes.sizeof(engine.Params), engine.TILE_DTYPE.itemsize, engine.TILE_RESULT_DTYPE.itemsize,
            engine.CAND_DTYPE.itemsize, engine.OVERLAP_DTYPE.itemsize, engine.OVERLAP_DTYPE.fields["cells"][1],
            ctypes.sizeof(engine.RunStats), engine.RunStats.seed_cells.offset, ctypes.sizeof(engine.DeviceInfo),
            ctypes.sizeof(engine.DsoftParams), ctypes.sizeof(engine.DsoftInfo), engine.DsoftInfo.build_ms.offset]
    assert got == want


def test_create_fails_loudly_without_a_device(hip_lib_path):
    """no CPU fallback: without a GPU gact_hip_create must fail with a message"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    from gact_amd import engine
    try:
        engine.Engine()
    except engine.GactHipError as e:
        assert "error" in str(e)
    else:
        raise AssertionError("Engine() succeeded without a device")


def test_format_overlap_matches_reference_line(hip_lib_path, oracle):
    from gact_amd import engine
    lib = engine.load()
    rec = np.zeros(1, dtype=engine.OVERLAP_DTYPE)
    rec[0] = (3, 4, 0, 3412, 3671, 7056, 2744, 1, 1, 50, 40, 0, 123)
    buf = ctypes.create_string_buffer(512)
    n = lib.gact_hip_format_overlap(rec.ctypes.data, b"S19_12255_5176", b"S24_8690_6836", buf, 512)
    line = buf.raw[:n].decode()
    assert line == "ref_id: S19_12255_5176, query_id: S24_8690_6836, ab: 0, ae: 3412, bb: 3671, be: 7056, score: 2744, comp: 1\n"


def test_product_never_touches_the_oracle():
    """the shipped path must not import, link or call anything under oracle/"""
    pkg = os.path.join(ROOT, "darwin-gpu_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in txt and "gact_oracle" not in txt and "liboracle" not in txt, f


def test_gather_entry_points_refuse_bad_arguments_and_the_line_record_is_the_python_sides(hip_lib_path, tmp_path):
    """gact_hip_comm_* (the C++ side's RCCL gather): argument checks come before anything touches a device or RCCL; the
    32-byte gact_line of the header is gact_amd/dist.py's LINE_DTYPE field for field"""
    import subprocess
    from gact_amd import dist
    lib = ctypes.CDLL(hip_lib_path)
    lib.gact_hip_last_error.restype = ctypes.c_char_p
    out = ctypes.c_void_p()
    assert lib.gact_hip_comm_create(None, 0, 1, b"id", 0, ctypes.byref(out)) != 0
    assert b"NULL" in lib.gact_hip_last_error()
    assert lib.gact_hip_comm_gather_lines(None, 0, 0, None, None, 0) != 0
    assert lib.gact_hip_comm_destroy(None) == 0
    names = [n for n in dist.LINE_DTYPE.names]
    (tmp_path / "line.c").write_text(
        "#include <stdio.h>\n#include <stddef.h>\n#include \"gact_hip.h\"\nint main(void) { printf(\"%zu\", sizeof(gact_line)); "
        + " ".join('printf(" %%zu", offsetof(gact_line, %s));' % n for n in names) + " return 0; }\n")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), "-o", str(tmp_path / "line"), str(tmp_path / "line.c")])
    got = [int(v) for v in subprocess.check_output([str(tmp_path / "line")]).split()]
    assert got == [dist.LINE_BYTES] + [dist.LINE_DTYPE.fields[n][1] for n in names]


def test_every_switch_of_the_library_is_in_the_one_table_and_in_the_docs():
    """the library's switches live in ONE table (csrc/gact_engine.hip kOptions, gact_hip_options_describe);
    INTEGRATION.md 7 prints it; nothing under csrc/ reads a GACT_HIP_* variable outside it."""
    import re
    from gact_amd import engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = engine.options_table()
    assert len(rows) >= 30 and len({r[0] for r in rows}) == len(rows)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    for name, env, when, klass, text in rows:
        assert "`%s`" % name in doc, "INTEGRATION.md 7 lacks the switch %s" % name
        if env:
            assert "`%s`" % env in doc, "INTEGRATION.md 7 lacks %s" % env
    envs = {r[1] for r in rows if r[1]}
    src_dir = os.path.join(root, "darwin-gpu_amd", "csrc")
    for fn in sorted(os.listdir(src_dir)):
        text = open(os.path.join(src_dir, fn)).read()
        for m in re.finditer(r'getenv\("(GACT_HIP_[A-Z0-9_]+)"\)', text):
            raise AssertionError("%s reads %s with getenv: switches go through the option table (opt_env)" % (fn, m.group(1)))
        for m in re.finditer(r'"(GACT_HIP_[A-Z0-9_]+)"', text):
            assert m.group(1) in envs, "%s names %s, which is not in the option table" % (fn, m.group(1))
    # diagnostic switches are dead in the default build: the table says so, and the default build is what is tested here
    assert any(k.startswith("diagnostic") for _, _, _, k, _ in rows)

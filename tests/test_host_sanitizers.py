"""CPU suite: the host side of the product (driver, gact.h/align.h shim, D-SOFT restatement) built with
-fsanitize=address,undefined and driven through the stages that need no GPU: FASTA and
params.cfg parsing, reverse complements, index build, the multi-threaded filter, the candidate dump.
GPU AddressSanitizer is not available on the pool, so the device side is covered by its parity tests only."""
import json
import os
import struct
import subprocess

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_dsoft_stage_under_asan_ubsan(tmp_path):
    from gact_amd import engine
    from test_dsoft import CFG
    drv = engine.build_driver_asan()
    g = json.load(open(os.path.join(GOLD, "dsoft.json")))
    # CRLF line ends, a header with fields after the name, blank lines: what parse_fasta / split_header must survive
    fasta = g["fasta"].replace("\n", "\r\n", 40).replace(">S2_", ">S2_", 1)
    (tmp_path / "reads.fasta").write_text(fasta)
    (tmp_path / "params.cfg").write_text("; comment\n# comment\n\n" + CFG % g["seed_size"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1")
    for threads in (1, 5):
        out = subprocess.run([drv, "reads.fasta", "reads.fasta", str(threads), "--dsoft-only", "--dump-candidates", "c.bin"],
                             capture_output=True, text=True, cwd=tmp_path, timeout=600, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr
        raw = open(tmp_path / "c.bin", "rb").read()
        got = [list(struct.unpack_from("<5i", raw, k)) for k in range(0, len(raw), 20)]
        assert got == g["candidates"]
        for label in ("Time finding seeds", "Time elapsed (seed position table construction)"):   # darwin.cpp:300,596
            assert label in out.stdout
    # bad options and unreadable inputs end in a message, not in a crash
    out = subprocess.run([drv, "reads.fasta", "reads.fasta", "1", "--shard", "3/2"], capture_output=True, text=True,
                         cwd=tmp_path, env=env)
    assert out.returncode == 1 and "AddressSanitizer" not in out.stderr
    out = subprocess.run([drv, "missing.fasta", "missing.fasta", "1", "--dsoft-only"], capture_output=True, text=True,
                         cwd=tmp_path, env=env)
    assert out.returncode == 1 and "AddressSanitizer" not in out.stderr

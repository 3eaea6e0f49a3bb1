"""The reference's own caller on the engine: oracle/_ref/darwin_on_hip is the reference's UNMODIFIED darwin.cpp
(with its filter, FASTA and config sources) compiled against host/gact.h + host/align.h + gact_shim.cpp and linked
with libgact_hip.so -- INTEGRATION.md section 3's link line, built by oracle/Makefile from the sources where they
lie under /root/reference (a throw-away directory of symbolic links; nothing is copied).  Test-only binaries,
git-ignored, they travel to the GPU box prebuilt like oracle/_ref/libdarwin_ref.so.

  CPU:  the two binaries link (with -DGPU: GACT_Batch path darwin.cpp:429-433; without: GACT per candidate
        darwin.cpp:240-246), depend on libgact_hip.so and carry none of the reference's GACT code
  GPU:  they run on tests/golden/dsoft.json's FASTA and print exactly the lines the reference's own CPU program
        printed (tests/golden/e2e.json), with 1, 2 and 4 feeder threads (GPU_init darwin.cpp:611, GPU_close :642)

Thread counts that divide the 16 reads only: the reference's GPU path deals reads to the D-SOFT phase in ranges of
ceil(N / T) (darwin.cpp:619-621) but recodes the base strings in place in ranges of floor(N / T)
(darwin.cpp:304-312, :340-347), and reads_char[] points into those very strings (darwin.cpp:581): where T does not
divide N a thread that is done filtering recodes a read another thread has not filtered yet, which then sees
0..3 bytes instead of letters and loses candidates -- a race of the reference's own (3 threads on this FASTA:
60 of 67 lines in one run here), nothing the engine is involved in.
"""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
ON_HIP = os.path.join(REF_DIR, "darwin_on_hip")
ON_HIP_CPU = os.path.join(REF_DIR, "darwin_on_hip_cpu")
REFERENCE = "/root/reference"


def test_reference_caller_links_against_the_shim(hip_lib_path):
    if not os.path.isdir(REFERENCE):
        pytest.skip("reference not mounted: oracle/_ref/darwin_on_hip is used prebuilt")
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/darwin_on_hip", "_ref/darwin_on_hip_cpu"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    for exe in (ON_HIP, ON_HIP_CPU):
        assert os.access(exe, os.X_OK)
        ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
        assert "libgact_hip.so" in ldd and "not found" not in ldd
        syms = subprocess.run(["nm", "-C", "--defined-only", exe], capture_output=True, text=True).stdout
        # the caller's own code and the shim's entry points are in, the engine's C-ABI is imported
        for name in ("AlignReads", "GACT_Batch(", "GPU_init(", "GPU_close(", "AlignWithBT(", "SeedPosTable::DSOFT"):
            assert name in syms, name
        und = subprocess.run(["nm", "-C", "--undefined-only", exe], capture_output=True, text=True).stdout
        assert "gact_hip_create" in und and "gact_hip_extend_candidates" in und


@pytest.mark.gpu
@pytest.mark.parametrize("exe,threads", [(ON_HIP, 1), (ON_HIP, 2), (ON_HIP, 4), (ON_HIP_CPU, 2), (ON_HIP_CPU, 3)])
def test_reference_darwin_cpp_runs_on_the_engine(tmp_path, exe, threads):
    from test_dsoft import CFG
    if not os.path.exists(exe):
        pytest.skip("%s not built (needs /root/reference at build time)" % os.path.relpath(exe, ROOT))
    gold = os.path.join(ROOT, "tests", "golden")
    fasta = json.load(open(os.path.join(gold, "dsoft.json")))["fasta"]
    e2e = json.load(open(os.path.join(gold, "e2e.json")))
    (tmp_path / "reads.fasta").write_text(fasta)
    (tmp_path / "params.cfg").write_text(CFG % e2e["seed_size"])
    # README:11-15: ./darwin <REF> <READS> CPU_THREADS [NUM_BLOCKS THREADS_PER_BLOCK]
    args = [exe, "reads.fasta", "reads.fasta", str(threads)] + (["32", "64"] if exe == ON_HIP else [])
    out = subprocess.run(args, capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = []
    for name in sorted(os.listdir(tmp_path)):
        if name.startswith("darwin.") and name.endswith(".out"):
            got += open(tmp_path / name).read().splitlines()
    assert sorted(got) == e2e["lines_sorted"]
    assert len(got) > 40

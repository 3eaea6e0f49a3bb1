"""The rule the linear-gap kernels' walker rests on (gact_chain.hpp, FMT 3): with gap_open == gap_extend == mismatch
the traceback of align.cpp:185-230 is the walk that follows the op code of every cell it enters.  Checked on the CPU,
model against oracle (tools/lin_walk_model.py); the kernels themselves are checked in test_gpu_chain.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_op_only_walk_equals_align_with_bt():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lin_walk_model.py"), "500", "11"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "500 random tiles identical" in out.stdout

"""CPU suite: host-side pieces of the engine headers that decide which kernels run and in which order chains are
popped (compiled as a host-only program with hipcc; no device code runs)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include "gact_lin.hpp"
int main()
{
    int prev = -1, ok = 1;
    for (int t = -5; t < 4000; t++) {
        const int c = gact::length_class(t);
        if (c < prev || c < 0 || c >= gact::kBuckets) ok = 0;
        prev = c;
    }
    printf("classes %d %d %d %d %d\n", ok, gact::length_class(0), gact::length_class(15), gact::length_class(16),
           gact::length_class(100000));
    // default scoring, a scoring too large for 16 bits, one too large for the tagged form only, one for the arg-max keys only
    printf("default %d %d %d\n", gact::p16_scoring_ok(320, 1, -1, -1, -1), gact::p16_argmax_ok(320, 1),
           gact::p16_tagged_ok(320, 1, -1, -1, -1));
    printf("large %d\n", gact::p16_scoring_ok(320, 100, -90, -200, -50));
    printf("mid %d %d %d\n", gact::p16_scoring_ok(320, 30, -40, -70, -20), gact::p16_argmax_ok(320, 30),
           gact::p16_tagged_ok(320, 30, -40, -70, -20));
    printf("tile512 %d %d\n", gact::p16_scoring_ok(512, 1, -1, -1, -1), gact::p16_tagged_ok(512, 1, -1, -1, -1));
    printf("pk2 %08x %08x\n", gact::pk2(-1), gact::pk2(3));
    // linear-gap pass: open == extend == mismatch, and the drift must fit next to the scaled scores
    printf("lin %d %d %d %d %d %d\n", gact::p16_lin_ok(320, 1, -1, -1, -1), gact::p16_lin_ok(320, 1, -1, -2, -1),
           gact::p16_lin_ok(320, 2, -1, -3, -3), gact::p16_lin_ok(320, 5, -2, -2, -2), gact::p16_lin_ok(320, 3, -20, -20, -20),
           gact::p16_lin_ok(512, 1, -1, -1, -1));
    return 0;
}
'''


def test_length_classes_and_kernel_predicates(tmp_path):
    (tmp_path / "t.hip").write_text(SRC)
    exe = str(tmp_path / "t")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "darwin-gpu_amd", "csrc"), "-o", exe, str(tmp_path / "t.hip")])
    out = subprocess.check_output([exe], text=True).split("\n")
    assert out[0] == "classes 1 0 15 16 63"         # monotone, one tile wide up to 15, everything very long in the last one
    assert out[1] == "default 1 1 1"
    assert out[2] == "large 0"
    assert out[3] == "mid 1 0 0"                    # packed main kernel yes; int32 seed kernel, explicit pointer comparisons
    assert out[4] == "tile512 1 1"
    assert out[5] == "pk2 ffffffff 00030003"
    assert out[6] == "lin 1 0 0 1 0 1"

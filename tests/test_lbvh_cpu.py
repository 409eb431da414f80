"""The radix-tree index arithmetic of the GPU BVH build (csrc/rtc_lbvh.hpp) compiles for the host too;
tests/cpp/lbvh_check.cpp runs it against the tree invariants on random, clustered and duplicate-heavy key sets."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_radix_tree_index_arithmetic(tmp_path):
    exe = str(tmp_path / "lbvh_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "lbvh_check.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.startswith("OK")

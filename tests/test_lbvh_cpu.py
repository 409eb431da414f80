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


def test_four_wide_collapse(tmp_path):
    """csrc/rtc_bvh.hpp collapse_bvh4 (host code): same leaves, containing boxes, stack need within the budget
    (tests/cpp/bvh4_check.cpp)."""
    exe = str(tmp_path / "bvh4_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "bvh4_check.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.startswith("OK"), out

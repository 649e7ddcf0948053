"""The comparator networks and the lane scheme of the register wave sort (csrc/sd_wsort.h), checked on the host: the
constexpr tables the kernels are built from are compiled into a small C++ program (tests/wsort_nets_check.cpp) that applies
the 0-1 principle to the per-lane sorter and the bitonic merger of every shipped width and runs a host model of the 21
cross-lane stages on random permutations."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sort_networks_and_lane_scheme(tmp_path):
    exe = tmp_path / "wsort_nets_check"
    src = os.path.join(ROOT, "tests", "wsort_nets_check.cpp")
    inc = os.path.join(ROOT, "scikit-downscale_amd", "csrc")
    # sd_wsort.h includes <hip/hip_runtime.h> for its device half: a stub directory keeps the host compile self-contained
    stub = tmp_path / "hip"
    stub.mkdir()
    (stub / "hip_runtime.h").write_text("#pragma once\n")
    res = subprocess.run(["g++", "-O2", "-std=c++17", f"-I{tmp_path}", f"-I{inc}", src, "-o", str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.count(": ok") == 6, run.stdout

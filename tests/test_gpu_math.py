"""The kernel's own math routines (csrc/rt_math.h) against the host's libm, the library the CPU oracle computes with:
tools/math_check.hip evaluates them on the GPU over 11 M arguments (cosines near 0 and 1, denormals, shininess 0 ... 5000,
negative exponents, the special values) and fails if a result is more than one unit in the last place from the host's powf
or a special case differs.  The program is built by __graft_entry__.build() (tools/math_check.sh build)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_specular_power_is_within_one_ulp_of_the_hosts_powf():
    exe = os.path.join(ROOT, "build", "math_check")
    assert os.path.exists(exe), "build/math_check is missing: run __graft_entry__.build() (or bash tools/math_check.sh build)"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "worst 1 ulp" in out.stdout or "worst 0 ulp" in out.stdout

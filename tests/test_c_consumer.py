"""The non-Python side of the boundary, computing: examples/c_consumer.c is plain C99 on include/rrtmgp_hip.h alone (no
ctypes mirror, no C++) — what a `ccall` / cgo / FFI host links against.  Built with gcc and run on the GPU here:

  1. gray longwave, no scattering, isothermal column over a black surface: the one-angle answer is exact,
     flux_dn(surface) = sigma T^4 (1 - exp(-D tau)) with tau from the published Schneider (2004) profile, flux_up = sigma T^4
     (the property of test/angular_discretization.jl:102-153 of the reference), to 1e-12 relative;
  2. a spectral all-sky two-stream LW + SW solve with McICA clouds (cld_frac 0.6, night columns) from the raw table dump
     written by examples/make_c_consumer_case.py, compared by the C program with the fluxes and cloud cover the CPU oracle
     wrote next to the inputs: Float64, 1e-8 W/m2, cover bit-equal.
The program prints what it measured; the test asserts its verdict and re-reads the numbers."""
import os
import re
import shutil
import subprocess
import sys

import pytest

from rrtmgp_jl_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c99_consumer_solves_gray_and_spectral_cases(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler on this box")
    _lib.require_gpu()
    libdir = os.path.dirname(_lib.SO_PATH)
    exe, case = str(tmp_path / "c_consumer"), str(tmp_path / "case.bin")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_consumer.c"), "-L" + libdir, "-lhip_rrtmgp", "-lm",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "examples", "make_c_consumer_case.py"), case, "24", "40"], check=True,
                   capture_output=True, timeout=300)
    env = dict(os.environ)
    env.pop("RRTMGP_HIP_LIBRARY", None)
    r = subprocess.run([exe, case], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout)
    assert r.returncode == 0 and "0 problem(s)" in r.stdout, r.stdout + r.stderr
    gray = re.findall(r"gray isothermal.*rel ([0-9.e+-]+)\).*sigma T\^4\| ([0-9.e+-]+)", r.stdout)
    assert len(gray) == 3 and all(float(a) < 1e-12 and float(b) < 1e-12 for a, b in gray)
    m = re.search(r"LW ([0-9.e+-]+)\s+SW ([0-9.e+-]+) W/m2, cloud cover ([0-9.e+-]+)", r.stdout)
    assert m and float(m.group(1)) < 1e-8 and float(m.group(2)) < 1e-8 and float(m.group(3)) == 0.0

"""C++ host mirror (robotoc::RiccatiRecursion over the C ABI): compiles with g++ on CPU, runs on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "riccati_recursion_test.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "riccati_recursion_test.out")


def _build():
    from robotoc_amd import capi
    capi.build()
    lib_dir = os.path.join(ROOT, "robotoc_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++11", SRC, "-o", EXE, "-L" + lib_dir, "-lrtoc_hip",
                           "-Wl,-rpath," + lib_dir])


def test_cpp_host_mirror_compiles():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_host_riccati_recursion_on_gpu():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)

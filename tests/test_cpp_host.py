"""C++ host mirror (robotoc::RiccatiRecursion / UnconstrRiccatiRecursion / contact-dynamics free functions over the C ABI): compiles with g++
on CPU, runs on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = ["riccati_recursion_test", "unconstr_riccati_recursion_test", "contact_dynamics_test"]
BUILD_ONLY = TESTS + ["ocp_solver_test", "unconstr_ocp_solver_test", "ocp_solver_device_test", "ocp_solver_trot_test", "ocp_solver_jump_sto_test",
                      "ocp_solver_icub_jump_sto_test"]  # ocp_solver_test needs a stage dump: run by tests/test_cpp_solver.py


def _paths(name):
    return (os.path.join(ROOT, "tests", "cpp", name + ".cpp"), os.path.join(ROOT, "tests", "cpp", name + ".out"))


def _build(name):
    from robotoc_amd import capi
    capi.build()
    lib_dir = os.path.join(ROOT, "robotoc_amd")
    src, exe = _paths(name)
    subprocess.check_call(["g++", "-O2", "-std=c++11", src, "-o", exe, "-L" + lib_dir, "-lrtoc_hip",
                           "-Wl,-rpath," + lib_dir])
    return exe


@pytest.mark.parametrize("name", BUILD_ONLY)
def test_cpp_host_mirror_compiles(name):
    assert os.path.exists(_build(name))


@pytest.mark.gpu
@pytest.mark.parametrize("name", TESTS)
def test_cpp_host_riccati_recursion_on_gpu(name):
    exe = _build(name)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["riccati_recursion_test", "unconstr_riccati_recursion_test"])
def test_cpp_host_riccati_recursion_horizon_scan_on_gpu(name):
    """robotoc::(Unconstr)RiccatiRecursion::setHorizonScan: the same closed-form stage identities with the scan."""
    exe = _build(name)
    out = subprocess.run([exe, "scan"], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "horizon scan" in out.stdout, (out.returncode, out.stdout, out.stderr)

"""The C++ host mirror of pick_ik's solver interface (pick_ik_amd/host/pick_ik_amd.hpp): compiles
with plain g++ against the C ABI, fails loudly without a GPU, passes the reference's cases on one."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "host_cpp_check.cpp")
EXE = os.path.join(ROOT, "tests", "native", "host_cpp_check")


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__ as g
    g.build()
    lib_dir = os.path.join(ROOT, "pick_ik_amd")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(
            os.path.getmtime(SRC), os.path.getmtime(os.path.join(lib_dir, "host", "pick_ik_amd.hpp"))):
        subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-Wextra", "-Werror", SRC, "-o", EXE,
                        "-L" + lib_dir, "-lpick_ik_amd", "-Wl,-rpath," + lib_dir,
                        "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return EXE


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="no-GPU behaviour")
def test_cpp_host_fails_loudly_without_gpu(exe):
    r = subprocess.run([exe, "nogpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no HIP device" in r.stdout


@pytest.mark.gpu
def test_cpp_host_reference_cases(exe):
    r = subprocess.run([exe, "gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host C++ checks OK" in r.stdout

"""GPU: compile the C++ host mirror test (tests/cpp/test_host_shim.cpp) against the C-ABI library
and the oracle, and run it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _cmd(source):
    return ["g++", "-std=c++17", "-O2", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
            "-I" + os.path.join(ROOT, "nvbio-gpl_amd", "host"), "-I/opt/rocm/include",
            os.path.join(ROOT, "tests", "cpp", source),
            "-L" + os.path.join(ROOT, "nvbio-gpl_amd", "lib"), "-lnvbio_amd", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
            "-L/opt/rocm/lib", "-lamdhip64",
            "-Wl,-rpath," + os.path.join(ROOT, "nvbio-gpl_amd", "lib"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
            "-Wl,-rpath,/opt/rocm/lib"]


def _build(out, source="test_host_shim.cpp"):
    import oracle
    oracle.build()
    subprocess.check_call(_cmd(source) + ["-o", out])


def test_host_shim_compiles(tmp_path):
    """CPU: the shim and its test build against the header and both libraries"""
    _build(str(tmp_path / "test_host_shim"))


@pytest.mark.gpu
def test_host_shim_runs(tmp_path):
    exe = str(tmp_path / "test_host_shim")
    _build(exe)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host shim ok" in out.stdout


def test_reference_streams_binding_compiles(tmp_path):
    """CPU: reference_streams.hpp -- the binding of nvBowtie's BestScoreStream and sw-benchmark's AlignmentStream -- builds against
    doubles that declare exactly the members the reference's classes have"""
    _build(str(tmp_path / "test_reference_streams"), "test_reference_streams.cpp")


@pytest.mark.gpu
def test_reference_streams_binding_runs(tmp_path):
    exe = str(tmp_path / "test_reference_streams")
    _build(exe, "test_reference_streams.cpp")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "reference streams ok" in out.stdout

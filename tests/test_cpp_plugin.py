"""The C++ host-side mirror (bio_ik_amd/cpp/bio_ik/*.h): compiled with g++ and driven by tests/cpp/test_plugin.cpp.
CPU suite: linked against the host simulator of the kernels; GPU suite: against libbioik_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_and_run(libdir, libname, tmp_path):
    exe = str(tmp_path / "test_plugin")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "bio_ik_amd", "cpp"), os.path.join(ROOT, "tests", "cpp", "test_plugin.cpp"),
           "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-pthread", "-o", exe]
    subprocess.run(cmd, check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_cpp_plugin_on_host_simulator(hostsim_lib, tmp_path):
    build_and_run(os.path.join(ROOT, "tests", "hostsim"), "bioik_hostsim", tmp_path)


@pytest.mark.gpu
def test_cpp_plugin_on_gpu(tmp_path):
    build_and_run(os.path.join(ROOT, "bio_ik_amd"), "bioik_hip", tmp_path)

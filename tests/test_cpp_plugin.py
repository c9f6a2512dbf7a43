"""The C++ host-side mirror (bio_ik_amd/cpp/bio_ik/*.h): compiled with g++ and driven by tests/cpp/test_plugin.cpp.
CPU suite: linked against the host simulator of the kernels; GPU suite: against libbioik_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_and_run(libdir, libname, tmp_path, timeout_s=None, source="test_plugin"):
    exe = str(tmp_path / source)
    cmd = ["g++", "-std=c++17", "-O1"] + (["-DTEST_TIMEOUT=%g" % timeout_s, "-DTEST_ISLANDS=2"] if timeout_s else []) + ["-I", os.path.join(ROOT, "bio_ik_amd", "cpp"), "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", source + ".cpp"),
           "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-pthread", "-o", exe]
    subprocess.run(cmd, check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_cpp_plugin_on_host_simulator(hostsim_lib, tmp_path):
    build_and_run(os.path.join(ROOT, "tests", "hostsim"), "bioik_hostsim", tmp_path, timeout_s=600.0)


@pytest.mark.gpu
def test_cpp_plugin_on_gpu(tmp_path):
    build_and_run(os.path.join(ROOT, "bio_ik_amd"), "bioik_hip", tmp_path)


def test_goal_evaluate_on_host_against_device_costs_on_host_simulator(hostsim_lib, tmp_path):
    """Goal::evaluate of bio_ik/goal_types.h (host) == the goal costs of the kernels, all opcodes; host-only goals through GoalContext"""
    build_and_run(os.path.join(ROOT, "tests", "hostsim"), "bioik_hostsim", tmp_path, source="test_goal_eval")


@pytest.mark.gpu
def test_goal_evaluate_on_host_against_device_costs_on_gpu(tmp_path):
    build_and_run(os.path.join(ROOT, "bio_ik_amd"), "bioik_hip", tmp_path, source="test_goal_eval")


def build_and_run_plugin_tu(libdir, libname, tmp_path, timeout_s=None):
    """libbio_ik.so (bio_ik_amd/cpp/src/kinematics_plugin_hip.cpp: `BioIKKinematicsPlugin : kinematics::KinematicsBase`, PLUGINLIB_EXPORT_CLASS)
    built by the package's Makefile against the stand-in MoveIt headers and the given solver library, then driven through the base class"""
    cpp = os.path.join(ROOT, "bio_ik_amd", "cpp")
    lib = str(tmp_path / "libbio_ik.so")
    subprocess.run(["make", "-s", "-C", cpp, "SOLVER_DIR=" + libdir, "SOLVER=" + libname, "OUT=" + lib], check=True)
    exe = str(tmp_path / "test_kinematics_base")
    cmd = ["g++", "-std=c++17", "-O1"] + (["-DTEST_TIMEOUT=%g" % timeout_s, "-DTEST_ISLANDS=2"] if timeout_s else []) + [
        "-I", cpp, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(cpp, "standin"), os.path.join(ROOT, "tests", "cpp", "test_kinematics_base.cpp"),
        "-L", str(tmp_path), "-lbio_ik", "-Wl,-rpath," + str(tmp_path), "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-pthread", "-o", exe]
    subprocess.run(cmd, check=True)
    r = subprocess.run([exe, os.path.join(cpp, "bio_ik_kinematics_description.xml")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_kinematics_base_plugin_on_host_simulator(hostsim_lib, tmp_path):
    build_and_run_plugin_tu(os.path.join(ROOT, "tests", "hostsim"), "bioik_hostsim", tmp_path, timeout_s=600.0)


@pytest.mark.gpu
def test_kinematics_base_plugin_on_gpu(tmp_path):
    build_and_run_plugin_tu(os.path.join(ROOT, "bio_ik_amd"), "bioik_hip", tmp_path)

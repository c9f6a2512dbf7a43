"""Source compatibility of the goal (cost) plugin interface with user code written for the reference: the usage example of the
reference's README ("PR2 turning a valve") is cut out of /root/reference/README.md, compiled UNCHANGED against bio_ik_amd/cpp (+ the
stand-in MoveIt / tf headers of this build image) inside the harness tests/cpp/test_readme_example.cpp, and run through the plugin.
The example is not stored in this repository; where the reference tree is absent (the GPU box) the test is skipped."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
README = "/root/reference/README.md"


def extract_example(path):
    """the three fenced code blocks of the valve example: options + goals, secondary goals, the turning loop"""
    text = open(README).read()
    start = text.index("bio_ik::BioIKKinematicsQueryOptions ik_options;")
    start = text.rindex("```", 0, start)
    end = text.index("When you execute the code", start)
    blocks = re.findall(r"```\n(.*?)```", text[start:end], flags=re.S)
    assert len(blocks) == 3, len(blocks)
    code = "\n".join(blocks)
    # the one line of the example that is not C++: its placeholder for "check solution validity and actually move the robot"
    n_placeholders = len(re.findall(r"^\s*\.\.\. //.*$", code, flags=re.M))
    assert n_placeholders == 1
    code = re.sub(r"^\s*\.\.\. //.*$", "        EXAMPLE_AFTER_IK(example_ik_ok)", code, flags=re.M)
    # ... and its setFromIK statement becomes the value the hook reports (the call itself is untouched)
    assert code.count("robot_state.setFromIK(") == 1
    code = code.replace("robot_state.setFromIK(", "const bool example_ik_ok = robot_state.setFromIK(")
    open(path, "w").write(code)
    return code


def build_and_run(libdir, libname, tmp_path, max_steps, islands=None):
    cpp = os.path.join(ROOT, "bio_ik_amd", "cpp")
    lib = str(tmp_path / "libbio_ik.so")
    subprocess.run(["make", "-s", "-C", cpp, "SOLVER_DIR=" + libdir, "SOLVER=" + libname, "OUT=" + lib], check=True)
    snippet = str(tmp_path / "valve_example.inc")
    code = extract_example(snippet)
    assert "tf::Vector3" in code and "ik_options.goals.emplace_back" in code and "LookAtGoal" in code
    exe = str(tmp_path / "test_readme_example")
    cmd = ["g++", "-std=c++17", "-O1", "-DEXAMPLE_FILE=\"%s\"" % snippet, "-DEXAMPLE_MAX_STEPS=%d" % max_steps] + (["-DTEST_ISLANDS=%d" % islands] if islands else []) + [
           "-I", cpp, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(cpp, "standin"), os.path.join(ROOT, "tests", "cpp", "test_readme_example.cpp"),
           "-L", str(tmp_path), "-lbio_ik", "-Wl,-rpath," + str(tmp_path), "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-pthread", "-o", exe]
    subprocess.run(cmd, check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists(README), reason="the reference tree is not on this machine")
def test_readme_valve_example_compiles_unchanged_and_runs(hostsim_lib, tmp_path):
    build_and_run(os.path.join(ROOT, "tests", "hostsim"), "bioik_hostsim", tmp_path, max_steps=12, islands=2)

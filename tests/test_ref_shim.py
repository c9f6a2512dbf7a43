"""Known-answer tests of the stand-in tf2 / KDL headers (oracle/ref_shim) that the reference's own sources are compiled against:
expected values derived by hand from the libraries' published definitions, so that "oracle == reference" cannot rest on a helper bug the
two sides share (tests/cpp/test_ref_shim.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ref_shim_known_answers(tmp_path):
    exe = str(tmp_path / "test_ref_shim")
    subprocess.run(["g++", "-std=c++14", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle", "ref_shim"), os.path.join(ROOT, "tests", "cpp", "test_ref_shim.cpp"), "-o", exe],
                   check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr

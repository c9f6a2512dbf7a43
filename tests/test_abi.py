"""The C-ABI boundary without a GPU: libbioik_hip.so loads, exports every symbol include/bioik_hip.h declares, its PODs
match the ctypes mirror, and it refuses to work without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from bio_ik_amd import PoseGoal, ProblemTemplate, abi, solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(solver.LIB_PATH):
        import __graft_entry__ as g
        g.build_hip()
    return solver.load_library()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "bioik_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bioik_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "include/bioik_hip.h declares %s but libbioik_hip.so does not export it" % n
    assert sorted(solver.EXPORTS) == names


def test_pod_layout_and_defaults(lib):
    assert lib.bioik_abi_version() == 6
    p = abi.SolveParams()
    lib.bioik_default_solve_params(C.byref(p))
    assert p.struct_size == C.sizeof(abi.SolveParams)
    d = abi.default_solve_params()
    for f, _ in abi.SolveParams._fields_:
        assert getattr(p, f) == getattr(d, f), f
    for op, n in abi.GOAL_PARAM_COUNT.items():
        assert lib.bioik_goal_param_count(op) == n
    assert lib.bioik_goal_param_count(99) == -1


def test_no_cpu_fallback(lib, pr2):
    """on a box without a GPU the product refuses to create a model: BIOIK_ERR_NO_DEVICE"""
    if lib.bioik_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(solver.BioIKError) as e:
        solver.HipSolver(ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")]))
    assert e.value.code == abi.ERR_NO_DEVICE


def test_product_does_not_touch_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/"""
    pkg = os.path.join(ROOT, "bio_ik_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text and "orc_" not in text, f


def test_the_environment_cannot_put_the_simulator_behind_the_product_classes(hostsim_lib, monkeypatch):
    """BIOIK_HIP_LIBRARY / BIOIK_PLUGIN_SHIM select other BUILDS of the product libraries (A/B variants, a profiling build); the test-suite's host simulator is
    refused there -- it reaches HipSolver / BioIKKinematicsPlugin only as an explicit `lib=` argument of a test"""
    import pytest
    from bio_ik_amd import plugin, solver
    monkeypatch.setenv("BIOIK_HIP_LIBRARY", os.path.join(ROOT, "tests", "hostsim", "libbioik_hostsim.so"))
    monkeypatch.setattr(solver, "_lib", None)
    with pytest.raises(ImportError):
        solver.load_library()
    monkeypatch.setattr(solver, "_lib", None)
    monkeypatch.setenv("BIOIK_PLUGIN_SHIM", os.path.join(ROOT, "tests", "hostsim", "libbio_ik_shim_hostsim.so"))
    with pytest.raises(ImportError):
        plugin.load_shim()

"""ctypes wrapper of oracle/_ref/libbioik_ref.so: the REFERENCE'S OWN sources (include/bio_ik/*.h, src/forward_kinematics.h,
src/problem.cpp, src/ik_base.h, src/ik_evolution_2.cpp) compiled unmodified against the stand-in third-party headers of
oracle/ref_shim (see oracle/ref_shim/README.md, oracle/ref_driver.cpp).  TEST INFRASTRUCTURE ONLY.  It exists where
/root/reference exists (this container) and travels to the GPU box as a prebuilt file; the fixtures generated from it are
committed under tests/golden/ (tests/golden/make_reference_golden.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

from bio_ik_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "_ref", "libbioik_ref.so")
PATH_RELEASE = os.path.join(HERE, "_ref", "libbioik_ref_release.so")  # reference Release flags: timing baseline
PATH_RELEASE_V3 = os.path.join(HERE, "_ref", "libbioik_ref_release_v3.so")  # ... + -march=x86-64-v3 (AVX2, FMA): hosts that have them
_lib = None
_libs = {}
_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)


def available():
    return os.path.exists(PATH) or os.path.isdir("/root/reference/src")


def release_available():
    return os.path.exists(PATH_RELEASE) or os.path.isdir("/root/reference/src")


def v3_available():
    """the -march=x86-64-v3 build exists and this host can execute it"""
    if not os.path.exists(PATH_RELEASE_V3):
        return False
    try:
        flags = set(next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split())
    except Exception:
        return False
    return {"avx2", "fma", "bmi2", "movbe", "f16c"} <= flags and ("abm" in flags or "lzcnt" in flags)


def lib(release=False):
    global _lib
    if release == "v3":
        if "v3" not in _libs:
            _libs["v3"] = _declare(C.CDLL(PATH_RELEASE_V3))
        return _libs["v3"]
    if release:
        if "release" not in _libs:
            if not os.path.exists(PATH_RELEASE):
                subprocess.run(["make", "-C", HERE, "-s", "ref"], check=True)
            _libs["release"] = _declare(C.CDLL(PATH_RELEASE))
        return _libs["release"]
    if _lib is None:
        if not os.path.exists(PATH):
            subprocess.run(["make", "-C", HERE, "-s", "ref"], check=True)
        _lib = _declare(C.CDLL(PATH))
    return _lib


def _declare(L):
    L.ref_last_error.restype = C.c_char_p
    L.ref_create.restype = C.c_void_p
    L.ref_create.argtypes = [C.POINTER(abi.ModelDesc), C.POINTER(abi.ProblemDesc), C.POINTER(abi.SolveParams)]
    L.ref_destroy.argtypes = [C.c_void_p]
    L.ref_solver_create.restype = C.c_void_p
    L.ref_solver_create.argtypes = [C.c_void_p, _pd, _pd]
    L.ref_solver_destroy.argtypes = [C.c_void_p]
    L.ref_solver_step.argtypes = [C.c_void_p]
    L.ref_solver_result.argtypes = [C.c_void_p, _pd, _pd, _pi]
    L.ref_solve_batch.argtypes = [C.c_void_p, C.c_size_t, _pd, _pd, C.c_int, _pd, _pd, _pi, _pi]
    if hasattr(L, "ref_solve_batch_timeout"):
        L.ref_solve_batch_timeout.argtypes = [C.c_void_p, C.c_size_t, _pd, _pd, C.c_double, _pd, _pd, _pi, _pi, _pd]
    return L


def _d(a):
    return a.ctypes.data_as(_pd)


def _i(a):
    return a.ctypes.data_as(_pi)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class RefError(RuntimeError):
    pass


class Reference:
    """One (model, problem template, solver parameters) loaded into the reference's own classes."""

    def __init__(self, template, params=None, release=False):
        self.L = lib(release)
        self.template = template
        self.params = params if params is not None else abi.default_solve_params()
        md, pd = template.model.desc(), template.desc()
        self.h = self.L.ref_create(C.byref(md), C.byref(pd), C.byref(self.params))
        if not self.h:
            raise RefError(self.L.ref_last_error().decode())
        info = np.zeros(4, dtype=np.int32)
        self.L.ref_info(C.c_void_p(self.h), _i(info))
        self.D, self.T, self.P, self.V = [int(x) for x in info]
        self.active_variables = np.zeros(self.D, dtype=np.int32)
        self.L.ref_active_variables(C.c_void_p(self.h), _i(self.active_variables))
        self.tip_links = np.zeros(self.T, dtype=np.int32)
        self.L.ref_tip_links(C.c_void_p(self.h), _i(self.tip_links))

    def __del__(self):
        try:
            if self.h:
                self.L.ref_destroy(C.c_void_p(self.h))
                self.h = None
        except Exception:
            pass

    @property
    def _h(self):
        return C.c_void_p(self.h)

    def _chk(self, rc):
        if rc != 0:
            raise RefError(self.L.ref_last_error().decode())

    def _gp(self, p):
        return _f64(np.concatenate([np.asarray(p, dtype=np.float64).ravel(), [0.0]]))

    def robot_info(self):
        out = np.zeros((self.V, 6))
        self.L.ref_robot_info(self._h, _d(out))
        return out

    def canonical_params(self, params):
        """The goal numbers as the reference's goal objects STORE them when constructed from `params`: some reference
        constructors normalise their argument (PoseGoal / OrientationGoal quaternions, LineGoal / PlaneGoal directions),
        and x/|x| is not idempotent in floating point.  Bit-exact comparisons therefore feed the reference `params` and
        the oracle `canonical_params(params)` — then both work on the very same numbers."""
        out = np.zeros(self.P + 1)
        self._chk(self.L.ref_canonical_params(self._h, _d(self._gp(params)), _d(out)))
        return out[:self.P].copy()

    def fk(self, vars_):
        v = _f64(vars_).reshape(-1, self.V)
        tips = np.zeros((v.shape[0], self.T, 7))
        self._chk(self.L.ref_fk(self._h, C.c_size_t(v.shape[0]), _d(v), _d(tips)))
        return tips

    def approx_eval(self, seed, base_genes, genes):
        g = _f64(genes).reshape(-1, self.D)
        base = np.zeros((self.T, 7))
        out = np.zeros((g.shape[0], self.T, 7))
        self._chk(self.L.ref_approx_eval(self._h, _d(_f64(seed)), _d(_f64(base_genes)), C.c_size_t(g.shape[0]), _d(g), _d(base), _d(out)))
        return base, out

    def fitness(self, fk_mode, seed, goal_params, genes, base_genes=None):
        g = _f64(genes).reshape(-1, self.D)
        n = g.shape[0]
        prim, sec = np.zeros(n), np.zeros(n)
        b = _f64(base_genes) if base_genes is not None else np.zeros(self.D)
        self._chk(self.L.ref_fitness(self._h, C.c_int(fk_mode), C.c_size_t(n), _d(_f64(seed)), _d(self._gp(goal_params)), _d(b), _d(g), _d(prim), _d(sec)))
        return prim, sec

    def check(self, seed, goal_params, genes):
        g = _f64(genes).reshape(-1, self.D)
        ok = np.zeros(g.shape[0], dtype=np.int32)
        self._chk(self.L.ref_check(self._h, C.c_size_t(g.shape[0]), _d(_f64(seed)), _d(self._gp(goal_params)), _d(g), _i(ok)))
        return ok

    def solve_batch(self, seeds, goal_params, max_steps):
        """n queries through one reference solver object (budget form of src/ik_parallel.h:160-184), single thread."""
        s = _f64(seeds).reshape(-1, self.V)
        n = s.shape[0]
        gp = _f64(goal_params).reshape(n, self.P) if self.P else np.zeros((n, 1))
        sol, fit = np.zeros((n, self.V)), np.zeros(n)
        suc, steps = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        self._chk(self.L.ref_solve_batch(self._h, C.c_size_t(n), _d(s), _d(gp), C.c_int(max_steps), _d(sol), _d(fit), _i(suc), _i(steps)))
        return sol, fit, suc, steps

    def solve_batch_timeout(self, seeds, goal_params, timeout):
        """n queries, one after the other, each under the reference's wall-clock loop with `timeout` [s] (src/ik_parallel.h:160-184, one solver thread)
        -> (solutions, fitness, success, steps, seconds per query)"""
        s = _f64(seeds).reshape(-1, self.V)
        n = s.shape[0]
        gp = _f64(goal_params).reshape(n, self.P) if self.P else np.zeros((n, 1))
        sol, fit, sec = np.zeros((n, self.V)), np.zeros(n), np.zeros(n)
        suc, steps = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        self._chk(self.L.ref_solve_batch_timeout(self._h, C.c_size_t(n), _d(s), _d(gp), C.c_double(timeout), _d(sol), _d(fit), _i(suc), _i(steps), _d(sec)))
        return sol, fit, suc, steps, sec

    def solve_steps(self, seed, goal_params, n_steps):
        """IKEvolution2 through the reference's IKFactory: solution / exact fitness / success after every step()."""
        s = self.L.ref_solver_create(self._h, _d(_f64(seed)), _d(self._gp(goal_params)))
        if not s:
            raise RefError(self.L.ref_last_error().decode())
        sols, fits, sucs = [], [], []
        try:
            for _ in range(n_steps):
                self._chk(self.L.ref_solver_step(C.c_void_p(s)))
                sol = np.zeros(self.V)
                f = C.c_double()
                ok = C.c_int32()
                self._chk(self.L.ref_solver_result(C.c_void_p(s), _d(sol), C.byref(f), C.byref(ok)))
                sols.append(sol), fits.append(f.value), sucs.append(ok.value)
        finally:
            self.L.ref_solver_destroy(C.c_void_p(s))
        return np.array(sols), np.array(fits), np.array(sucs)


def _frame_fn(name, n_in, n_out):
    def f(*args):
        out = np.zeros(n_out)
        getattr(lib(), name)(*[_d(_f64(a)) for a in args], _d(out))
        return out
    return f


quat_mul_vec = _frame_fn("ref_quat_mul_vec", 2, 3)
quat_mul_quat = _frame_fn("ref_quat_mul_quat", 2, 4)
frame_concat = _frame_fn("ref_frame_concat", 2, 7)
frame_invert = _frame_fn("ref_frame_invert", 1, 7)
frame_change = _frame_fn("ref_frame_change", 3, 7)
frame_twist = _frame_fn("ref_frame_twist", 2, 6)


def normalize_fast(q):
    out = _f64(q).copy()
    lib().ref_normalize_fast(_d(out))
    return out

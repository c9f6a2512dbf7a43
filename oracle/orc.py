"""ctypes wrapper of the CPU oracle (oracle/bioik_oracle.h).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg —
never by the product package bio_ik_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

from bio_ik_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
RNG_REFERENCE, RNG_COUNTER = 0, 1
_libs = {}

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)


def build():
    subprocess.run(["make", "-C", HERE, "-s"], check=True)


def lib(kind="strict"):
    """kind: 'strict' (IEEE, parity checker) or 'ref' (reference Release flags, timing baseline)."""
    if kind not in _libs:
        path = os.path.join(HERE, "liboracle_%s.so" % kind)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_last_error.restype = C.c_char_p
        L.orc_model_create.restype = C.c_void_p
        L.orc_model_create.argtypes = [C.POINTER(abi.ModelDesc)]
        L.orc_model_destroy.argtypes = [C.c_void_p]
        L.orc_problem_create.restype = C.c_void_p
        L.orc_problem_create.argtypes = [C.c_void_p, C.POINTER(abi.ProblemDesc)]
        L.orc_problem_destroy.argtypes = [C.c_void_p]
        L.orc_solver_create.restype = C.c_void_p
        L.orc_solver_create.argtypes = [C.c_void_p, C.POINTER(abi.SolveParams), C.c_int, C.c_uint32, _pd, _pd]
        L.orc_solver_destroy.argtypes = [C.c_void_p]
        L.orc_counter_gauss32.restype = C.c_double
        L.orc_counter_uniform.restype = C.c_double
        L.orc_query_key.restype = C.c_uint32
        L.orc_query_key.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        _libs[kind] = L
    return _libs[kind]


def set_trig_mode(mode, kind="strict"):
    """0 = libm (reference behaviour, default), 1 = bioik_sincos shared bit-for-bit with the device kernels."""
    lib(kind).orc_set_trig_mode(C.c_int(mode))


def shared_sincos(x, kind="strict"):
    """bioik_sincos (the implementation shared with the device kernels) of an array"""
    x = _f64(x).ravel()
    s, c = np.zeros_like(x), np.zeros_like(x)
    lib(kind).orc_shared_sincos(C.c_size_t(x.size), _d(x), _d(s), _d(c))
    return s, c


def unbounded_candidates(kind="strict"):
    """how many line-search candidates with a gene of magnitude >= 1e300 the oracle has met so far (quirk Q7, orc_evolution.h)"""
    L = lib(kind)
    L.orc_debug_unbounded_candidates.restype = C.c_ulonglong
    return int(L.orc_debug_unbounded_candidates())


def set_quirk_mode(mode, kind="strict"):
    """0 = reference quirks Q1 / Q4 / Q5 / Q7 fixed as on the device (default); 1 = literal reference behaviour (for oracle/_ref)."""
    lib(kind).orc_set_quirk_mode(C.c_int(mode))


def _d(a):
    return a.ctypes.data_as(_pd)


def _i(a):
    return a.ctypes.data_as(_pi)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleError(RuntimeError):
    pass


class Oracle:
    """One (model, problem template) pair loaded into the oracle."""

    def __init__(self, template, kind="strict"):
        self.L = lib(kind)
        self.template = template
        md = template.model.desc()
        self.model = self.L.orc_model_create(C.byref(md))
        if not self.model:
            raise OracleError(self.L.orc_last_error().decode())
        pd = template.desc()
        self.problem = self.L.orc_problem_create(self.model, C.byref(pd))
        if not self.problem:
            msg = self.L.orc_last_error().decode()
            self.L.orc_model_destroy(self.model)
            self.model = None
            raise OracleError(msg)
        info = np.zeros(4, dtype=np.int32)
        self.L.orc_problem_info(C.c_void_p(self.problem), _i(info))
        self.D, self.T, self.P, self.V = [int(x) for x in info]
        self.active_variables = np.zeros(self.D, dtype=np.int32)
        self.L.orc_problem_active_variables(C.c_void_p(self.problem), _i(self.active_variables))
        self.tip_links = np.zeros(self.T, dtype=np.int32)
        self.L.orc_problem_tip_links(C.c_void_p(self.problem), _i(self.tip_links))

    def close(self):
        if getattr(self, "problem", None):
            self.L.orc_problem_destroy(C.c_void_p(self.problem))
            self.problem = None
        if getattr(self, "model", None):
            self.L.orc_model_destroy(C.c_void_p(self.model))
            self.model = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(self.L.orc_last_error().decode())

    @property
    def _p(self):
        return C.c_void_p(self.problem)

    def robot_info(self):
        out = np.zeros((self.V, 6))
        self.L.orc_model_robot_info(C.c_void_p(self.model), _d(out))
        return out

    def velocity_weights(self):
        out = np.zeros(self.D)
        self.L.orc_problem_velocity_weights(self._p, _d(out))
        return out

    def fk(self, vars_, want_global=False):
        v = _f64(vars_).reshape(-1, self.V)
        n = v.shape[0]
        tips = np.zeros((n, self.T, 7))
        nl = self.template.model.n_links
        glob = np.zeros((n, nl, 7)) if want_global else None
        self._chk(self.L.orc_fk(self._p, C.c_size_t(n), _d(v), _d(tips), _d(glob) if want_global else None))
        return (tips, glob) if want_global else tips

    def fk_genes(self, seed, genes):
        g = _f64(genes).reshape(-1, self.D)
        s = _f64(seed)
        tips = np.zeros((g.shape[0], self.T, 7))
        self._chk(self.L.orc_fk_genes(self._p, C.c_size_t(g.shape[0]), _d(s), _d(g), _d(tips)))
        return tips

    def jacobian(self, seed, base_genes):
        jac = np.zeros((6 * self.T, self.D))
        self._chk(self.L.orc_jacobian(self._p, _d(_f64(seed)), _d(_f64(base_genes)), _d(jac)))
        return jac

    def approximator(self, seed, base_genes):
        tips = np.zeros((self.T, 7))
        deltas = np.zeros((self.T, self.D, 7))
        mask = np.zeros((self.T, self.D), dtype=np.int32)
        self._chk(self.L.orc_approximator(self._p, _d(_f64(seed)), _d(_f64(base_genes)), _d(tips), _d(deltas), _i(mask)))
        return tips, deltas, mask

    def approx_eval(self, seed, base_genes, genes):
        g = _f64(genes).reshape(-1, self.D)
        out = np.zeros((g.shape[0], self.T, 7))
        self._chk(self.L.orc_approx_eval(self._p, _d(_f64(seed)), _d(_f64(base_genes)), C.c_size_t(g.shape[0]), _d(g), _d(out)))
        return out

    def fitness_frames(self, seed, goal_params, frames, genes):
        prim, sec = C.c_double(), C.c_double()
        gp = _f64(np.concatenate([np.asarray(goal_params, dtype=np.float64).ravel(), [0.0]]))
        self._chk(self.L.orc_fitness_frames(self._p, _d(_f64(seed)), _d(gp), _d(_f64(frames)), _d(_f64(genes)), C.byref(prim), C.byref(sec)))
        return prim.value, sec.value

    def fitness(self, fk_mode, seed, goal_params, genes, base_genes=None):
        g = _f64(genes).reshape(-1, self.D)
        n = g.shape[0]
        prim, sec = np.zeros(n), np.zeros(n)
        gp = _f64(np.concatenate([np.asarray(goal_params, dtype=np.float64).ravel(), [0.0]]))
        b = _f64(base_genes) if base_genes is not None else np.zeros(self.D)
        self._chk(self.L.orc_fitness(self._p, C.c_int(fk_mode), C.c_size_t(n), _d(_f64(seed)), _d(gp), _d(b), _d(g), _d(prim), _d(sec)))
        return prim, sec

    def check(self, params, seed, goal_params, genes):
        g = _f64(genes).reshape(-1, self.D)
        ok = np.zeros(g.shape[0], dtype=np.int32)
        gp = _f64(np.concatenate([np.asarray(goal_params, dtype=np.float64).ravel(), [0.0]]))
        self._chk(self.L.orc_check(self._p, C.byref(params), C.c_size_t(g.shape[0]), _d(_f64(seed)), _d(gp), _d(g), _i(ok)))
        return ok

    def reproduce_counter(self, population, rng_key, species, generation, parents):
        par = _f64(parents).reshape(2, 2, self.D)
        genes = np.zeros((population, self.D))
        grads = np.zeros((population, self.D))
        self._chk(self.L.orc_reproduce_counter(self._p, C.c_int(population), C.c_uint32(rng_key), C.c_int(species), C.c_uint32(generation),
                                               _d(par), _d(genes), _d(grads)))
        return genes, grads

    def solver(self, params, rng_mode, rng_key, seed, goal_params):
        return OracleSolver(self, params, rng_mode, rng_key, seed, goal_params)

    def solve_batch(self, params, rng_mode, seeds, goal_params, n_threads=1, timeout_s=0.0, first_query_index=0):
        s = _f64(seeds).reshape(-1, self.V)
        n = s.shape[0]
        gp = _f64(goal_params).reshape(n, self.P) if self.P else np.zeros((n, 1))
        sol = np.zeros((n, self.V))
        fit = np.zeros(n)
        suc = np.zeros(n, dtype=np.int32)
        steps = np.zeros(n, dtype=np.int32)
        self._chk(self.L.orc_solve_batch(self._p, C.byref(params), C.c_int(rng_mode), C.c_size_t(n), _d(s), _d(gp), _d(sol), _d(fit),
                                         _i(suc), _i(steps), C.c_int(n_threads), C.c_double(timeout_s), C.c_uint64(first_query_index)))
        return sol, fit, suc, steps

    def wrap_angles(self, seed, state):
        st = _f64(state).copy()
        self._chk(self.L.orc_wrap_angles(self._p, _d(_f64(seed)), _d(st)))
        return st


class OracleSolver:
    def __init__(self, oracle, params, rng_mode, rng_key, seed, goal_params):
        self.o = oracle
        gp = _f64(np.concatenate([np.asarray(goal_params, dtype=np.float64).ravel(), [0.0]]))
        self.h = oracle.L.orc_solver_create(oracle._p, C.byref(params), C.c_int(rng_mode), C.c_uint32(rng_key), _d(_f64(seed)), _d(gp))
        if not self.h:
            raise OracleError(oracle.L.orc_last_error().decode())

    def step(self):
        self.o._chk(self.o.L.orc_solver_step(C.c_void_p(self.h)))

    def state(self):
        D, V = self.o.D, self.o.V
        g = np.zeros((2, 2, 2, D))
        f = np.zeros(2)
        sol = np.zeros(V)
        sf = C.c_double()
        self.o.L.orc_solver_state(C.c_void_p(self.h), _d(g), _d(f), _d(sol), C.byref(sf))
        return g, f, sol, sf.value

    def check(self):
        ok = C.c_int32()
        f = C.c_double()
        self.o.L.orc_solver_check(C.c_void_p(self.h), C.byref(ok), C.byref(f))
        return bool(ok.value), f.value

    def __del__(self):
        try:
            if self.h:
                self.o.L.orc_solver_destroy(C.c_void_p(self.h))
                self.h = None
        except Exception:
            pass


# ---- free functions (L1 math, RNG) ----
def frame_concat(a, b, kind="strict"):
    out = np.zeros(7)
    lib(kind).orc_frame_concat(_d(_f64(a)), _d(_f64(b)), _d(out))
    return out


def frame_change(a, b, c, kind="strict"):
    out = np.zeros(7)
    lib(kind).orc_frame_change(_d(_f64(a)), _d(_f64(b)), _d(_f64(c)), _d(out))
    return out


def frame_invert(a, kind="strict"):
    out = np.zeros(7)
    lib(kind).orc_frame_invert(_d(_f64(a)), _d(out))
    return out


def quat_mul_vec(q, v, kind="strict"):
    out = np.zeros(3)
    lib(kind).orc_quat_mul_vec(_d(_f64(q)), _d(_f64(v)), _d(out))
    return out


def quat_mul_quat(p, q, kind="strict"):
    out = np.zeros(4)
    lib(kind).orc_quat_mul_quat(_d(_f64(p)), _d(_f64(q)), _d(out))
    return out


def normalize_fast(q, kind="strict"):
    out = _f64(q).copy()
    lib(kind).orc_normalize_fast(_d(out))
    return out


def frame_twist(a, b, kind="strict"):
    out = np.zeros(6)
    lib(kind).orc_frame_twist(_d(_f64(a)), _d(_f64(b)), _d(out))
    return out


def pose_twist(goal, tip, kind="strict"):
    out = np.zeros(6)
    lib(kind).orc_pose_twist(_d(_f64(goal)), _d(_f64(tip)), _d(out))
    return out


def linear_int_distribution_hist(seed, n, iters, kind="strict"):
    out = np.zeros(n)
    lib(kind).orc_linear_int_distribution_hist(C.c_uint32(seed), C.c_uint32(n), C.c_uint32(iters), _d(out))
    return out


def philox2x32(key, c0, c1, kind="strict"):
    out = (C.c_uint32 * 2)()
    lib(kind).orc_philox2x32(C.c_uint32(key), C.c_uint32(c0), C.c_uint32(c1), out)
    return int(out[0]), int(out[1])


def philox4x32(key2, ctr4, kind="strict"):
    k = (C.c_uint32 * 2)(*key2)
    c = (C.c_uint32 * 4)(*ctr4)
    out = (C.c_uint32 * 4)()
    lib(kind).orc_philox4x32(k, c, out)
    return [int(x) for x in out]


def counter_gauss32(word, kind="strict"):
    """the Gaussian the solver derives from one 32-bit random word"""
    return lib(kind).orc_counter_gauss32(C.c_uint32(word))


def preselect_children(seed, query, island, step, generation, species_id, population, kind="strict"):
    """how many pre-selected children a species walks in one generation (secondary goals present), drawn by the solver's own random source"""
    L = lib(kind)
    L.orc_preselect_children.restype = C.c_uint32
    return int(L.orc_preselect_children(C.c_uint64(seed), C.c_uint64(query), C.c_uint32(island), C.c_uint32(step), C.c_uint32(generation), C.c_uint32(species_id), C.c_uint32(population)))


def child_word(key, ctr1, child, w, kind="strict"):
    """random word w of child `child` in the stream (key, ctr1): word 0 is the mutation-rate exponent, word 1 + g the Gaussian of gene g"""
    L = lib(kind)
    L.orc_child_word.restype = C.c_uint32
    return int(L.orc_child_word(C.c_uint32(key), C.c_uint32(ctr1), C.c_uint32(child), C.c_uint32(w)))


def counter_child_gauss(key, child, gene, ctr1, kind="strict"):
    """Gaussian of gene `gene` of child `child`: from random word 1 + gene of the child's stream"""
    return counter_gauss32(child_word(key, ctr1, child, gene + 1, kind), kind)


def counter_uniform(key, c0, c1, kind="strict"):
    return lib(kind).orc_counter_uniform(C.c_uint32(key), C.c_uint32(c0), C.c_uint32(c1))


def query_key(seed, query, island, kind="strict"):
    return int(lib(kind).orc_query_key(seed, query, island))

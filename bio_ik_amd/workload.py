"""Synthetic IK workloads: the reference's own self-test recipe (README.md:410-418) in batched form.

For query i a TARGET configuration is drawn uniformly inside the joint limits, forward kinematics gives the goal
pose(s) of the tip link(s), and an independent SEED configuration is drawn the same way ("global" workload), or
the seed is the target plus N(0, 0.1 rad) noise clipped to the limits ("tracking", cf. src/ik_test.cpp:95).
The FK used to produce the goals is passed in by the caller (the HIP `eval_fk` on a GPU box)."""
import numpy as np


def _uniform_configs(model, active, rng, n):
    lo = np.asarray(model.var_min)[active]
    hi = np.asarray(model.var_max)[active]
    return lo + (hi - lo) * rng.random((n, len(active)))


def make_queries(template, active_variables, fk_genes, n, seed=0xB101C, kind="global", defaults=None, noise=0.1):
    """Returns (seeds [n][V], goal_params [n][P], targets [n][D]).

    template: ProblemTemplate whose link goals are PoseGoal/PositionGoal/OrientationGoal (their parameters are
    overwritten with the FK of the target); fk_genes(seed_vars[V], genes[n][D]) -> tip frames [n][T][7];
    active_variables: robot variable index per gene (from the problem handle)."""
    from . import abi
    model = template.model
    rng = np.random.default_rng(seed)
    active = np.asarray(active_variables, dtype=np.int64)
    base = model.default_positions() if defaults is None else np.asarray(defaults, dtype=np.float64)
    targets = _uniform_configs(model, active, rng, n)
    if kind == "global":
        seed_genes = _uniform_configs(model, active, rng, n)
    elif kind == "tracking":
        lo = np.asarray(model.var_min)[active]
        hi = np.asarray(model.var_max)[active]
        seed_genes = np.clip(targets + noise * rng.normal(size=targets.shape), lo, hi)
    else:
        raise ValueError(kind)
    seeds = np.tile(base, (n, 1))
    seeds[:, active] = seed_genes
    frames = fk_genes(base, targets)  # [n][T][7]
    # tip index of each link goal = order of first appearance of its link among the template's link goals
    tip_of_link = {}
    for g in template.goals:
        ln = g.link_name()
        if ln is not None and ln not in tip_of_link:
            tip_of_link[ln] = len(tip_of_link)
    params = np.tile(template.pack_params(), (n, 1))
    for g, off in zip(template.goals, template.param_offsets):
        ln = g.link_name()
        if ln is None:
            continue
        t = tip_of_link[ln]
        if g.opcode == abi.GOAL_POSE:
            params[:, off:off + 7] = frames[:, t, :]
        elif g.opcode == abi.GOAL_POSITION:
            params[:, off:off + 3] = frames[:, t, :3]
        elif g.opcode == abi.GOAL_ORIENTATION:
            params[:, off:off + 4] = frames[:, t, 3:]
    return seeds, params, targets


# ---------------------------------------------------------------------------------------------------------------------------
# How many children a solve has walked: the counter RNG on the host (what bench.py's roofline counts for problems with secondary goals)
# ---------------------------------------------------------------------------------------------------------------------------
def philox2x32_10(key, c0, c1):
    """Philox2x32-10 (Random123 constants) on uint32 arrays -- the control-draw generator of the kernels (csrc/bioik_device.h: philox2x32_10)"""
    key = np.asarray(key, dtype=np.uint64) & 0xFFFFFFFF
    c0 = np.asarray(c0, dtype=np.uint64) & 0xFFFFFFFF
    c1 = np.asarray(c1, dtype=np.uint64) & 0xFFFFFFFF
    key, c0, c1 = np.broadcast_arrays(key, c0, c1)
    key, c0, c1 = key.copy(), c0.copy(), c1.copy()
    for r in range(10):
        if r > 0:
            key = (key + 0x9E3779B9) & 0xFFFFFFFF
        p = 0xD256D193 * c0  # (< 2^64: both factors are below 2^32)
        hi, lo = p >> 32, p & 0xFFFFFFFF
        c0 = hi ^ key ^ c1
        c1 = lo
    return c0.astype(np.uint32), c1.astype(np.uint32)


def preselected_children(random_seed, first_query, n_queries, max_steps, population, generations=8):
    """cum[q, s] = the children the kernels WALK (exact FK, primary goals) in the first s steps of query q of a problem with secondary goals: in every
    generation of every step each of the two species walks a random prefix of its pre-selected children, n = o0 % (lambda - 1) + 1 with
    o0 = philox2x32_10(key(q), 0, (step * 16 + generation) << 4 | species << 3 | RNG_PRESELECT)  (bioik_kernels.h: solve_body, `n_eval`;
    reference: ik_evolution_2.cpp:366-378).  A function of the counters alone, so the host can count what the device did from the steps it reports."""
    lam = population  # (bioik_solve_params::population IS lambda, the children per species and generation: bioik_compile.cpp normalize_params, bioik_kernels.h n_eval)
    q = np.arange(n_queries, dtype=np.uint64) + np.uint64(first_query)
    seed = np.uint64(random_seed)
    key, _ = philox2x32_10(seed & np.uint64(0xFFFFFFFF), q & np.uint64(0xFFFFFFFF), (seed >> np.uint64(32)) ^ (q >> np.uint64(32)))  # (island 0: rng_query_key)
    step = np.arange(max_steps, dtype=np.uint64)[None, :, None, None]
    gen = np.arange(generations, dtype=np.uint64)[None, None, :, None]
    species = np.arange(2, dtype=np.uint64)[None, None, None, :]
    ctr1 = ((step * np.uint64(16) + gen) << np.uint64(4)) | (species << np.uint64(3)) | np.uint64(1)
    o0, _ = philox2x32_10(key.astype(np.uint64)[:, None, None, None], np.uint64(0), ctr1)
    per_step = (o0.astype(np.uint64) % np.uint64(lam - 1) + np.uint64(1)).sum(axis=(2, 3))
    cum = np.zeros((n_queries, max_steps + 1), dtype=np.uint64)
    np.cumsum(per_step, axis=1, out=cum[:, 1:])
    return cum

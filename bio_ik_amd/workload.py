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

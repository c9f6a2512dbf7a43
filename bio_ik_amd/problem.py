"""Problem template: the host half of reference Problem::initialize (src/problem.cpp:72-189).

It runs each goal's describe-to-POD hook, resolves link / variable / joint names against the RobotModel
(raising like the reference's ERROR("link not found") / ERROR("joint variable not found"), problem.cpp:125,141)
and produces the `bioik_problem_desc` consumed by `bioik_problem_create`; the tip-link / active-variable
derivation itself happens behind the C-ABI.  `pack_params` flattens the per-query numbers of a goal list with
the same structure."""
import ctypes as C

import numpy as np

from . import abi


class ProblemTemplate:
    def __init__(self, model, group_name, goals, fixed_joints=()):
        self.model = model
        self.group = model.groups[group_name]
        self.goals = list(goals)
        self.fixed_joints = list(fixed_joints)
        n = len(self.goals)
        self._goal_descs = (abi.GoalDesc * max(n, 1))()
        self.param_offsets = []
        off = 0
        for i, g in enumerate(self.goals):
            if g.opcode is None:
                raise NotImplementedError("%s has no device implementation (host-callback goal)" % type(g).__name__)
            d = self._goal_descs[i]
            d.type = g.opcode
            ln = g.link_name()
            d.link = model.link_index(ln) if ln is not None else -1
            vn = g.variable_name()
            d.variable = model.variable_index(vn) if vn is not None else -1
            d.secondary = 1 if g.isSecondary() else 0
            d.weight = g.getWeight()
            self.param_offsets.append(off)
            off += abi.GOAL_PARAM_COUNT[g.opcode]
        self.param_count = off
        self._group_joints = np.asarray(self.group.active_joints, dtype=np.int32)
        self._fixed = np.asarray([model.joint_index(j) for j in self.fixed_joints], dtype=np.int32)

    def desc(self):
        d = abi.ProblemDesc()
        d.struct_size = C.sizeof(abi.ProblemDesc)
        d.n_group_joints = len(self._group_joints)
        d.group_joints = abi.iptr(self._group_joints)
        d.n_goals = len(self.goals)
        d.n_fixed_joints = len(self._fixed)
        d.goals = C.cast(self._goal_descs, C.POINTER(abi.GoalDesc))
        d.fixed_joints = abi.iptr(self._fixed) if len(self._fixed) else None
        return d

    def pack_params(self, goals=None):
        """Flat parameter vector [P] of `goals` (default: the template's own goal objects)."""
        goals = self.goals if goals is None else goals
        if len(goals) != len(self.goals):
            raise ValueError("goal list does not match the problem template")
        out = np.zeros(self.param_count)
        for g, t, off in zip(goals, self.goals, self.param_offsets):
            if g.opcode != t.opcode:
                raise ValueError("goal list does not match the problem template")
            p = g.params()
            out[off:off + len(p)] = p
        return out

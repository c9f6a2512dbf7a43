"""Independent forward kinematics in NumPy long double with ROTATION MATRICES (not quaternions), used to pin the
oracle's quaternion FK (SURVEY.md §4 "golden FK ... cross-checked against an independent restatement").
Follows the URDF/MoveIt definition directly: T_link = T_parent * T_origin * T_joint(q)."""
import numpy as np

LD = np.longdouble


def rot_from_quat(q):
    x, y, z, w = [LD(v) for v in q]
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=LD)


def rot_axis_angle(axis, angle):
    a = np.asarray(axis, dtype=LD)
    a = a / np.sqrt(np.dot(a, a))
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]], dtype=LD)
    s, c = np.sin(LD(angle)), np.cos(LD(angle))
    return np.eye(3, dtype=LD) + s * K + (1 - c) * (K @ K)


def fk_all(model, variables):
    """global (R, p) of every link for one full variable vector."""
    v = np.asarray(variables, dtype=LD).copy()
    for j in range(model.n_links):  # mimic
        if model.joint_mimic[j] >= 0 and model.joint_first_variable[j] >= 0:
            src = model.joint_first_variable[model.joint_mimic[j]]
            v[model.joint_first_variable[j]] = v[src] * LD(model.joint_mimic_factor[j]) + LD(model.joint_mimic_offset[j])
    R = [None] * model.n_links
    p = [None] * model.n_links
    for l in range(model.n_links):
        o = model.link_origin[l]
        Ro, po = rot_from_quat(o[3:7]), np.asarray(o[0:3], dtype=LD)
        jt = model.joint_type[l]
        fv = model.joint_first_variable[l]
        Rj, pj = np.eye(3, dtype=LD), np.zeros(3, dtype=LD)
        if jt == 1:
            Rj = rot_axis_angle(model.joint_axis[l], v[fv])
        elif jt == 2:
            pj = np.asarray(model.joint_axis[l], dtype=LD) * v[fv]
        elif jt == 3:
            pj = v[fv:fv + 3]
            Rj = rot_from_quat(v[fv + 3:fv + 7])
        elif jt == 4:
            pj = np.array([v[fv], v[fv + 1], 0], dtype=LD)
            Rj = rot_axis_angle((0, 0, 1), v[fv + 2])
        Rl, pl = Ro @ Rj, po + Ro @ pj
        par = model.link_parent[l]
        if par >= 0:
            R[l] = R[par] @ Rl
            p[l] = p[par] + R[par] @ pl
        else:
            R[l], p[l] = Rl, pl
    return R, p


def quat_to_rot64(q):
    return np.asarray(rot_from_quat(q), dtype=np.float64)

"""Minimal URDF / SRDF reader -> bio_ik_amd.RobotModel (no MoveIt, no urdfdom).

What the reference gets from moveit::core::RobotModel (src/forward_kinematics.h:192-213, include/bio_ik/robot_info.h:70-106,
src/kinematics_plugin.cpp:167-189) is built here from the robot description itself:
  * links in the order MoveIt's RobotModel::buildRecursive visits them (depth first from the root, siblings in alphabetical order of
    their joint names, as urdfdom's name-keyed joint map yields them), so
    link / variable indices match a MoveIt-loaded model of the same URDF;
  * joints: fixed | revolute | continuous | prismatic | floating | planar, <origin xyz rpy>, <axis> (URDF default 1 0 0),
    <limit lower upper velocity>, <mimic joint multiplier offset>;
  * SRDF <group>: <chain base_link tip_link>, <joint name>, <link name> (= its parent joint), nested <group name>;
    <end_effector parent_link parent_group> supplies the tips of a group without a chain;
    <virtual_joint name type parent_frame child_link> (fixed | floating | planar) puts a new root link `parent_frame` in front of
    the URDF's root, as MoveIt does for a mobile or free-flying base.
Read as well: <inertial> mass and origin of every link (BalanceGoal, goal_types.cpp:236-247).  Not read: collision / visual geometry, transmissions, <safety_controller>, xacro macros."""
import xml.etree.ElementTree as ET

from .robot import RobotModel


def _floats(text, n, default):
    if text is None:
        return tuple(default)
    v = [float(x) for x in text.split()]
    if len(v) != n:
        raise ValueError("expected %d numbers, got %r" % (n, text))
    return tuple(v)


def load_urdf(urdf_xml, srdf_xml=None):
    """urdf_xml / srdf_xml: XML text (not file names).  Returns a RobotModel with the SRDF's groups added."""
    root = ET.fromstring(urdf_xml)
    if root.tag != "robot":
        raise ValueError("not a URDF: root element is <%s>" % root.tag)
    links = [l.get("name") for l in root.findall("link")]
    inertial = {}
    for l in root.findall("link"):
        ine = l.find("inertial")
        if ine is not None and ine.find("mass") is not None:
            o = ine.find("origin")
            inertial[l.get("name")] = (float(ine.find("mass").get("value", 0.0)), _floats(o.get("xyz") if o is not None else None, 3, (0, 0, 0)))
    joints = []
    for j in root.findall("joint"):
        jt = j.get("type")
        if jt not in ("fixed", "revolute", "continuous", "prismatic", "floating", "planar"):
            raise ValueError("joint %r: unsupported type %r" % (j.get("name"), jt))
        o = j.find("origin")
        lim = j.find("limit")
        ax = j.find("axis")
        mim = j.find("mimic")
        joints.append({
            "name": j.get("name"), "type": jt, "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
            "xyz": _floats(o.get("xyz") if o is not None else None, 3, (0, 0, 0)),
            "rpy": _floats(o.get("rpy") if o is not None else None, 3, (0, 0, 0)),
            "axis": _floats(ax.get("xyz") if ax is not None else None, 3, (1, 0, 0)),
            "lower": float(lim.get("lower", 0.0)) if lim is not None else 0.0,
            "upper": float(lim.get("upper", 0.0)) if lim is not None else 0.0,
            "velocity": float(lim.get("velocity", 0.0)) if lim is not None else 0.0,
            "mimic": (mim.get("joint"), float(mim.get("multiplier", 1.0)), float(mim.get("offset", 0.0))) if mim is not None else None,
        })
    children = {l: [] for l in links}
    is_child = set()
    for j in joints:
        if j["parent"] not in children or j["child"] not in children:
            raise ValueError("joint %r references an unknown link" % j["name"])
        children[j["parent"]].append(j)
        if j["child"] in is_child:
            raise ValueError("link %r has two parent joints" % j["child"])
        is_child.add(j["child"])
    # urdfdom keeps a model's joints in a map keyed by joint name and fills every link's child list from it (ModelInterface::initTree):
    # the siblings RobotModel::buildRecursive walks are in alphabetical order of their joint names, whatever the order of the file
    for l in children:
        children[l].sort(key=lambda j: j["name"])
    roots = [l for l in links if l not in is_child]
    if len(roots) != 1:
        raise ValueError("a URDF tree has exactly one root link, found %r" % roots)
    m = RobotModel(root.get("name", "robot"))
    virtual = None
    if srdf_xml is not None:
        vj = ET.fromstring(srdf_xml).findall("virtual_joint")
        if len(vj) > 1:
            raise ValueError("more than one <virtual_joint>")
        if vj:
            virtual = vj[0]
            if virtual.get("child_link") != roots[0]:
                raise ValueError("<virtual_joint> child_link %r is not the root link %r" % (virtual.get("child_link"), roots[0]))
            if virtual.get("type") not in ("fixed", "floating", "planar"):
                raise ValueError("<virtual_joint> type %r" % virtual.get("type"))
    if virtual is not None:
        m.add_link(virtual.get("parent_frame"))
        m.add_link(roots[0], virtual.get("parent_frame"), virtual.get("name"), virtual.get("type"))
    else:
        m.add_link(roots[0])
    stack = [iter(children[roots[0]])]
    while stack:  # depth first, siblings by joint name (RobotModel::buildRecursive over urdfdom's child lists)
        j = next(stack[-1], None)
        if j is None:
            stack.pop()
            continue
        m.add_link(j["child"], j["parent"], j["name"], j["type"], xyz=j["xyz"], rpy=j["rpy"], axis=j["axis"], lower=j["lower"], upper=j["upper"],
                   velocity=j["velocity"])
        stack.append(iter(children[j["child"]]))
    for j in joints:  # a mimicked joint may come later in the file than the joint that follows it
        if j["mimic"] is not None:
            i = m.joint_names.index(j["name"])
            if j["mimic"][0] not in m.joint_names:
                raise ValueError("joint %r mimics unknown joint %r" % (j["name"], j["mimic"][0]))
            m.joint_mimic[i] = m.joint_names.index(j["mimic"][0])
            m.joint_mimic_factor[i] = j["mimic"][1]
            m.joint_mimic_offset[i] = j["mimic"][2]
    m.resolve_mimic_chains()
    for name, (mass, com) in inertial.items():
        i = m.link_names.index(name)
        m.link_mass[i] = mass
        m.link_center[i] = [float(com[0]), float(com[1]), float(com[2])]
    m._keep = None
    if srdf_xml is not None:
        add_srdf_groups(m, srdf_xml)
    return m


def add_srdf_groups(m, srdf_xml):
    s = ET.fromstring(srdf_xml)
    raw = {g.get("name"): g for g in s.findall("group") if len(g)}  # <group name=".."/> inside a group is a reference
    effectors = [(e.get("parent_group"), e.get("parent_link")) for e in s.findall("end_effector")]

    def joints_of(name, seen=()):
        if name in seen:
            raise ValueError("SRDF groups include each other: %r" % (seen + (name,),))
        g = raw[name]
        out, tips = [], []
        for c in g:
            if c.tag == "chain":
                l, b = m.link_names.index(c.get("tip_link")), m.link_names.index(c.get("base_link"))
                tips.append(c.get("tip_link"))
                while l != b:
                    if l < 0:
                        raise ValueError("group %r: %r is not an ancestor of %r" % (name, c.get("base_link"), c.get("tip_link")))
                    out.append(l)
                    l = m.link_parent[l]
            elif c.tag == "joint":
                out.append(m.joint_names.index(c.get("name")))
            elif c.tag == "link":
                out.append(m.link_names.index(c.get("name")))  # the link's parent joint
            elif c.tag == "group":
                o2, t2 = joints_of(c.get("name"), seen + (name,))
                out += o2
                tips += t2
        return out, tips

    for name in raw:
        idx, tips = joints_of(name)
        idx = sorted(set(i for i in idx if m.link_parent[i] >= 0))  # model order, like JointModelGroup::getActiveJointModels
        if not tips:
            tips = [link for grp, link in effectors if grp == name]
        if not tips:  # leaves of the group's joint set
            inner = set(m.link_parent[i] for i in idx)
            tips = [m.link_names[i] for i in idx if i not in inner]
        m.add_group(name, joints=[m.joint_names[i] for i in idx], tips=tips)
    return m

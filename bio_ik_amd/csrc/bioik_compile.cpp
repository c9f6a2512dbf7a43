// bioik_compile.cpp — host half of the drop-in boundary.
//
// HostModel restates what the reference reads from moveit::core::RobotModel (src/forward_kinematics.h:192-213,
// include/bio_ik/robot_info.h:70-106); HostProblem restates Problem::initialize (src/problem.cpp:72-228) and the
// link schedule of RobotFK_Fast_Base::initialize (src/forward_kinematics.h:253-330), then compiles both into the
// flat joint program (DevProblem) the gfx950 kernels walk: fixed links are folded into per-joint constant frames,
// branch frames get LDS slots, goals are grouped by the tip they read.
#include "bioik_compile.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace bioik {

static Frame identity() { return Frame{{0, 0, 0}, {0, 0, 0, 1}}; }
static void qrot(const double* q, const double* v, double* r) {
    double tx = q[1] * v[2] - q[2] * v[1], ty = q[2] * v[0] - q[0] * v[2], tz = q[0] * v[1] - q[1] * v[0];
    double rx = q[3] * tx + q[1] * tz - q[2] * ty, ry = q[3] * ty + q[2] * tx - q[0] * tz, rz = q[3] * tz + q[0] * ty - q[1] * tx;
    r[0] = rx + rx + v[0];
    r[1] = ry + ry + v[1];
    r[2] = rz + rz + v[2];
}
static void qmul(const double* p, const double* q, double* r) {
    double x = (p[3] * q[0] + p[0] * q[3]) + (p[1] * q[2] - p[2] * q[1]);
    double y = (p[3] * q[1] - p[0] * q[2]) + (p[1] * q[3] + p[2] * q[0]);
    double z = (p[3] * q[2] + p[0] * q[1]) - (p[1] * q[0] - p[2] * q[3]);
    double w = (p[3] * q[3] - p[0] * q[0]) - (p[1] * q[1] + p[2] * q[2]);
    r[0] = x, r[1] = y, r[2] = z, r[3] = w;
}
static Frame concat(const Frame& a, const Frame& b) {
    Frame r;
    double d[3];
    qrot(a.q, b.p, d);
    for (int i = 0; i < 3; i++) r.p[i] = a.p[i] + d[i];
    qmul(a.q, b.q, r.q);
    return r;
}
static bool is_identity(const Frame& f) { return f.p[0] == 0 && f.p[1] == 0 && f.p[2] == 0 && f.q[0] == 0 && f.q[1] == 0 && f.q[2] == 0 && f.q[3] == 1; }

int goal_param_count(int type) {
    switch (type) {
        case BIOIK_GOAL_POSITION: return 3;
        case BIOIK_GOAL_ORIENTATION: return 4;
        case BIOIK_GOAL_POSE: return 8;
        case BIOIK_GOAL_LOOK_AT: return 6;
        case BIOIK_GOAL_MAX_DISTANCE: return 4;
        case BIOIK_GOAL_MIN_DISTANCE: return 4;
        case BIOIK_GOAL_LINE: return 6;
        case BIOIK_GOAL_PLANE: return 6;
        case BIOIK_GOAL_AVOID_JOINT_LIMITS: return 0;
        case BIOIK_GOAL_CENTER_JOINTS: return 0;
        case BIOIK_GOAL_REGULARIZATION: return 0;
        case BIOIK_GOAL_MINIMAL_DISPLACEMENT: return 0;
        case BIOIK_GOAL_JOINT_VARIABLE: return 1;
        case BIOIK_GOAL_SIDE: return 6;
        case BIOIK_GOAL_DIRECTION: return 6;
        case BIOIK_GOAL_CONE: return 11;
        case BIOIK_GOAL_BALANCE: return 6;
    }
    return -1;
}

static int joint_var_count(int type) {
    switch (type) {
        case BIOIK_JOINT_FIXED: return 0;
        case BIOIK_JOINT_REVOLUTE: return 1;
        case BIOIK_JOINT_PRISMATIC: return 1;
        case BIOIK_JOINT_FLOATING: return 7;
        case BIOIK_JOINT_PLANAR: return 3;
    }
    throw Error(BIOIK_ERR_INVALID_ARGUMENT, "unknown joint type");
}

HostModel::HostModel(const bioik_model_desc& d) {
    if (d.struct_size != sizeof(bioik_model_desc)) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_model_desc: struct_size mismatch");
    if (d.n_links == 0) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "model has no links");
    if (!d.link_parent || !d.link_origin || !d.joint_type || !d.joint_axis || !d.joint_first_variable)
        throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_model_desc: null link array");
    if (d.n_variables > 0 && (!d.var_min || !d.var_max || !d.var_bounded || !d.var_max_velocity))
        throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_model_desc: null variable array");
    links.resize(d.n_links);
    std::vector<int> var_joint(d.n_variables, -1);
    for (uint32_t i = 0; i < d.n_links; i++) {
        Link& l = links[i];
        l.parent = d.link_parent[i];
        if (l.parent >= (int)i || l.parent < -1) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "links must be ordered parent before child");
        for (int c = 0; c < 3; c++) l.origin.p[c] = d.link_origin[7 * i + c];
        for (int c = 0; c < 4; c++) l.origin.q[c] = d.link_origin[7 * i + 3 + c];
        l.type = d.joint_type[i];
        for (int c = 0; c < 3; c++) l.axis[c] = d.joint_axis[3 * i + c];
        l.first_var = d.joint_first_variable[i];
        l.var_count = joint_var_count(l.type);
        l.mimic = d.joint_mimic ? d.joint_mimic[i] : -1;
        l.mimic_factor = (l.mimic >= 0 && d.joint_mimic_factor) ? d.joint_mimic_factor[i] : 1.0;
        l.mimic_offset = (l.mimic >= 0 && d.joint_mimic_offset) ? d.joint_mimic_offset[i] : 0.0;
        if (l.mimic >= (int)d.n_links || l.mimic == (int)i) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "mimic joint index out of range");
        l.mass = d.link_mass ? d.link_mass[i] : 0.0;
        if (d.link_mass && !d.link_center) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_model_desc: link_mass without link_center");
        for (int c = 0; c < 3; c++) l.center[c] = d.link_center ? d.link_center[3 * i + c] : 0.0;
        if (l.var_count > 0) {
            if (l.first_var < 0 || l.first_var + l.var_count > (int)d.n_variables) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "joint variable index out of range");
            for (int v = 0; v < l.var_count; v++) var_joint[l.first_var + v] = (int)i;
        }
    }
    // A joint that mimics a joint that itself mimics another: resolved to the joint at the end of the chain with the composed factor and offset, as MoveIt's
    // RobotModel::buildMimic does before bio_ik ever sees the model (so forward_kinematics.h:230-246 only meets plain mimics):
    //     x = f1 (f2 y + o2) + o1  ->  factor f1 f2, offset o1 + f1 o2
    for (size_t pass = 0;; pass++) {
        bool changed = false;
        for (Link& l : links)
            if (l.mimic >= 0 && links[l.mimic].mimic >= 0) {
                const Link& via = links[l.mimic];
                l.mimic_offset = l.mimic_offset + l.mimic_factor * via.mimic_offset;
                l.mimic_factor = l.mimic_factor * via.mimic_factor;
                l.mimic = via.mimic;
                changed = true;
            }
        if (!changed) break;
        if (pass > links.size()) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "mimic joints that follow each other in a circle");
    }
    // RobotInfo (robot_info.h:70-106)
    vars.resize(d.n_variables);
    for (uint32_t v = 0; v < d.n_variables; v++) {
        Var info;
        bool bounded = d.var_bounded[v] != 0;
        int j = var_joint[v];
        if (j >= 0 && links[j].type == BIOIK_JOINT_REVOLUTE)
            if (d.var_max[v] - d.var_min[v] >= 2 * M_PI * 0.9999) bounded = false;  // :82-84
        info.vmin = d.var_min[v];
        info.vmax = d.var_max[v];
        info.clip_min = bounded ? info.vmin : -DBL_MAX;
        info.clip_max = bounded ? info.vmax : +DBL_MAX;
        info.span = info.vmax - info.vmin;
        if (!(info.span >= 0 && info.span < FLT_MAX)) info.span = 1;  // :94
        double mv = d.var_max_velocity[v];
        info.max_velocity_rcp = mv > 0.0 ? 1.0 / mv : 0.0;
        info.joint = j;
        vars[v] = info;
    }
}

HostProblem::HostProblem(const HostModel* m, const bioik_problem_desc& d) : model(m) {
    if (d.struct_size != sizeof(bioik_problem_desc)) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_problem_desc: struct_size mismatch");
    const int nl = (int)m->links.size(), nv = (int)m->vars.size();
    std::memset(&dev, 0, sizeof(dev));
    dev.multi_op = -1;

    // ---- Problem::initialize, problem.cpp:72-228 ----
    std::vector<char> group_variable(nv, 0), fixed_joint(nl, 0);
    std::vector<int> group_joints(d.group_joints, d.group_joints + d.n_group_joints);
    for (int j : group_joints) {
        if (j < 0 || j >= nl) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "group joint out of range");
        for (int v = 0; v < m->links[j].var_count; v++) group_variable[m->links[j].first_var + v] = 1;
    }
    for (uint32_t i = 0; i < d.n_fixed_joints; i++) {
        int j = d.fixed_joints[i];
        if (j < 0 || j >= nl) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "fixed joint out of range");
        fixed_joint[j] = 1;
    }
    std::vector<long> link_tip(nl, -1);
    auto add_tip_link = [&](int link) -> int {  // problem.cpp:57-65
        if (link_tip[link] < 0) {
            link_tip[link] = (long)tip_links.size();
            tip_links.push_back(link);
        }
        return (int)link_tip[link];
    };
    auto add_active_variable = [&](int ivar) -> long {  // problem.cpp:103-126
        int joint = m->vars[ivar].joint;
        if (joint >= 0 && fixed_joint[joint]) return (long)-1 - (long)ivar;
        for (size_t i = 0; i < active_variables.size(); i++)
            if (active_variables[i] == ivar) return (long)i;
        if (group_variable[ivar]) {
            active_variables.push_back(ivar);
            return (long)active_variables.size() - 1;
        }
        throw Error(BIOIK_ERR_NOT_FOUND, "joint variable not found");
    };
    struct G {
        int type, tip;
        long var;
        double weight;
        int secondary, param_off;
    };
    std::vector<G> goals;
    double balance_total = 0.0;
    int n_balance_goals = 0;
    for (uint32_t gi = 0; gi < d.n_goals; gi++) {
        const bioik_goal_desc& g = d.goals[gi];
        int np = goal_param_count(g.type);
        if (np < 0) throw Error(BIOIK_ERR_UNSUPPORTED, "goal type has no device implementation");
        G info{g.type, -1, LONG_MIN, g.weight, g.secondary != 0, param_count};
        if (g.link >= 0) {
            if (g.link >= nl) throw Error(BIOIK_ERR_NOT_FOUND, "link not found");
            info.tip = add_tip_link(g.link);
        }
        if (g.variable >= 0) {
            if (g.variable >= nv) throw Error(BIOIK_ERR_NOT_FOUND, "joint variable not found");
            info.var = add_active_variable(g.variable);
        }
        if (g.type == BIOIK_GOAL_BALANCE) {  // BalanceGoal::describe (goal_types.cpp:231-255): every link with mass becomes a tip, in link order
            if (g.secondary) throw Error(BIOIK_ERR_UNSUPPORTED, "BalanceGoal cannot be a secondary goal (the reference's has no such constructor)");
            double total = 0.0;
            for (int l = 0; l < nl; l++)
                if (m->links[l].mass > 0) total += m->links[l].mass, add_tip_link(l);
            if (!(total > 0)) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "BalanceGoal: the model has no link with a positive mass (bioik_model_desc::link_mass)");
            balance_total = total;
            n_balance_goals++;
        }
        param_count += np;
        goals.push_back(info);
    }
    {  // active variables of the active subtree, problem.cpp:191-204
        std::vector<char> usage(nl, 0);
        for (int tip : tip_links)
            for (int l = tip; l >= 0; l = m->links[l].parent) usage[l] = 1;
        for (int j = 0; j < nl; j++)
            if (fixed_joint[j]) usage[j] = 0;
        for (int j : group_joints)
            if (usage[j] && m->links[j].mimic < 0)
                for (int v = 0; v < m->links[j].var_count; v++) add_active_variable(m->links[j].first_var + v);
    }
    const int D = (int)active_variables.size();
    std::vector<double> vw(D);  // problem.cpp:206-225
    {
        double s = 0;
        for (int v : active_variables) s += m->vars[v].max_velocity_rcp;
        for (int i = 0; i < D; i++) vw[i] = s > 0 ? m->vars[active_variables[i]].max_velocity_rcp / s : 1.0 / D;
    }

    // ---- link schedule (forward_kinematics.h:268-282) folded into the joint program ----
    std::vector<int> schedule;
    std::vector<char> scheduled(nl, 0);
    for (int tip : tip_links) {
        std::vector<int> chain;
        for (int l = tip; l >= 0; l = m->links[l].parent) chain.push_back(l);
        std::reverse(chain.begin(), chain.end());
        for (int l : chain)
            if (!scheduled[l]) scheduled[l] = 1, schedule.push_back(l);
    }
    std::vector<int> gene_of_var(nv, -1);
    for (int i = 0; i < D; i++) gene_of_var[active_variables[i]] = i;
    std::vector<int> src_of(nl, -1), op_of_link(nl, -1);
    std::vector<Frame> c_of(nl, identity());
    std::vector<DevOp> ops;
    // A branch whose parent frame is not the running frame: if the chain root -> parent is short (a torso in front of two
    // arms), walk it again in front of the branch instead of parking the parent frame in LDS (7 doubles per lane and
    // slot set).  The repeated ops read the value of the op they repeat (mimic factor 1) and are on no tip's chain mask,
    // so genes, Jacobian columns and the linear model are untouched and the frames are the same bits.
    auto rewalk_short_parent_chain = [&](DevOp& op, int base_src, bool multi) {
        if (!multi && base_src >= 0 && base_src != (int)ops.size() - 1) {
            std::vector<int> chain;
            bool plain = true;
            for (int o = base_src; o >= 0; o = ops[o].src) {
                chain.push_back(o);
                plain = plain && (ops[o].type == BIOIK_OP_REVOLUTE || ops[o].type == BIOIK_OP_PRISMATIC);
            }
            if (plain && chain.size() <= 2 && ops.size() + chain.size() < (size_t)BIOIK_MAX_OPS) {
                int prev = -1;
                for (size_t c = chain.size(); c-- > 0;) {
                    DevOp rep = ops[chain[c]];
                    rep.gene = -1;
                    rep.src = prev;
                    rep.load_slot = rep.save_slot = -1;
                    if (rep.mimic_src == -1) rep.mimic_src = chain[c], rep.mimic_factor = 1.0, rep.mimic_offset = 0.0;
                    prev = (int)ops.size();
                    ops.push_back(rep);
                }
                op.src = prev;
            }
        }
    };
    // BIOIK_COMPILE_EXACT=1 (diagnostics): the joint program WITHOUT folding.  By default a joint's origin and the fixed links in front of it become one
    // constant frame of the joint's op -- p + R (o + v a) where the reference concatenates frame by frame, (p + R o) + R' (v a): the same frames to the last
    // bit or two, exactly the same only for origins without rotation whose offsets do not add up.  Here every origin that is not the identity (a joint's, a
    // fixed link's) is an op of its own -- a revolute op whose angle is the constant 0 (it "mimics" itself with factor 0), so that its frame is the constant
    // alone -- and the joint's op behind it carries the bare joint: parent o origin o joint in the reference's association (forward_kinematics.h:283-330), on
    // any robot.  About twice the ops; such a program is not a serial chain, so the kernels compiled for one do not run it.
    const bool exact_program = std::getenv("BIOIK_COMPILE_EXACT") != nullptr && std::atoi(std::getenv("BIOIK_COMPILE_EXACT")) != 0;
    int some_joint_var = -1;  // (the variable a constant op of a FIXED link is filed under: any -- its value is multiplied by nought)
    for (int l : schedule)
        if (m->links[l].type != BIOIK_JOINT_FIXED && some_joint_var < 0) some_joint_var = m->links[l].first_var;
    if (some_joint_var < 0 && D > 0) some_joint_var = active_variables[0];  // (every tip hangs off the root behind fixed links: the genes are goal variables off the chains)
    for (int l : schedule) {
        const HostModel::Link& L = m->links[l];
        int base_src = L.parent >= 0 ? src_of[L.parent] : -1;
        Frame base_c = L.parent >= 0 ? c_of[L.parent] : identity();
        Frame C = concat(base_c, L.origin);
        if (exact_program && some_joint_var >= 0) {
            if (!is_identity(L.origin)) {
                DevOp op;
                std::memset(&op, 0, sizeof(op));
                op.type = BIOIK_OP_REVOLUTE;
                op.var = L.type != BIOIK_JOINT_FIXED ? L.first_var : some_joint_var;
                op.gene = -1;
                op.src = base_src;
                op.load_slot = op.save_slot = -1;
                op.val_first = op.joint_op = -1, op.multi_slot = -1;
                for (int c = 0; c < 3; c++) op.cpos[c] = L.origin.p[c], op.axis[c] = c == 2 ? 1.0 : 0.0;
                for (int c = 0; c < 4; c++) op.ca[c] = L.origin.q[c];
                double aq[4] = {0.0, 0.0, 1.0, 0.0};
                qmul(L.origin.q, aq, op.cb);  // (multiplied by sin 0 = 0)
                rewalk_short_parent_chain(op, base_src, false);
                op.mimic_src = (int)ops.size(), op.mimic_factor = 0.0, op.mimic_offset = 0.0;  // its own value times nought: the angle 0
                base_src = (int)ops.size();
                ops.push_back(op);
            }
            C = identity();
        }
        if (L.type == BIOIK_JOINT_FIXED) {
            src_of[l] = base_src;
            c_of[l] = C;
            continue;
        }
        if (L.mimic >= 0 && L.type != BIOIK_JOINT_REVOLUTE && L.type != BIOIK_JOINT_PRISMATIC)
            throw Error(BIOIK_ERR_UNSUPPORTED, "floating / planar mimic joints have no device implementation in this version");
        DevOp op;
        std::memset(&op, 0, sizeof(op));
        op.type = L.type == BIOIK_JOINT_REVOLUTE ? BIOIK_OP_REVOLUTE : L.type == BIOIK_JOINT_PRISMATIC ? BIOIK_OP_PRISMATIC :
                  L.type == BIOIK_JOINT_FLOATING ? BIOIK_OP_FLOATING : BIOIK_OP_PLANAR;
        op.var = L.first_var;
        op.gene = gene_of_var[L.first_var];
        op.src = base_src;
        op.load_slot = op.save_slot = -1;
        op.mimic_src = -1, op.mimic_factor = 1.0, op.mimic_offset = 0.0;
        op.val_first = op.joint_op = -1, op.multi_slot = -1;
        bool multi = op.type >= BIOIK_OP_FLOATING;
        op.multi_slot = -1;
        if (multi) {
            // A floating / planar joint, anywhere on the chains and any number of them (forward_kinematics.h:120-135, 331-354 take them wherever they are).  Its
            // variables are value ops of their own (below).  Its joint frame J(values) does not depend on the frames in front of it: an out-of-line call
            // computes it in front of a chain walk and parks it in an LDS slot; inside the walk the op applies its constant frame like any other
            // (F_src o C, cb = 0) and then the parked joint frame: (F_src o C) o J, the association of the reference's three-frame concat.
            op.gene = -1;
            if (dev.multi_op < 0) dev.multi_op = (int)ops.size();
        }
        if (L.mimic >= 0) {  // resolved to an op index below, once every op exists
            op.gene = -1;
            op.mimic_src = -2 - L.mimic;
            op.mimic_factor = L.mimic_factor, op.mimic_offset = L.mimic_offset;
        }
        for (int c = 0; c < 3; c++) op.cpos[c] = C.p[c], op.axis[c] = L.axis[c];
        for (int c = 0; c < 4; c++) op.ca[c] = C.q[c];
        if (multi) {
            for (int c = 0; c < 4; c++) op.cb[c] = 0.0;
        } else if (op.type == BIOIK_OP_REVOLUTE) {
            double aq[4] = {L.axis[0], L.axis[1], L.axis[2], 0.0};
            qmul(C.q, aq, op.cb);
        } else {
            qrot(C.q, L.axis, op.cb);
            op.cb[3] = 0.0;
        }
        rewalk_short_parent_chain(op, base_src, multi);
        int k = (int)ops.size();
        ops.push_back(op);
        src_of[l] = k;
        op_of_link[l] = k;
        c_of[l] = identity();
    }
    const int n_chain = (int)ops.size();
    // floating / planar joints: one value-only op per variable (7 / 3 consecutive), genes where the variable is active
    std::vector<char> var_has_op(nv, 0);
    for (int k = 0; k < n_chain; k++) {
        if (ops[k].type < BIOIK_OP_FLOATING) {
            if (ops[k].mimic_src != k) var_has_op[ops[k].var] = 1;  // (not the constant ops of BIOIK_COMPILE_EXACT: they read no variable)
            continue;
        }
        const int cnt = ops[k].type == BIOIK_OP_FLOATING ? 7 : 3;
        ops[k].val_first = (int)ops.size();
        for (int c = 0; c < cnt; c++) {
            DevOp op;
            std::memset(&op, 0, sizeof(op));
            op.type = BIOIK_OP_NONE;
            op.var = ops[k].var + c;
            op.gene = gene_of_var[op.var];
            op.src = op.load_slot = op.save_slot = -1;
            op.mimic_src = -1, op.mimic_factor = 1.0, op.mimic_offset = 0.0;
            op.val_first = -1, op.joint_op = k, op.multi_slot = -1;
            var_has_op[op.var] = 1;
            ops.push_back(op);
        }
        if (ops[k].type == BIOIK_OP_FLOATING && gene_of_var[ops[k].var + 3] >= 0) {  // ik_evolution_2.cpp:203-215
            for (int c = 4; c < 7; c++)
                if (gene_of_var[ops[k].var + c] != gene_of_var[ops[k].var + 3] + (c - 3))
                    throw Error(BIOIK_ERR_UNSUPPORTED, "the four orientation variables of a floating joint must be consecutive genes");
            if (dev.n_quat >= 4) throw Error(BIOIK_ERR_UNSUPPORTED, "more than 4 floating joints with active orientation");
            dev.quat_op[dev.n_quat++] = ops[k].val_first + 3;
        }
    }
    for (int i = 0; i < D; i++) {  // active variables that move no scheduled link (goal variables off the chains)
        int v = active_variables[i];
        if (var_has_op[v]) continue;
        int j = m->vars[v].joint;
        if (j >= 0 && (m->links[j].type != BIOIK_JOINT_REVOLUTE && m->links[j].type != BIOIK_JOINT_PRISMATIC))
            throw Error(BIOIK_ERR_UNSUPPORTED, "variables of floating / planar joints outside the goal chains have no device implementation in this version");
        DevOp op;
        std::memset(&op, 0, sizeof(op));
        op.type = BIOIK_OP_NONE;
        op.var = v;
        op.gene = i;
        op.src = op.load_slot = op.save_slot = -1;
        op.mimic_src = -1, op.mimic_factor = 1.0, op.mimic_offset = 0.0;
        op.val_first = op.joint_op = -1, op.multi_slot = -1;
        var_has_op[v] = 1;
        ops.push_back(op);
    }
    // mimic joints read the value of the joint they follow: resolve it to an op (a followed joint that is on no goal chain
    // gets a value-only op; it is not a gene then, so it carries the seed's value)
    std::vector<int> value_op_of_var(nv, -1);
    for (size_t k = 0; k < ops.size(); k++)
        if (ops[k].mimic_src == -1) value_op_of_var[ops[k].var] = (int)k;
    for (int k = 0; k < n_chain; k++) {
        if (ops[k].mimic_src >= -1) continue;
        const int src_link = -2 - ops[k].mimic_src;
        const int sv = m->links[src_link].first_var;
        if (sv < 0 || m->links[src_link].var_count != 1) throw Error(BIOIK_ERR_UNSUPPORTED, "mimic of a joint without exactly one variable");
        if (value_op_of_var[sv] < 0) {
            DevOp op;
            std::memset(&op, 0, sizeof(op));
            op.type = BIOIK_OP_NONE;
            op.var = sv;
            op.gene = gene_of_var[sv];
            op.src = op.load_slot = op.save_slot = -1;
            op.mimic_src = -1, op.mimic_factor = 1.0, op.mimic_offset = 0.0;
            op.val_first = op.joint_op = -1, op.multi_slot = -1;
            if (op.gene >= 0) {
                const HostModel::Var& vi = m->vars[sv];
                op.clip_min = vi.clip_min, op.clip_max = vi.clip_max, op.span = vi.span, op.vmin = vi.vmin, op.vmax = vi.vmax;
                op.unbounded = vi.clip_max == DBL_MAX;
                op.vw = vw[op.gene];
                dev.op_of_gene[op.gene] = (int)ops.size();
            }
            value_op_of_var[sv] = (int)ops.size();
            ops.push_back(op);
        }
        ops[k].mimic_src = value_op_of_var[sv];
    }
    if ((int)ops.size() > BIOIK_MAX_OPS)
        throw Error(BIOIK_ERR_UNSUPPORTED, exact_program ? "more than 64 ops in the unfolded joint program (BIOIK_COMPILE_EXACT: an op per origin and per joint)"
                                                         : "more than 64 moving joints on the goal chains");
    if ((int)tip_links.size() > BIOIK_MAX_TIPS) throw Error(BIOIK_ERR_UNSUPPORTED, "more than 64 tip links");
    if (n_balance_goals > BIOIK_MAX_BALANCE) throw Error(BIOIK_ERR_UNSUPPORTED, "more than 4 BalanceGoals");
    for (size_t k = 0; k < ops.size(); k++) {
        DevOp& op = ops[k];
        const HostModel::Var& vi = m->vars[op.var];
        op.clip_min = vi.clip_min, op.clip_max = vi.clip_max, op.span = vi.span, op.vmin = vi.vmin, op.vmax = vi.vmax;
        op.unbounded = vi.clip_max == DBL_MAX;
        op.vw = op.gene >= 0 ? vw[op.gene] : 0.0;
        if (op.gene >= 0) dev.op_of_gene[op.gene] = (int)k;
    }
    // branch frames: an op whose parent frame is not the running frame fetches it from an LDS slot
    int n_slots = 0;
    for (int k = 0; k < n_chain; k++)
        if (ops[k].type >= BIOIK_OP_FLOATING) ops[k].multi_slot = n_slots++;
    for (int k = 0; k < n_chain; k++) {
        int s = ops[k].src;
        if (s >= 0 && s != k - 1) {
            if (ops[s].save_slot < 0) ops[s].save_slot = n_slots++;
            ops[k].load_slot = ops[s].save_slot;
        }
    }
    // tips, ordered by the op that completes them
    struct TipTmp {
        int pub, src;
        Frame e;
    };
    std::vector<TipTmp> tt;
    for (size_t t = 0; t < tip_links.size(); t++) tt.push_back(TipTmp{(int)t, src_of[tip_links[t]], c_of[tip_links[t]]});
    std::stable_sort(tt.begin(), tt.end(), [](const TipTmp& a, const TipTmp& b) { return a.src < b.src; });
    const int T = (int)tt.size();
    for (int k = 0; k < (int)ops.size(); k++) ops[k].tip_first = 0, ops[k].tip_count = 0;
    for (int t = 0; t < T; t++) {
        DevTip& dt = dev.tips[t];
        dt.src = tt[t].src;
        dt.has_e = !is_identity(tt[t].e);
        for (int c = 0; c < 3; c++) dt.e[c] = tt[t].e.p[c];
        for (int c = 0; c < 4; c++) dt.e[3 + c] = tt[t].e.q[c];
        dt.out_index = tt[t].pub;
        dev.tip_of_out[tt[t].pub] = t;
        uint64_t mask = 0;
        for (int l = tip_links[tt[t].pub]; l >= 0; l = m->links[l].parent)
            if (op_of_link[l] >= 0) mask |= 1ull << op_of_link[l];
        dt.dep_mask = mask;
        dt.bal_w = 0.0;
        if (n_balance_goals > 0 && m->links[tip_links[tt[t].pub]].mass > 0) {
            const HostModel::Link& L = m->links[tip_links[tt[t].pub]];
            dt.bal_w = L.mass / balance_total;  // goal_types.cpp:252-254
            for (int c = 0; c < 3; c++) dt.bal_c[c] = L.center[c];
        }
        if (dt.src < 0) {
            dev.n_root_tips++;
        } else {
            DevOp& op = ops[dt.src];
            if (op.tip_count == 0) op.tip_first = t;
            op.tip_count++;
        }
    }
    // goals: primary link goals grouped by device tip, then gene-only primary goals; secondary goals in order
    auto to_dev = [&](const G& g) {
        DevGoal o;
        std::memset(&o, 0, sizeof(o));
        o.weight_sq = g.weight * g.weight;
        o.type = g.type;
        o.tip = g.tip >= 0 ? dev.tip_of_out[g.tip] : -1;
        o.var_op = -1, o.var_seed = -1;
        if (g.var != LONG_MIN) {
            if (g.var >= 0) o.var_op = dev.op_of_gene[g.var];
            else o.var_seed = (int)(-1 - g.var);
        }
        o.param_off = g.param_off;
        return o;
    };
    int np = 0, ns = 0;
    for (int t = 0; t < T; t++) {
        dev.tips[t].goal_first = np;
        for (const G& g : goals)
            if (!g.secondary && g.tip >= 0 && dev.tip_of_out[g.tip] == t) {
                if (np >= BIOIK_MAX_GOALS) throw Error(BIOIK_ERR_UNSUPPORTED, "too many goals");
                dev.primary[np++] = to_dev(g);
            }
        dev.tips[t].goal_count = np - dev.tips[t].goal_first;
        const bool one_pose = dev.tips[t].goal_count == 1 && dev.primary[dev.tips[t].goal_first].type == BIOIK_GOAL_POSE;
        dev.tips[t].pose_off = one_pose ? dev.primary[dev.tips[t].goal_first].param_off : -1;
        dev.tips[t].pose_weight_sq = one_pose ? dev.primary[dev.tips[t].goal_first].weight_sq : 0.0;
    }
    dev.n_link_primary = np;
    for (const G& g : goals) {
        if (g.type == BIOIK_GOAL_BALANCE) {  // reads many tips: kept apart from the per-tip and the gene-only goals
            dev.balance[dev.n_balance++] = to_dev(g);
        } else if (g.secondary) {
            if (ns >= BIOIK_MAX_GOALS) throw Error(BIOIK_ERR_UNSUPPORTED, "too many goals");
            dev.secondary[ns++] = to_dev(g);
        } else if (g.tip < 0) {
            if (np >= BIOIK_MAX_GOALS) throw Error(BIOIK_ERR_UNSUPPORTED, "too many goals");
            dev.primary[np++] = to_dev(g);
        }
    }
    {   // `jac`: tipObjectives[goal.tip_index] = goal.frame over the primary goals in order (ik_gradient.cpp:64-66); a goal without a link
        // has tip_index 0 and an identity frame (problem.cpp:152-155), so it resets tip 0's objective
        std::vector<int> obj_type(T, -1), obj_off(T, -1);
        for (const G& g : goals) {
            if (g.secondary) continue;
            const int pub = g.tip >= 0 ? g.tip : 0;
            if (pub >= T) continue;
            const bool framed = g.tip >= 0 && (g.type == BIOIK_GOAL_POSITION || g.type == BIOIK_GOAL_ORIENTATION || g.type == BIOIK_GOAL_POSE);
            obj_type[pub] = framed ? g.type : -1;
            obj_off[pub] = framed ? g.param_off : -1;
        }
        for (int pub = 0; pub < T; pub++) dev.tips[dev.tip_of_out[pub]].obj_type = obj_type[pub], dev.tips[dev.tip_of_out[pub]].obj_param_off = obj_off[pub];
    }
    dev.n_primary = np;
    dev.n_secondary = ns;
    dev.pose_only = (T == 1 && np == 1 && ns == 0 && dev.n_balance == 0 && dev.n_link_primary == 1 && dev.primary[0].type == BIOIK_GOAL_POSE &&
                     dev.primary[0].tip == 0) ? 1 : 0;
    dev.pose_param_off = dev.pose_only ? dev.primary[0].param_off : 0;
    dev.pose_weight_sq = dev.pose_only ? dev.primary[0].weight_sq : 0.0;
    dev.n_ops = (int)ops.size();
    dev.n_chain_ops = n_chain;
    // the memetic phase gives gene i to lane i of a 64-lane wavefront and the unperturbed elite to lane D
    if (D > 63) throw Error(BIOIK_ERR_UNSUPPORTED, "more than 63 active variables");
    dev.D = D;
    dev.T = T;
    dev.V = nv;
    dev.P = param_count;
    dev.n_slots = n_slots;
    int n_prefix = 0;
    while (n_prefix < n_chain && ops[n_prefix].gene < 0 && ops[n_prefix].mimic_src < 0 && ops[n_prefix].type < BIOIK_OP_FLOATING &&
           ops[n_prefix].src == n_prefix - 1 &&
           ops[n_prefix].tip_count == 0 && ops[n_prefix].save_slot < 0 && ops[n_prefix].load_slot < 0)
        n_prefix++;
    if (n_prefix == n_chain) n_prefix = 0;  // nothing left to walk: no point
    dev.n_prefix = n_prefix;
    dev.serial_chain = n_chain > 0 ? 1 : 0;
    for (int k = 0; k < n_chain; k++)
        if (ops[k].src != k - 1 || ops[k].load_slot >= 0 || ops[k].save_slot >= 0 || ops[k].mimic_src >= 0 || ops[k].type >= BIOIK_OP_FLOATING) dev.serial_chain = 0;
    dev.reserved0 = 0;
    dev.genes_follow_ops = 1;
    for (int i = 1; i < D; i++)
        if (dev.op_of_gene[i] <= dev.op_of_gene[i - 1]) dev.genes_follow_ops = 0;
    // sparsity classes of the revolute ops' constants (bioik_types.h: BIOIK_POS_*, BIOIK_ROT_*): exact zeros only, so the terms the
    // walk drops are exact no-ops of the general transform
    for (size_t k = 0; k < ops.size(); k++) {
        DevOp& op = ops[k];
        op.pos_kind = BIOIK_POS_GENERAL, op.rot_kind = BIOIK_ROT_GENERAL;
#ifndef BIOIK_NO_SPARSE_OPS
        if (op.type != BIOIK_OP_REVOLUTE) continue;
        const bool zx = op.cpos[0] == 0.0, zy = op.cpos[1] == 0.0, zz = op.cpos[2] == 0.0;
        if (zx && zy && zz) op.pos_kind = BIOIK_POS_ZERO;
        else if (zy && zz) op.pos_kind = BIOIK_POS_X;
        else if (zx && zz) op.pos_kind = BIOIK_POS_Y;
        else if (zx && zy) op.pos_kind = BIOIK_POS_Z;
        if (op.ca[0] == 0.0 && op.ca[1] == 0.0 && op.ca[2] == 0.0 && op.ca[3] == 1.0 && op.cb[3] == 0.0) {
            const bool bx = op.cb[0] == 0.0, by = op.cb[1] == 0.0, bz = op.cb[2] == 0.0;
            if (by && bz && !bx) op.rot_kind = BIOIK_ROT_X;
            else if (bx && bz && !by) op.rot_kind = BIOIK_ROT_Y;
            else if (bx && by && !bz) op.rot_kind = BIOIK_ROT_Z;
        }
#endif
    }
    for (size_t k = 0; k < ops.size(); k++) {
        dev.ops[k] = ops[k];
        if (ops[k].gene >= 0) dev.active_mask |= 1ull << k;
        if ((int)k < n_chain && ops[k].mimic_src >= 0) dev.mimic_followers[ops[k].mimic_src] |= 1ull << k;
    }
}

DevSolveParams normalize_params(const bioik_solve_params& p, uint64_t first_query, size_t n_queries, size_t resident_units) {
    if (p.struct_size != sizeof(bioik_solve_params)) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_solve_params: struct_size mismatch");
    DevSolveParams o;
    std::memset(&o, 0, sizeof(o));
    auto thr = [](double v) { return (v < 0.0 || v >= FLT_MAX || !std::isfinite(v)) ? DBL_MAX : v; };  // problem.cpp:90-95
    o.dpos = thr(p.dpos), o.drot = thr(p.drot), o.dtwist = thr(p.dtwist);
    o.random_seed = p.random_seed;
    o.first_query = first_query;
    if (p.mode != BIOIK_MODE_BIO2 && p.mode != BIOIK_MODE_BIO2_MEMETIC && p.mode != BIOIK_MODE_BIO2_MEMETIC_L && p.mode != BIOIK_MODE_GD_C &&
        p.mode != BIOIK_MODE_JAC && p.mode != BIOIK_MODE_GD && p.mode != BIOIK_MODE_GD_R)
        throw Error(BIOIK_ERR_INVALID_ARGUMENT, "unknown solver mode");
    o.memetic = p.mode == BIOIK_MODE_BIO2 ? 0 : (p.mode == BIOIK_MODE_BIO2_MEMETIC_L ? 'l' : 'q');
    o.solver = p.mode == BIOIK_MODE_GD_C ? 1 : (p.mode == BIOIK_MODE_JAC ? 2 : (p.mode == BIOIK_MODE_GD ? 3 : (p.mode == BIOIK_MODE_GD_R ? 4 : 0)));
    if (p.fk_mode != BIOIK_FK_LINEAR && p.fk_mode != BIOIK_FK_EXACT) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "unknown fk_mode");
    o.fk_mode = p.fk_mode;
    o.lambda = p.population > 0 ? p.population : 16;  // reference: 16 children (ik_evolution_2.cpp:138)
    o.islands = p.islands > 0 ? p.islands : 1;
    // BIOIK_ISLANDS_AUTO: the islands the idle part of the chip carries (`resident_units` workgroups of the latency schedule's kernel are resident at once: 2048 on MI355X), at most
    // sixteen per query; they stop each other (profiles/r05_small_batches.log)
    const bool auto_islands = p.islands <= 0 && o.solver == 0 && n_queries > 0;
    if (auto_islands) {
        // (measured, profiles/r05_small_batches.log: sixteen at most; the resident workgroups shared out down to four per query -- calls of up to half the resident
        // workgroups still gain from four, 896 queries 6.1 -> 5.1 ms --; beyond that one)
        size_t isl = std::min<size_t>(16, resident_units / n_queries);
        if (n_queries <= resident_units / 2) isl = std::max<size_t>(isl, 4);
        // (round 6: a call of a few poses -- MoveIt's one pose per searchPositionIK -- leaves nearly all of the chip idle: 64 islands up to eight queries, 32 up to sixteen.
        // One pose per call, 64 against 16 islands: PoseGoal arm 0.80 -> 0.75 ms, the arm with a MinimalDisplacementGoal 3.9 -> 3.6 ms and 0.84 -> 0.95 of the poses
        // within a 5 ms timeout, the 31-joint chain 7.5 -> 6.8 ms: profiles/r06_one_pose_islands.log)
        isl = std::max<size_t>(isl, std::min<size_t>(64, resident_units / (4 * n_queries)));
        o.islands = (int32_t)std::max<size_t>(1, isl);
    }
    o.max_steps = p.max_steps > 0 ? p.max_steps : 0;
    if (p.timeout > 0.0 && std::isfinite(p.timeout)) {  // seconds -> ticks of the 100 MHz constant device clock, at least one
        const double ticks = p.timeout * 1e8;
        o.timeout_ticks = ticks >= 9e18 ? (uint64_t)9e18 : (ticks < 1.0 ? 1ull : (uint64_t)ticks);
    }
    o.no_wipeout = p.no_wipeout;
    if (p.schedule != BIOIK_SCHEDULE_LATENCY && p.schedule != BIOIK_SCHEDULE_THROUGHPUT && p.schedule != BIOIK_SCHEDULE_AUTO)
        throw Error(BIOIK_ERR_INVALID_ARGUMENT, "unknown schedule");
    o.schedule = p.schedule;
    o.generations = o.memetic ? 8 : 16;  // ik_evolution_2.cpp:349-351
    if (p.island_sync != 0 && p.island_sync != 1) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "island_sync must be 0 or 1");
    o.island_sync = ((p.island_sync || auto_islands) && o.islands > 1) ? 1 : 0;
    return o;
}

}  // namespace bioik

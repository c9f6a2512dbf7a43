// bioik_types.h — PODs shared by the host-side problem compiler and the gfx950 kernels.
//
// A problem template (robot model + goal structure, reference Problem::initialize, src/problem.cpp:72-228, plus
// RobotFK::initialize, src/forward_kinematics.h:253-330) is compiled ONCE on the host into a `DevProblem`: a flat
// "joint program" that every lane of every wavefront walks.  The block is read through uniform (scalar) loads, so
// its numbers reach the FP64 VALU as SGPR operands and cost neither VGPRs nor LDS bandwidth.
#pragma once
#include <stdint.h>

#define BIOIK_MAX_OPS 64    // moving joints on the union of the root->tip chains (+ off-chain goal variables)
#define BIOIK_MAX_TIPS 64  // (a BalanceGoal makes every link with mass a tip)
#define BIOIK_MAX_BALANCE 4
#define BIOIK_MAX_GOALS 24  // per class (primary / secondary)

enum { BIOIK_OP_NONE = 0, BIOIK_OP_REVOLUTE = 1, BIOIK_OP_PRISMATIC = 2, BIOIK_OP_FLOATING = 3, BIOIK_OP_PLANAR = 4 };
// Sparsity classes of a revolute op's constants, found once by the host-side problem compiler.  Most robot descriptions put a
// joint's origin on a coordinate axis of its parent (or at its parent's origin), leave it unrotated and turn it about x, y or z: then
// most terms of the general transform multiply an exact zero.  `x * 0 + y` is `y` bit for bit, so the chain walk drops those terms
// (a wavefront-uniform branch per joint) and returns the same numbers as the general form: 12 instead of 21 FP64 instructions for
// the position, 9 instead of 24 for the rotation, per joint and individual.
enum { BIOIK_POS_GENERAL = 0, BIOIK_POS_ZERO = 1, BIOIK_POS_X = 2, BIOIK_POS_Y = 3, BIOIK_POS_Z = 4 };
enum { BIOIK_ROT_GENERAL = 0, BIOIK_ROT_X = 1, BIOIK_ROT_Y = 2, BIOIK_ROT_Z = 3 };  // ca == (0,0,0,1), cb == (v e_i, 0)

// One moving joint of the link schedule (reference src/forward_kinematics.h:268-282).  The fixed links between the
// previous moving joint and this one are folded into the constant frame C on the host, so one op is
//     F_out = F_src o C o J(x)            (reference :331-354: global[l] = global[parent] o origin o joint_frame)
// with  revolute : C o J = ( C.pos , cos(x/2) * A + sin(x/2) * B ),  A = C.rot, B = C.rot (x) (axis, 0)
//       prismatic: C o J = ( C.pos + x * Pv , A ),                  Pv = C.rot * axis   (stored in cb[0..2])
struct DevOp {
    // What a joint of the chain walk reads comes first and lies together -- 14 doubles and 6 ints = 34 dwords, three scalar loads in one burst (fk_walk_n) --;
    // scattered over the record the same numbers took eight loads in three bursts, each waited for on its own: nothing for a full chip, a fifth of the
    // lone step of a call that cannot fill it (round 6)
    double cpos[3];
    double ca[4];
    double cb[4];
    double clip_min, clip_max, span; // RobotInfo (include/bio_ik/robot_info.h:70-106) of the variable
    int32_t type;                    // BIOIK_OP_*
    int32_t pos_kind;                // BIOIK_POS_*: which components of cpos are non-zero (revolute ops; the walk skips the exact no-ops)
    int32_t rot_kind;                // BIOIK_ROT_*: revolute op behind an unrotated constant frame turning about a coordinate axis
    int32_t gene;                    // index into Problem::active_variables, or -1 (inactive: value comes from the seed)
    int32_t tip_first, tip_count;    // device tips whose frame is F_out o E (evaluated right after this op)
    // ... and what the other phases read
    double axis[3];                  // joint axis in the joint frame (analytic Jacobian, forward_kinematics.h:639-693)
    double vmin, vmax;               // variable bounds (random re-initialisation, ik_evolution_2.cpp:626-631)
    double vw;                       // minimal_displacement_factors (src/problem.cpp:207-225)
    double mimic_factor, mimic_offset;  // mimic joint: value = x(mimic_src) * factor + offset (forward_kinematics.h:230-246)
    int32_t var;                     // robot variable index
    int32_t src;                     // op whose output frame is the parent frame, -1 = model root (identity)
    int32_t load_slot;               // >=0: parent frame must be fetched from this LDS slot (branching trees)
    int32_t save_slot;               // >=0: output frame is parked in this LDS slot for a later branch
    int32_t unbounded;               // clip_max == DBL_MAX (goal_types.h:394,419)
    int32_t mimic_src;               // op whose value this joint mimics, -1: the joint has its own value x(k)
    int32_t val_first;               // FLOATING / PLANAR chain op: first of its 7 / 3 consecutive value ops (type NONE), else -1
    int32_t joint_op;                // value op of a FLOATING / PLANAR joint: the chain op it belongs to, else -1
    int32_t multi_slot;              // FLOATING / PLANAR chain op: the LDS slot its joint frame J(values) is parked in before a walk (multi_joint_prologue); else -1
    int32_t pad_;
};

struct DevTip {
    double e[7];                       // constant trailing frame E (fixed links behind the last moving joint)
    int32_t src;                       // op index, -1 = root
    int32_t has_e;                     // 0: E is the identity
    int32_t out_index;                 // index in Problem::tip_link_indices (order of the public API)
    int32_t goal_first, goal_count;    // primary link goals reading this tip: DevProblem::primary[goal_first..)
    int32_t pose_off;                  // >= 0: exactly ONE primary goal reads this tip and it is a PoseGoal whose numbers start at this offset of the
    int32_t pad0;                      //       query's parameters -- the usual case, evaluated without a look at the goal table
    double pose_weight_sq;             //       its weight_sq
    uint64_t dep_mask;                 // bit k: op k lies on the root->tip chain (tip_dependencies, forward_kinematics.h:588-598)
    int32_t obj_type;                  // `jac` solver: what last wrote tipObjectives[tip] (ik_gradient.cpp:64-66): BIOIK_GOAL_POSITION /
    int32_t obj_param_off;             // ORIENTATION / POSE with its parameter offset, -1 = an identity frame
    double bal_w;                      // BalanceGoal: the link's share of the robot's mass (goal_types.cpp:246-254), 0 = none
    double bal_c[3];                   // BalanceGoal: the link's centre of mass in the link frame (urdf inertial origin)
};

struct DevGoal {
    double weight_sq;   // Problem::GoalInfo::weight_sq (problem.cpp:178)
    int32_t type;       // BIOIK_GOAL_*
    int32_t tip;        // device tip index, -1 when the goal reads no link
    int32_t var_op;     // op index of the goal's variable (JointVariableGoal), -1 none
    int32_t var_seed;   // robot variable index when the goal's joint is fixed (GoalContext::getVariablePosition, goal.h:70-77), -1 none
    int32_t param_off;  // offset of the goal's numbers inside the per-query parameter vector
    int32_t pad;
};

struct DevProblem {
    // the scalars one code site reads together sit in one 16-byte group, so that they arrive with one scalar load
    int32_t n_ops;        // ops[0..n_chain_ops) walk the kinematic tree; [n_chain_ops..n_ops) are off-chain goal variables
    int32_t D;            // Problem::active_variables.size()
    int32_t n_quat;       // floating joints whose orientation is four genes: renormalised after reproduction (ik_evolution_2.cpp:320-324)
    int32_t genes_follow_ops;  // 1: op_of_gene is increasing, so a walk over the ops meets the genes in gene order
    int32_t n_chain_ops;
    int32_t n_prefix;     // ops[0..n_prefix): a straight run of joints at the root that are not genes and carry no tip and no
                          // branch: their frame is the same for every individual of a query (the seed's), computed once
    int32_t multi_op;     // FIRST chain op that is a floating / planar joint (any depth, any number since round 5), -1 none: the joint frames J(values) of
                          // all such ops are computed in front of a walk and parked in their LDS slots (ops[k].multi_slot); inside the walk
                          // such an op is F_out = (F_src o C) o J with J fetched from the slot
    int32_t n_root_tips;  // tips[0..n_root_tips) hang off the model root without any moving joint
    int32_t T;            // Problem::tip_link_indices.size()
    int32_t V;            // robot variables
    int32_t P;            // doubles per query in goal_params
    int32_t n_slots;
    int32_t n_link_primary;  // primary[0..n_link_primary) are link goals grouped by tip; the rest read genes only
    int32_t n_primary;
    int32_t n_secondary;
    int32_t n_balance;     // BalanceGoals (goal_types.cpp:231-272): they read every tip with bal_w != 0; balance[0..n_balance)
    // The plugin's default problem -- ONE tip link carrying ONE PoseGoal and no other goal (kinematics_plugin.cpp:279-297) -- spelled out, so that the
    // latency-bound evaluations of a single individual (memetic phase, ranking) read three scalars instead of walking the goal tables:
    int32_t pose_only;       // 1: the primary fitness is primary[0].weight_sq * PoseGoal cost of tip 0, and there is no secondary / balance goal
    int32_t pose_param_off;  // primary[0].param_off
    double pose_weight_sq;   // primary[0].weight_sq
    int32_t serial_chain;    // 1: ops[0..n_chain_ops) are ONE serial chain -- every op continues the frame of the op in front of it, none fetches or parks
                             //    a branch frame, none is a mimic joint: the walk of the dense kernels then carries its frames in place
    int32_t reserved0;
    uint64_t active_mask;  // bit k: op k is an active gene (ops[k].gene >= 0)
    double multi_c[7];     // (unused since round 5: the constant frame in front of a floating / planar joint lies in its op's cpos / ca)
    int32_t quat_op[4];    // op index of the first of the four orientation value ops
    uint64_t mimic_followers[BIOIK_MAX_OPS];  // bit m: chain op m is a mimic joint following op k
    int32_t op_of_gene[BIOIK_MAX_OPS];
    int32_t tip_of_out[BIOIK_MAX_TIPS];  // device tip index of public tip i
    DevOp ops[BIOIK_MAX_OPS];
    DevTip tips[BIOIK_MAX_TIPS];
    DevGoal primary[BIOIK_MAX_GOALS];
    DevGoal secondary[BIOIK_MAX_GOALS];
    DevGoal balance[BIOIK_MAX_BALANCE];  // params: target[3] axis[3]
};

// Same block, read by the lean kernel flavour (bioik_platform.h: pb_flavour): the host hands it out only for problems without
// floating / planar joints (multi_op < 0, n_quat == 0) whose ops meet the genes in gene order (genes_follow_ops).
struct DevProblemLean : DevProblem {};

// solver parameters as the kernels see them (bioik_solve_params after normalisation, problem.cpp:90-95)
struct DevSolveParams {
    double dpos, drot, dtwist;  // DBL_MAX = disabled
    uint64_t random_seed;
    uint64_t first_query;       // global index of query 0 of this launch (multi-GPU shards keep their RNG streams)
    uint64_t timeout_ticks;     // wall-clock budget of the launch in ticks of the 100 MHz device clock, 0 = none (ik_parallel.h:160)
    int32_t memetic;            // 0, 'q', 'l'
    int32_t solver;             // 0: the bio2 family (k_solve), 1: gd_c, 2: jac, 3: gd, 4: gd_r (k_solve_point, src/ik_gradient.cpp)
    int32_t columnless;         // 1: children are computed where they are read (ChildX) instead of living in genotype columns in LDS
    int32_t fk_mode;            // BIOIK_FK_*
    int32_t lambda;             // children per species per generation
    int32_t islands;
    int32_t max_steps;
    int32_t no_wipeout;
    int32_t generations;        // 8 (memetic) or 16 (ik_evolution_2.cpp:349-351)
    int32_t child_cols;         // genotype columns per lane: ceil(lambda / lanes) = every child of a generation stays in LDS
                                // until selection; 1 = only the lane's current child (winners are re-derived from the RNG)
    int32_t species_parallel;   // 1: the workgroup splits into two lane groups, one species each, running concurrently
    int32_t schedule;           // BIOIK_SCHEDULE_* (host side only: what the launcher optimises for)
    int32_t child_pairs;        // 1: a lane reproduces and scores its children two at a time (two independent dependency chains per lane)
    int32_t island_sync;        // 1: the islands of a query stop once one of them has passed the success test (bioik_solve_params::island_sync)
};

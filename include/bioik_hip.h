/*
 * bioik_hip.h — C-ABI boundary of the MI355X-native bio2_memetic IK solver.
 *
 * This header is the drop-in boundary described in DESIGN.md §1.  It replaces, for the hot path
 * only, the pair of calls
 *
 *     ik->initialize(problem);   ik->solve();           (reference src/kinematics_plugin.cpp:566-578)
 *
 * i.e. `bio_ik::IKParallel::initialize/solve/getSolution/getSuccess/getSolutionFitness`
 * (reference src/ik_parallel.h:141-145, 193-276), and everything those calls execute:
 * `IKEvolution2<'q'>::initialize/step` (src/ik_evolution_2.cpp:111-230, 328-646), `RobotFK`
 * (src/forward_kinematics.h:65-1234), `Problem::computeGoalFitness/checkSolutionActiveVariables`
 * (src/problem.cpp:244-341) and the built-in `Goal::evaluate`s (include/bio_ik/goal_types.h).
 *
 * Only plain C types cross this boundary: no C++ classes, no torch types, no HIP types (streams and
 * device pointers travel as `void*`).  The reference-side adapter that marshals
 * `moveit::core::RobotModel` / `bio_ik::Goal` objects into these PODs is shown in INTEGRATION.md and
 * implemented (against stand-in MoveIt headers) in bio_ik_amd/cpp/.
 *
 * Conventions
 *   - a frame is 7 doubles: px py pz qx qy qz qw  (Hamilton quaternion, reference
 *     include/bio_ik/frame.h:51-56 stores the same 7 numbers + 1 pad)
 *   - every function returns BIOIK_OK (0) or a negative status; it never aborts or throws.
 *     `bioik_last_error()` returns a thread-local human-readable message for the last failure.
 *   - handles are opaque and must be destroyed by the matching *_destroy.
 */
#ifndef BIOIK_HIP_H
#define BIOIK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BIOIK_ABI_VERSION 6 /* 6 (round 6): bioik_resolve_islands (the island count BIOIK_ISLANDS_AUTO gives a call of n queries, for callers that shard a request themselves);
                               a rendezvous time-out inside a workgroup is BIOIK_ERR_HIP for its call; a line-search candidate with a joint value of magnitude >= 1e300 is no candidate.
                               5 (round 5): bioik_solve_params::islands = 0 means BIOIK_ISLANDS_AUTO (versions up to 4 took 0 as 1); `timeout` counts from the call */

/* ---- status codes ------------------------------------------------------------------------- */
enum {
    BIOIK_OK = 0,
    BIOIK_ERR_INVALID_ARGUMENT = -1, /* malformed descriptor (reference: ERROR(...) -> std::runtime_error, src/utils.h:122-129) */
    BIOIK_ERR_UNSUPPORTED = -2,      /* joint/goal type that has no device implementation */
    BIOIK_ERR_NO_DEVICE = -3,        /* no HIP device / extension not usable: the product path never falls back to CPU */
    BIOIK_ERR_HIP = -4,              /* a HIP runtime call failed; message in bioik_last_error() */
    BIOIK_ERR_NOT_FOUND = -5         /* link/variable not part of the group (reference src/problem.cpp:125,141) */
};

/* ---- joint types: moveit::core::JointModel::JointType as used by
 *      RobotJointEvaluator::getJointFrame (reference src/forward_kinematics.h:78-139) ---------- */
enum {
    BIOIK_JOINT_FIXED = 0,
    BIOIK_JOINT_REVOLUTE = 1,  /* also URDF "continuous" (unbounded revolute) */
    BIOIK_JOINT_PRISMATIC = 2,
    BIOIK_JOINT_FLOATING = 3,  /* 7 variables x y z qx qy qz qw; device: anywhere on the goal chains (general flavour of the solver kernel); refused: such
                                  a joint as a mimic of another, its variables off the goal chains, more than four with an active orientation              */
    BIOIK_JOINT_PLANAR = 4     /* 3 variables x y theta;          device: as FLOATING                                                     */
};

/* ---- goal opcodes: one per closed-form class of reference include/bio_ik/goal_types.h.
 *      `n_params` doubles per query, laid out as listed.  Vectors that the reference normalises
 *      in its setters/constructors (quaternions, directions) must arrive normalised. ------------ */
enum {
    BIOIK_GOAL_POSITION = 0,             /* goal_types.h:80-97    params: pos[3]                                  */
    BIOIK_GOAL_ORIENTATION = 1,          /* goal_types.h:99-124   params: quat[4]                                 */
    BIOIK_GOAL_POSE = 2,                 /* goal_types.h:126-181  params: pos[3] quat[4] rotation_scale           */
    BIOIK_GOAL_LOOK_AT = 3,              /* goal_types.h:183-212  params: axis[3] target[3]                       */
    BIOIK_GOAL_MAX_DISTANCE = 4,         /* goal_types.h:214-241  params: target[3] distance                      */
    BIOIK_GOAL_MIN_DISTANCE = 5,         /* goal_types.h:243-270  params: target[3] distance                      */
    BIOIK_GOAL_LINE = 6,                 /* goal_types.h:272-298  params: position[3] direction[3]                */
    BIOIK_GOAL_PLANE = 7,                /* goal_types.h:300-328  params: position[3] normal[3]                   */
    BIOIK_GOAL_AVOID_JOINT_LIMITS = 8,   /* goal_types.h:379-402  params: -                                       */
    BIOIK_GOAL_CENTER_JOINTS = 9,        /* goal_types.h:404-426  params: -                                       */
    BIOIK_GOAL_REGULARIZATION = 10,      /* goal_types.h:428-445  params: -                                       */
    BIOIK_GOAL_MINIMAL_DISPLACEMENT = 11,/* goal_types.h:447-466  params: -                                       */
    BIOIK_GOAL_JOINT_VARIABLE = 12,      /* goal_types.h:468-499  params: position                                */
    BIOIK_GOAL_SIDE = 13,                /* goal_types.h:585-614  params: axis[3] direction[3]                    */
    BIOIK_GOAL_DIRECTION = 14,           /* goal_types.h:616-644  params: axis[3] direction[3]                    */
    BIOIK_GOAL_CONE = 15,                /* goal_types.h:646-712  params: position[3] position_weight axis[3] direction[3] angle */
    BIOIK_GOAL_BALANCE = 16,             /* goal_types.h:540-566, goal_types.cpp:231-272  params: target[3] axis[3].  Reads the frames of
                                            EVERY link of the model with a positive mass (bioik_model_desc::link_mass), each of which
                                            becomes a tip of the problem, as BalanceGoal::describe does                              */
    BIOIK_GOAL_TYPE_COUNT = 17
    /* TouchGoal (FCL), JointFunctionGoal / LinkFunctionGoal (std::function) have no device opcode: DESIGN.md §7. */
};

/* number of per-query parameter doubles of a goal opcode, or -1 for an unknown opcode */
int bioik_goal_param_count(int goal_type);

/* ---- solver modes: IKFactory names of reference src/ik_evolution_2.cpp:652-654 -------------- */
enum {
    BIOIK_MODE_BIO2 = 0,            /* "bio2":           16 generations per step, no memetic phase      */
    BIOIK_MODE_BIO2_MEMETIC = 1,    /* "bio2_memetic":   8 generations + quadratic ('q') line search    */
    BIOIK_MODE_BIO2_MEMETIC_L = 2,  /* "bio2_memetic_l": 8 generations + linear ('l') line search       */
    /* IKFactory names of reference src/ik_gradient.cpp:254-292 (population / fk_mode are not used).  Island 0 of a query starts at the seed;
       with islands = N > 1 the islands 1 ... N - 1 start at random configurations, as the solver threads 1 ... N - 1 of the reference's
       "gd_2" ... "gd_8", "gd_r_2" ..., "gd_c_2" ..., "jac_2" ... "jac_8" do (:157-159, :283-285), and the best island is returned        */
    BIOIK_MODE_GD_C = 3,            /* "gd_c": gradient descent by central differences of the exact fitness, linear step estimate,
                                       always continue (ik_gradient.cpp:136-251, if_stuck = 'c')                                   */
    BIOIK_MODE_GD = 5,              /* "gd":   as gd_c, but a step is kept only if it lowers the fitness (src/ik_gradient.cpp:225-232) */
    BIOIK_MODE_JAC = 4,             /* "jac":  pseudo-inverse-Jacobian steps on the twists between the tips and their pose goals
                                       (ik_gradient.cpp:42-133, 269-292)                                                          */
    BIOIK_MODE_GD_R = 6             /* "gd_r": as gd, but a configuration that a step failed to improve is replaced by a random one
                                       before the next step (:165-170, :233-237); the best configuration seen is returned          */
};

/* What the launcher optimises a solve for (the measured figures of the current round: DESIGN.md section 6).
 * LATENCY (default): the time of THIS call.  A query gets the lanes that make its steps short (128 and two helper wavefronts: ~90 us per step of the 7-joint
 *   arm), and a batch beyond what the chip holds of such workgroups (3072 queries and more) starts under the denser mapping below -- every query resident from
 *   the first moment -- and hands its stragglers to the short-step mapping when the chip runs empty.  Three such solves in flight keep an MI355X busy.
 * THROUGHPUT: solves per second of a STREAM of batches.  Both species of a query share one wavefront and the children are computed where they
 *   are read, sixteen queries per CU: a third more steps per ms on a full chip, but a step takes 2.5 x as long, so the stragglers of a batch run for
 *   up to 12 ms.  It pays with six to ten batches in flight on as many streams -- and hardware queues: the HIP runtime maps streams onto four unless
 *   GPU_MAX_HW_QUEUES says otherwise -- and costs an isolated call about a fifth more time.  Problems the denser mapping does not exist for (secondary
 *   goals with more than 256 children, 32 or more genes, floating joints) run as under LATENCY.
 * AUTO: LATENCY, except in bioik_solve_batch_submit when two or more solves of the handle are already in flight: THROUGHPUT then (a caller
 *   that streams batches through the asynchronous entry gets the dense mapping once its pipeline is three deep; an isolated call stays as fast as it
 *   can be). */
enum { BIOIK_SCHEDULE_LATENCY = 0, BIOIK_SCHEDULE_THROUGHPUT = 1, BIOIK_SCHEDULE_AUTO = 2 };
/* bioik_solve_params::islands = BIOIK_ISLANDS_AUTO (0): as many islands per query as the part of the chip the call leaves idle can carry --
 * for a call of n queries (per device) max(min(16, 2048 / n), min(64, 512 / n)), at least four up to n = 1024, one beyond that: 64 for up to eight queries
 * (MoveIt's one pose per call), 32 for sixteen, 16 up to 128 ... -- stopping each other (island_sync is then taken as 1): the reference's four island
 * threads with "any thread succeeds => all stop" (ik_parallel.h:102, 141-178), sized to the hardware.  A call that cannot fill the chip is bound by its
 * slowest query's number of steps, and islands cut exactly that (MI355X, PoseGoal on a 7-joint arm, pop 128: 16 queries 3.5 -> 1.25 ms per call, 256 queries
 * 6.2 -> 2.8 ms, one query 1.2 -> 0.71 ms: bench.py's small_batches).  bioik_resolve_islands() returns the count a call of n queries would get.
 * THE ANSWER OF A QUERY THEN DEPENDS ON THE SIZE OF THE CALL IT CAME IN (the island count is a function of n, and the islands' streams of random numbers are
 * part of the answer): give an explicit count where answers must not depend on batching.  The bio2 family only: for gd / jac an island count names another
 * solver ("gd_8"). */
enum { BIOIK_ISLANDS_AUTO = 0 };

/* how a child's phenotype (tip frames) is obtained inside the evolution loop */
enum {
    BIOIK_FK_LINEAR = 0, /* the reference's first-order extrapolation RobotFK_Mutator::computeApproximateMutations
                            (src/forward_kinematics.h:1172-1233) around the species' elite                       */
    BIOIK_FK_EXACT = 1   /* exact chain walk per individual, RobotFK_Fast_Base::applyConfiguration semantics
                            (src/forward_kinematics.h:331-354) — the north-star GPU path                         */
};

/* ---- flattened robot model = what RobotJointEvaluator/RobotFK_Fast_Base/RobotInfo read from
 *      moveit::core::RobotModel (reference src/forward_kinematics.h:192-213, 230-246, 268-329;
 *      include/bio_ik/robot_info.h:70-106).  One entry per link; entry i also describes the link's
 *      parent joint (MoveIt: every link has exactly one parent joint). --------------------------- */
typedef struct bioik_model_desc {
    uint32_t struct_size;               /* sizeof(bioik_model_desc), for ABI evolution */
    uint32_t n_links;
    uint32_t n_variables;
    uint32_t reserved;
    const int32_t* link_parent;          /* [n_links] parent link index, -1 for the root link; parent < child */
    const double* link_origin;           /* [n_links*7] LinkModel::getJointOriginTransform()            */
    const int32_t* joint_type;           /* [n_links] BIOIK_JOINT_*                                      */
    const double* joint_axis;            /* [n_links*3] Revolute/PrismaticJointModel::getAxis()          */
    const int32_t* joint_first_variable; /* [n_links] JointModel::getFirstVariableIndex(), -1 if none   */
    const int32_t* joint_mimic;          /* [n_links] link index whose joint this joint mimics, -1 (a mimic of a mimic is resolved to the joint at the end of the chain with the composed factor and offset, as MoveIt's RobotModel::buildMimic does) */
    const double* joint_mimic_factor;    /* [n_links] getMimicFactor()                                   */
    const double* joint_mimic_offset;    /* [n_links] getMimicOffset()                                   */
    const double* var_min;               /* [n_variables] VariableBounds::min_position_                  */
    const double* var_max;               /* [n_variables] VariableBounds::max_position_                  */
    const uint8_t* var_bounded;          /* [n_variables] VariableBounds::position_bounded_              */
    const double* var_max_velocity;      /* [n_variables] VariableBounds::max_velocity_                  */
    const double* link_mass;             /* [n_links] urdf::Link::inertial->mass, 0 where the link has no <inertial>; NULL: no link
                                            has one (only BIOIK_GOAL_BALANCE reads it, goal_types.cpp:236-247)                  */
    const double* link_center;           /* [n_links*3] urdf::Link::inertial->origin.position (link frame); NULL with link_mass   */
} bioik_model_desc;

/* ---- one goal of the problem template (structure shared by every query of a batch; the numeric
 *      parameters are per query).  Mirrors what Goal::describe() puts into GoalContext
 *      (reference include/bio_ik/goal.h:87-90, 113-117) plus the goal's class. ------------------ */
typedef struct bioik_goal_desc {
    int32_t type;      /* BIOIK_GOAL_*                                                     */
    int32_t link;      /* link index for LinkGoalBase-derived goals, -1 otherwise          */
    int32_t variable;  /* variable index for BIOIK_GOAL_JOINT_VARIABLE, -1 otherwise       */
    int32_t secondary; /* Goal::isSecondary()                                              */
    double weight;     /* Goal::getWeight()                                                */
} bioik_goal_desc;

/* ---- problem template = the arguments of Problem::initialize (reference src/problem.cpp:72-228)
 *      that do not change from query to query. -------------------------------------------------- */
typedef struct bioik_problem_desc {
    uint32_t struct_size;
    uint32_t n_group_joints;
    const int32_t* group_joints; /* link indices of JointModelGroup::getActiveJointModels(), in that order */
    uint32_t n_goals;
    uint32_t n_fixed_joints;
    const bioik_goal_desc* goals;
    const int32_t* fixed_joints; /* link indices of BioIKKinematicsQueryOptions::fixed_joints (goal.h:124) */
} bioik_problem_desc;

/* ---- solver parameters = IKParams (reference src/utils.h:64-85) restricted to bio2*, plus the
 *      additive GPU keys of DESIGN.md §5. -------------------------------------------------------- */
typedef struct bioik_solve_params {
    uint32_t struct_size;
    int32_t mode;            /* BIOIK_MODE_*; yaml "mode" (kinematics_plugin.cpp:252)                    */
    int32_t fk_mode;         /* BIOIK_FK_*;  new key "gpu_fk"                                            */
    int32_t population;      /* children per species per generation (reference hard-codes 16,
                                ik_evolution_2.cpp:138); new key "gpu_population"                        */
    int32_t islands;         /* independent islands per query (reference concurrency()==4 identical
                                clones, ik_evolution_2.cpp:649 + utils.h:423); new key "gpu_islands".  Applies to
                                every mode: for gd / gd_r / gd_c / jac, islands = N > 1 is the reference's "gd_N" ...
                                (islands 1 ... N - 1 start at random configurations).  0 = BIOIK_ISLANDS_AUTO (above);
                                bioik_default_solve_params sets 1                                            */
    int32_t max_steps;       /* budget in IKEvolution2::step() calls per island (deterministic; checked after
                                every step like the success test of ik_parallel.h:173-181)               */
    uint64_t random_seed;    /* yaml "random_seed" (kinematics_plugin.cpp:256)                           */
    double dpos, drot, dtwist; /* yaml keys (kinematics_plugin.cpp:259-261); <0 or >=FLT_MAX disables   */
    int32_t no_wipeout;      /* debugging aid: disable species wipe-outs                                 */
    int32_t schedule;        /* BIOIK_SCHEDULE_*: what the launcher optimises for; new key "gpu_schedule".  The results do not
                                depend on it (the lane mapping never changes a trajectory).                              */
    double timeout;          /* the caller's `timeout` of searchPositionIK [s] (kinematics_plugin.cpp:504, 574;
                                ik_parallel.h:160, 200 `ros::WallTime::now() < timeout`): wall-clock budget of ONE
                                bioik_solve_batch* call, counted FROM THE CALL (round 5): the library keeps the device's
                                100 MHz clock paired with the host's steady clock and hands the kernels the call's deadline in
                                device ticks, so a solve that waits behind other solves of the caller's pipeline (several
                                submits in flight, several device-pointer solves on busy streams) is not granted its wait on
                                top.  Every query runs at least one step (ik_parallel.h:160 `iteration != 0`), then stops at
                                the first of: success, max_steps, timeout.  A call CAPTURED into a hipGraph cannot know when it
                                will be replayed: its budget counts from the moment the replay's first workgroup starts (at most
                                64 captured calls with a timeout per handle).  <= 0: no wall-clock limit (results are then
                                independent of timing).                                                                    */
    int32_t island_sync;     /* islands > 1: 0 = every island runs to its own success or budget and the best one is returned (ik_parallel.h:220-269
                                over independent islands); 1 = "any island succeeds => all stop" (the reference's `finished` flag, ik_parallel.h:102,
                                160-178) in its deterministic form: the islands of a query advance step by step together and all stop at the end of the
                                FIRST step in which any of them passes the success test; the result is the best of the islands that passed in that
                                step, `steps` is that step's number.  (On the device an island leaves as soon as it sees that another one has passed at
                                an earlier or the same step -- its own result can no longer be chosen -- so WHEN it leaves depends on timing, WHAT is
                                returned does not.)  New key "gpu_island_sync". */
    int32_t reserved0;
} bioik_solve_params;

/* defaults: bio2_memetic, exact FK, population 128, 1 island, 64 steps, no timeout, seed 0, dpos=drot=off, dtwist=1e-5 */
void bioik_default_solve_params(bioik_solve_params* p);

typedef struct bioik_model bioik_model;
typedef struct bioik_problem bioik_problem;

const char* bioik_last_error(void);
int bioik_abi_version(void);
/* number of usable HIP devices (0 when none); never fails */
int bioik_device_count(void);
/* Diagnostics (no reference counterpart): the library reads its BIOIK_SOLVE_* / BIOIK_PHASE_DUMP switches (tools/README.md: forced lane mappings,
 * hand-overs, the mapping report) from the environment ONCE, when it is loaded; this call reads them again.  The test-suite uses it to run every
 * lane mapping in one process; nothing on the solve path touches the environment. */
int bioik_debug_reload_switches(void);
/* The launcher's lane mappings are chosen by rules fitted to the robots of BASELINE.json -- and, since round 5, checked by measurement: the FIRST
 * chip-filling call (2048 (query, island) units and more) of a kind -- (population, fk_mode, islands or not) under the latency schedule, no timeout -- that a
 * handle sees through a host-pointer entry (bioik_solve_batch, bioik_solve_batch_submit, bioik_solve_batch_multi) is run once per eligible mapping (each
 * run IS the caller's solve: every mapping returns the same bits), timed with events, and the fastest is kept for the handle; bioik_solve_batch_device uses
 * a handle's choice but never waits for its stream to make one (BIOIK_SOLVE_AUTOTUNE=2: it does; =0: the rules alone).  That first call takes about five
 * times as long as the ones after it.  BIOIK_SOLVE_REPORT=1 prints the table (profiles/r05_measured_mapping_choice.log). */

/* replaces RobotFK/RobotInfo construction from a RobotModel (ik_base.h:144-151) */
int bioik_model_create(const bioik_model_desc* desc, int device, bioik_model** out);
void bioik_model_destroy(bioik_model* m);

/* replaces Problem::initialize + IKBase::initialize(problem) -> RobotFK::initialize(tips)
 * (problem.cpp:72-228, ik_base.h:154-161, forward_kinematics.h:253-330, 566-599).
 * Size limits of one problem (BIOIK_ERR_UNSUPPORTED beyond them): 64 moving joints on the union of the goal chains (a short
 * chain in front of a branch counts once per branch), 63 active variables, 64 tip links, 24 primary + 24 secondary goals. */
int bioik_problem_create(bioik_model* model, const bioik_problem_desc* desc, bioik_problem** out);
void bioik_problem_destroy(bioik_problem* p);

/* introspection of what Problem::initialize derived */
int bioik_problem_active_variable_count(const bioik_problem* p);          /* Problem::active_variables.size()  */
int bioik_problem_active_variables(const bioik_problem* p, int32_t* out); /* robot variable index per gene      */
int bioik_problem_tip_count(const bioik_problem* p);                      /* Problem::tip_link_indices.size()   */
int bioik_problem_tip_links(const bioik_problem* p, int32_t* out);
int bioik_problem_param_count(const bioik_problem* p);                    /* doubles per query in goal_params   */
int bioik_problem_variable_count(const bioik_problem* p);                 /* robot variables V                  */
/* Global index of query 0 of the following bioik_solve_batch* calls (default 0).  The random stream of a query is
 * keyed by (random_seed, global query index, island), so a batch sharded over several GPUs / calls reproduces the
 * unsharded run when every shard announces its offset.  (New: the reference has one query per call.) */
int bioik_problem_set_first_query(bioik_problem* p, uint64_t first_query);
/* What bioik_solve_params::islands = BIOIK_ISLANDS_AUTO means for a call of n queries on this handle's device: the island count and island_sync the call would
 * run with (an explicit count comes back as it is).  For callers that cut ONE request into several calls -- the MoveIt plugin's shards over `gpu_devices` --
 * and want every part to run the same solve whatever its size: resolve once for the largest part (ceil(rows / W), as bioik_solve_batch_multi does) and pass the
 * explicit figures to every call.  (The reference has no counterpart: its island count is the host's thread count, src/ik_parallel.h:84, utils.h:423.) */
int bioik_resolve_islands(const bioik_problem* p, const bioik_solve_params* params, size_t n, int32_t* islands, int32_t* island_sync);

/*
 * The batched solve: replaces IKParallel::solve() for n independent queries
 * (reference src/ik_parallel.h:193-270, per query).
 *   seeds        [n][V]  Problem::initial_guess (kinematics_plugin.cpp:466-485, 507): full variable vectors
 *   goal_params  [n][P]  per-query goal parameters, goals in template order (P = bioik_problem_param_count)
 *   solutions    [n][V]  IKParallel::getSolution(): full variable vector (inactive variables = seed)
 *   fitness      [n]     IKParallel::getSolutionFitness(): primary fitness of the exact-FK pose
 *   success      [n]     IKParallel::getSuccess(): Problem::checkSolutionActiveVariables on the exact-FK pose
 *   steps        [n]     number of step() calls executed by the winning island (new: device counter)
 * Host-pointer variant: copies in, solves, copies out, synchronises.
 */
int bioik_solve_batch(bioik_problem* p, const bioik_solve_params* params, size_t n, const double* seeds,
                      const double* goal_params, double* solutions, double* fitness, int32_t* success,
                      int32_t* steps);

/* The same solve without waiting for it — for a caller with a STREAM of batches, the batched counterpart of calling
 * IKParallel::solve() from several threads (reference src/ik_parallel.h:193-218 keeps its own worker threads busy the same way).
 * `submit` copies the inputs into a page-locked arena of the handle, enqueues transfer in / solve / transfer out on one of the
 * handle's SIX internal streams (tickets rotate over them) and returns a ticket; `wait` blocks until that solve is complete and
 * its results are in the arrays given to `submit`, which must stay valid until then.  Up to six solves of a handle are in flight
 * together (three fill the chip under BIOIK_SCHEDULE_LATENCY, six under BIOIK_SCHEDULE_THROUGHPUT): the slow tail of one (a few queries that use the whole step budget) runs behind the bulk of the next, which is worth
 * about a factor of two in solves per second on 4096-query batches.  Submitting a seventh solve first
 * completes the oldest one (its results are delivered; its `wait` then returns at once).  Tickets may be waited for in any
 * order, from any thread; a ticket that is never waited for is completed by a later submit or by bioik_problem_destroy. */
int bioik_solve_batch_submit(bioik_problem* p, const bioik_solve_params* params, size_t n, const double* seeds,
                             const double* goal_params, double* solutions, double* fitness, int32_t* success,
                             int32_t* steps, uint64_t* ticket);
int bioik_solve_batch_wait(bioik_problem* p, uint64_t ticket);

/* The same batch over several problem handles of ONE template — one handle per GPU of the node (the "device list" of the batched
 * searchPositionIK()), or several handles on one device.  Shard r = queries [r n / W, (r + 1) n / W) goes
 * to problems[r] on its own host thread and stream; there is no exchange between shards, and because the random stream of a query is
 * keyed by its global index (the offset set on problems[0] + its position in this batch) the result equals the unsharded solve bit
 * for bit.  Host pointers; returns when every shard is complete; the first failing shard's status is returned. */
int bioik_solve_batch_multi(bioik_problem* const* problems, int n_problems, const bioik_solve_params* params, size_t n,
                            const double* seeds, const double* goal_params, double* solutions, double* fitness,
                            int32_t* success, int32_t* steps);

/* Device-pointer variant: all arrays already resident in HBM on the problem's device; enqueues on
 * `hip_stream` (a hipStream_t passed as void*, NULL = default stream) and returns without synchronising.
 * Solves of ONE problem handle on different streams may be in flight together (their result arrays must differ; the scratch
 * of a solve is one persistent buffer per (handle, stream), taken from the stream-ordered pool by the first call on that stream and
 * released with the handle): keeping two or three batches in flight hides the slow tail of each
 * solve behind the bulk of the next.  A call enqueues one kernel or a short chain of kernels (large
 * batches pass their stragglers from a first launch to a second through device memory; islands > 1 add a selection
 * kernel); nothing in it waits for the host, so it may be captured into a hipGraph and replayed any number of times, on new contents of the
 * captured arrays too: make one eager call of the same size on the stream first (it sizes the scratch).  The words a solve resets per
 * call (hand-over counters, the timeout's clock, the islands' first-success words) are written by a kernel of this library, NOT by
 * hipMemsetAsync: on ROCm 7.2 a graph with a memset node in front of these kernels faults on its second replay (HISTORY.md).
 * Lifetime of the scratch under capture: a buffer a captured call has used is PINNED -- the graph holds its address -- and stays alive until
 * bioik_problem_destroy; the next eager call on that stream moves on to a buffer of its own, whatever its size.  Graphs captured from ONE
 * stream of one handle share that stream's pinned buffer: launch them on one stream (or capture from different streams).  Destroy the
 * graphs before the handle. */
int bioik_solve_batch_device(bioik_problem* p, const bioik_solve_params* params, size_t n, const double* d_seeds,
                             const double* d_goal_params, double* d_solutions, double* d_fitness,
                             int32_t* d_success, int32_t* d_steps, void* hip_stream);

/* ---- function-level entry points (device-resident math exposed one function at a time so that
 *      each kernel can be parity-checked against the matching reference function) -------------- */

/* RobotFK_Fast_Base::applyConfiguration + getTipFrames (forward_kinematics.h:331-356) for n genotypes.
 *   seed [V] supplies the inactive variables; genes [n][D]; tip_frames [n][T][7].  Host pointers. */
int bioik_eval_fk(bioik_problem* p, size_t n, const double* seed, const double* genes, double* tip_frames);

/* Problem::computeGoalFitness over goals / secondary_goals (problem.cpp:244-257) with the phenotype from
 * exact FK (fk_mode = BIOIK_FK_EXACT) or from the linear extrapolation around `base_genes`
 * (fk_mode = BIOIK_FK_LINEAR; base_genes [D]).  One query's seed [V] and goal_params [P]; genes [n][D];
 * primary [n], secondary [n].  Host pointers. */
int bioik_eval_fitness(bioik_problem* p, int fk_mode, size_t n, const double* seed, const double* goal_params,
                       const double* base_genes, const double* genes, double* primary, double* secondary);

/* RobotFK_Mutator::initializeMutationApproximator (forward_kinematics.h:802-930): the per-(tip,gene) delta
 * frames around base_genes.  deltas [T][D][7] (dpx dpy dpz dqx dqy dqz dqw), tip_frames [T][7]. Host pointers. */
int bioik_eval_approximator(bioik_problem* p, const double* seed, const double* base_genes, double* tip_frames,
                            double* deltas);

/* IKEvolution2::reproduce (ik_evolution_2.cpp:242-326) with the counter-based RNG of DESIGN.md §4:
 * parents [2][2][D] (parent, {genes,gradients}, gene); out children_genes / children_gradients [population][D].
 * rng_key/species/generation select the random stream. Host pointers. */
int bioik_eval_reproduce(bioik_problem* p, int population, uint32_t rng_key, int species, uint32_t generation,
                         const double* parents, double* children_genes, double* children_gradients);

/* Problem::checkSolutionActiveVariables (problem.cpp:259-341) on the exact-FK pose of n genotypes. */
int bioik_eval_check(bioik_problem* p, const bioik_solve_params* params, size_t n, const double* seed,
                     const double* goal_params, const double* genes, int32_t* ok);

/* The shared arithmetic of both sides of the boundary, one function at a time on the device (bio_ik_amd/csrc/bioik_sincos.h, bioik_fused.h, bioik_acos.h -- the
 * headers the kernels and the test-suite's CPU checker both include): so that a test can hold them against an INDEPENDENT high-precision
 * reference (tests/test_arith_headers.py).  op / doubles in / doubles out per element:
 *   0 sincos        x                                      -> sin x, cos x           (reference: libm, src/forward_kinematics.h:89-112)
 *   1 qrot          q[4] v[3]                              -> q v q^-1                (frame.h:108-149)
 *   2 qmul          p[4] q[4]                              -> p (x) q                 (frame.h:151-172)
 *   3 dot3          a[3] b[3]   4 dot4  a[4] b[4]          -> a . b
 *   5 revolute      frame[7] half_angle cpos[3] ca[4] cb[4] pos_kind rot_kind pad  -> frame'[7] by the general form, frame'[7] by the sparse form
 *   6 acos          x                                      -> acos x                  (reference: tf2Acos -> libm, include/bio_ik/goal_types.h:183-212, 646-712)
 *   7 atan2         y x                                    -> atan2(y, x)             (reference: KDL::Rotation::GetRot -> libm, src/problem.cpp:281-321)
 * Host pointers. */
int bioik_eval_arith(int device, int op, size_t n, const double* in, double* out);

/* streamed (unfused) generation: n_units (query,species) populations resident in HBM, layout
 * genes [n_units][D][population] (individual index fastest — coalesced), fitness [n_units][population].
 * One launch evaluates exact-FK fitness of every individual.  Used for the HBM-streamed measurement of
 * DESIGN.md §6; device pointers, enqueued on hip_stream. */
int bioik_stream_fitness_device(bioik_problem* p, size_t n_units, int population, const double* d_seeds,
                                const double* d_goal_params, const double* d_genes, double* d_fitness,
                                void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* BIOIK_HIP_H */

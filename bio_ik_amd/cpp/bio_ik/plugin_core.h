// bio_ik/plugin_core.h — the ONE implementation of what `BioIKKinematicsPlugin::searchPositionIK` does around the solver call
// (reference src/kinematics_plugin.cpp:437-655), shared by every face of the plugin: the MoveIt translation unit
// (src/kinematics_plugin_hip.cpp, `kinematics::KinematicsBase`), the header-only class for programs without MoveIt
// (bio_ik/kinematics_plugin.h) and, through the C shim of src/plugin_shim.cpp, the Python package.
//
//   seed state over the context / default state (:465-485)            -> Engine::submit
//   default goals first, then the caller's; `replace` (:550-556)      -> the caller builds the goal list, Engine compiles / caches it
//   tip poses into the model frame, per-query goal numbers (:487-546)  -> Engine::submit
//   problem.initialize / ik->initialize / ik->solve (:560-578)         -> bioik_solve_batch_submit (include/bioik_hip.h), n queries at once
//   angle wrap towards the seed, at the limits, clamp (:580-613)       -> Engine::wait
//   enforcePositionBounds, result -> group variables (:616-629)        -> Engine::wait
//   solution_fitness, approximate solutions (:632-641)                 -> Engine::wait
//
// A batch is SUBMITTED (inputs marshalled, transfers and kernels enqueued, nothing waited for) and later WAITED for: a caller with a
// stream of batches keeps up to six in flight per device and reaches the throughput of DESIGN.md section 6 through this boundary;
// the synchronous calls are submit + wait.  Robot-model access goes through `ModelView`, a handful of arrays and name look-ups that
// each face fills from its own model type.  Header-only; link with libbioik_hip.so.
#pragma once
#include <cfloat>
#include <chrono>
#include <cmath>
#include <functional>
#include <list>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

#include "goal.h"
#include "goal_eval.h"
#include "goal_types.h"

namespace bio_ik {
namespace core {

struct ModelView {
    size_t n_variables = 0;
    std::vector<uint8_t> var_revolute;  // the variable of a revolute joint (what the angle wrap applies to, :584-586)
    std::vector<uint8_t> var_bounded;   // VariableBounds::position_bounded_
    std::vector<double> var_min, var_max;
    bool has_mimic = false;             // the reference skips the angle wrap for models with mimic joints (:585)
    std::vector<int> group_vars;        // robot variable behind every entry of an ik_seed_state / solution vector (:473-484, :619-629)
    std::vector<int32_t> group_joints;  // link index (= its parent joint) per JointModelGroup::getActiveJointModels()
    std::function<int(const std::string&)> link_index, variable_index, joint_link_index;  // by name; negative: unknown
    std::function<void(double*)> enforce_bounds;  // RobotModel::enforcePositionBounds where the face has MoveIt's; else the generic form below
    // For goals WITHOUT a device implementation (JointFunctionGoal, LinkFunctionGoal, a user's Goal subclass: evaluated on the host, see Engine::wait):
    std::vector<double> var_max_velocity;  // VariableBounds::max_velocity_ (RobotInfo, include/bio_ik/robot_info.h:70-106)
    std::vector<uint8_t> var_prismatic;
    std::function<void(const std::string& link, const double* positions, double* frame7)> link_frame;  // global frame of a link at a full variable vector
};

struct Settings {  // IKParams (src/utils.h:64-85) as far as the device path reads them, + the additive gpu_* keys
    std::string mode = "bio2_memetic";
    int random_seed = 0;
    double dpos = DBL_MAX, drot = DBL_MAX, dtwist = 1e-5;
    bool no_wipeout = false;
    int gpu_population = 128, gpu_islands = 0, gpu_max_steps = 4096;  // gpu_islands: 0 = BIOIK_ISLANDS_AUTO (bioik_hip.h): the reference's island threads, as many as the
                                                                    // part of the chip a call leaves idle carries -- MoveIt's one pose per call gets sixteen,
                                                                    // a batch of 2048 and more one; N > 0: exactly N
    bool gpu_island_sync = true;  // gpu_islands > 1 (and the _2 / _4 / _8 solver names): "any island succeeds => all stop", the reference's island loop
                                  // (ik_parallel.h:102, 160-178) in its deterministic form (bioik_solve_params::island_sync); false: every island to its own end
    std::string gpu_fk = "exact";
    std::string gpu_schedule = "auto";     // "latency": every call as fast as it can be; "throughput": for callers that keep six or more batches in
                                           // flight (searchPositionIKBatchAsync): +30 % solves per second, an isolated call a quarter slower; "auto":
                                           // throughput for a batch submitted while two or more are in flight, else latency (bioik_hip.h)
    bool gpu_reproducible_calls = false;  // true: every call draws from the same random streams (query k of a call = stream k), so a repeated
                                          // call returns the same answer; false (default): the streams advance from call to call like the
                                          // reference's generator state, and a retry of a failed query explores differently
    std::vector<int> devices = {0};  // a batch is sharded over them (contiguous shards, no exchange)
    int gpu_host_goal_candidates = 4;  // solves whose goal list holds goals without a device implementation: candidates per query (independent random
                                       // streams of the device search over the device-capable goals) that the host then scores with ALL goals
};

// IKFactory names with a device implementation (src/ik_evolution_2.cpp:652-654, src/ik_gradient.cpp:254-292): the bio2 family, and gd /
// gd_r / gd_c / jac with their `_2 / _4 / _8` forms -- N solver threads of which the threads 1 ... N - 1 start at random configurations
// (ik_parallel.h:141-145, ik_gradient.cpp:157-159) are N islands of a query on the device.  Not available: bio1, the cppoptlib and FANN
// families -- an unknown name is a configuration error, as in IKFactory::create.
struct SolverMode {
    int mode;     // BIOIK_MODE_*
    int threads;  // IKSolver::concurrency() of the name where the device models it as islands (the gradient / Jacobian solvers), else 0
};
inline SolverMode solverMode(const std::string& name) {
    if (name == "bio2") return {BIOIK_MODE_BIO2, 0};
    if (name == "bio2_memetic") return {BIOIK_MODE_BIO2_MEMETIC, 0};
    if (name == "bio2_memetic_l") return {BIOIK_MODE_BIO2_MEMETIC_L, 0};
    static const struct { const char* stem; int mode; } stems[] = {{"gd_r", BIOIK_MODE_GD_R}, {"gd_c", BIOIK_MODE_GD_C}, {"gd", BIOIK_MODE_GD}, {"jac", BIOIK_MODE_JAC}};
    for (auto& st : stems) {
        const std::string stem = st.stem;
        if (name == stem) return {st.mode, 1};
        for (int n : {2, 4, 8})
            if (name == stem + "_" + std::to_string(n)) return {st.mode, n};
    }
    throw std::runtime_error("bio_ik (MI355X): solver mode '" + name + "' has no device implementation (available: bio2, bio2_memetic, bio2_memetic_l, "
                             "gd, gd_r, gd_c, jac and their _2 / _4 / _8 forms)");
}

inline void concat7(const double* a, const double* b, double* r) {  // a o b for frames px py pz qx qy qz qw (include/bio_ik/frame.h:174-187)
    const double *q = a + 3, *v = b;
    const double tx = 2 * (q[1] * v[2] - q[2] * v[1]), ty = 2 * (q[2] * v[0] - q[0] * v[2]), tz = 2 * (q[0] * v[1] - q[1] * v[0]);
    double o[7];
    o[0] = a[0] + v[0] + q[3] * tx + q[1] * tz - q[2] * ty;
    o[1] = a[1] + v[1] + q[3] * ty + q[2] * tx - q[0] * tz;
    o[2] = a[2] + v[2] + q[3] * tz + q[0] * ty - q[1] * tx;
    const double* p = b + 3;
    o[3] = q[3] * p[0] + q[0] * p[3] + q[1] * p[2] - q[2] * p[1];
    o[4] = q[3] * p[1] - q[0] * p[2] + q[1] * p[3] + q[2] * p[0];
    o[5] = q[3] * p[2] + q[0] * p[1] - q[1] * p[0] + q[2] * p[3];
    o[6] = q[3] * p[3] - q[0] * p[0] - q[1] * p[1] - q[2] * p[2];
    for (int i = 0; i < 7; i++) r[i] = o[i];
}

// The goals a plugin solves for when the caller brings none of its own (kinematics_plugin.cpp:279-329): a PoseGoal per tip frame, then
// the optional regularisers whose weight is positive.
inline void makeDefaultGoals(const std::vector<std::string>& tip_frames, double rotation_scale, bool position_only_ik, double center_joints_weight,
                             double avoid_joint_limits_weight, double minimal_displacement_weight, std::vector<std::unique_ptr<Goal>>& out) {
    out.clear();
    for (auto& tip : tip_frames) {
        auto* g = new PoseGoal();
        g->setLinkName(tip);
        g->setRotationScale(position_only_ik ? 0.0 : rotation_scale);
        out.emplace_back(g);
    }
    if (center_joints_weight > 0) out.emplace_back(new CenterJointsGoal(center_joints_weight));
    if (avoid_joint_limits_weight > 0) out.emplace_back(new AvoidJointLimitsGoal(avoid_joint_limits_weight));
    if (minimal_displacement_weight > 0) out.emplace_back(new MinimalDisplacementGoal(minimal_displacement_weight));
}

// One batch of queries that share a goal structure, as every face hands it over.
struct Request {
    std::vector<const Goal*> goals;         // all goals: the plugin's defaults first (unless `replace`), then the caller's (:550-556)
    size_t n_pose_goals = 0;                // the leading goals are PoseGoals that take the per-query tip poses (the default goals of the tips)
    std::vector<std::string> fixed_joints;  // BioIKKinematicsQueryOptions::fixed_joints
    const std::vector<std::vector<double>>* seed_states = nullptr;  // [n][group variables]
    std::vector<double> tip_poses;          // [n][n_pose_goals][7], in the frame `base_frame` is given in
    double base_frame[7] = {0, 0, 0, 0, 0, 0, 1};  // global transform of the plugin's base frame (:487-502)
    std::vector<double> context;            // the full variable vector the seeds are laid over: context state or default positions (:465-472)
    double timeout = 0.0;                   // the caller's timeout [s], counted from the moment submit() is entered (:504)
    bool return_approximate_solution = false;
    const BioIKKinematicsQueryOptions* bio = nullptr;  // receives solution_fitness (:632-634)
};

class Engine {
    struct ProblemSet {  // one compiled problem per device for ONE goal structure
        std::vector<bioik_problem*> per_device;
        ~ProblemSet() {
            for (auto* p : per_device) bioik_problem_destroy(p);
        }
    };
    ModelView mv_;
    Settings settings_;
    std::vector<bioik_model*> models_;
    // compiled problems, least recently used first; bounded, so a caller that varies its goal weights from call to call does not grow
    // device and host memory without limit (a batch in flight keeps its set alive through its ticket)
    std::list<std::pair<std::string, std::shared_ptr<ProblemSet>>> cache_;
    static constexpr size_t kCacheCapacity = 32;
    uint64_t next_query_ = 0;  // global index of the next query: the random streams advance from call to call, as the reference's generator does

    std::shared_ptr<ProblemSet> problemFor(const std::vector<const Goal*>& goals, const std::vector<std::string>& fixed) {
        std::ostringstream key;
        key << std::hexfloat;  // exact weight bits: goals whose weights differ in any digit get their own compiled problem
        for (auto* g : goals) key << g->gpuOpcode() << ':' << g->gpuLinkName() << ':' << g->gpuVariableName() << ':' << g->getWeight() << ':' << g->isSecondary() << ';';
        for (auto& f : fixed) key << '#' << f;
        const std::string k = key.str();
        for (auto it = cache_.begin(); it != cache_.end(); ++it)
            if (it->first == k) {
                cache_.splice(cache_.end(), cache_, it);  // most recently used last
                return cache_.back().second;
            }
        std::vector<bioik_goal_desc> gd;
        for (auto* g : goals) {
            if (g->gpuOpcode() < 0) throw std::logic_error("bio_ik (MI355X): a goal without a device implementation reached the problem compiler");  // (submit() splits them off)
            bioik_goal_desc d{g->gpuOpcode(), -1, -1, g->isSecondary() ? 1 : 0, g->getWeight()};
            if (!g->gpuLinkName().empty()) {
                d.link = mv_.link_index(g->gpuLinkName());
                if (d.link < 0) throw std::runtime_error("link not found: " + g->gpuLinkName());  // problem.cpp:141
            }
            if (!g->gpuVariableName().empty()) {
                d.variable = mv_.variable_index(g->gpuVariableName());
                if (d.variable < 0) throw std::runtime_error("joint variable not found: " + g->gpuVariableName());  // problem.cpp:125
            }
            gd.push_back(d);
        }
        std::vector<int32_t> fixed_idx;
        for (auto& f : fixed) {
            const int l = mv_.joint_link_index(f);
            if (l < 0) throw std::runtime_error("joint not found: " + f);
            fixed_idx.push_back(l);
        }
        bioik_problem_desc pd{};
        pd.struct_size = sizeof(pd);
        pd.n_group_joints = (uint32_t)mv_.group_joints.size(), pd.group_joints = mv_.group_joints.data();
        pd.n_goals = (uint32_t)gd.size(), pd.goals = gd.data();
        pd.n_fixed_joints = (uint32_t)fixed_idx.size(), pd.fixed_joints = fixed_idx.data();
        auto set = std::make_shared<ProblemSet>();
        for (auto* m : models_) {
            bioik_problem* p = nullptr;
            if (bioik_problem_create(m, &pd, &p) != BIOIK_OK) throw std::runtime_error(std::string("bio_ik (MI355X): ") + bioik_last_error());
            set->per_device.push_back(p);
        }
        cache_.emplace_back(k, set);
        if (cache_.size() > kCacheCapacity) cache_.pop_front();
        return set;
    }

public:
    // A submitted batch.  Holds everything the post-processing needs; the arrays the device writes into live here until wait().
    struct Ticket {
        size_t n = 0;
        std::shared_ptr<void> problems;  // keeps the compiled problems alive
        std::vector<bioik_problem*> handles;
        std::vector<uint64_t> tickets;   // one per device shard (0: an empty shard)
        std::vector<double> seeds, params, sol, fit;
        std::vector<int32_t> suc, steps, active;
        bool approximate = false, failed = false, waited = false;
        const BioIKKinematicsQueryOptions* bio = nullptr;  // receives solution_fitness in wait(): the caller's options object must outlive the wait
        // goals without a device implementation (the hybrid path): every query was searched `replicas` times with independent random streams over
        // the device-capable goals; wait() scores the candidates with the host goals added and keeps the best
        size_t replicas = 1;
        std::vector<const Goal*> host_goals, device_goals;  // goals without / with a device implementation (the latter: what the device's fitness figures hold)
        double dmax_sq = 0.0;  // min(dpos, dtwist)^2: what a primary goal of a class the success test does not know must stay below (problem.cpp:327-334)
        Ticket() {}
        Ticket(const Ticket&) = delete;
        Ticket& operator=(const Ticket&) = delete;
        // A ticket that is dropped -- forgotten, or unwound by an exception -- while its solves are in flight: the device still writes into the
        // arrays above (through the handle's staging arena at completion), so they must not go before the solves are over.
        ~Ticket() {
            if (!waited)
                for (size_t r = 0; r < handles.size() && r < tickets.size(); r++)
                    if (tickets[r]) (void)bioik_solve_batch_wait(handles[r], tickets[r]);
        }
    };

    Engine() {}
    Engine(const Engine&) = delete;
    ~Engine() { release(); }
    void release() {
        cache_.clear();
        for (auto* m : models_) bioik_model_destroy(m);
        models_.clear();
    }
    bool ready() const { return !models_.empty(); }
    const Settings& settings() const { return settings_; }

    // `md` describes the model for the device (bioik_model_desc, include/bioik_hip.h), `mv` for the host-side logic of this file
    void initialize(const bioik_model_desc& md, const ModelView& mv, const Settings& s) {
        release();
        mv_ = mv, settings_ = s;
        solverMode(s.mode);  // (an unknown mode is a configuration error: throw here, as IKFactory::create does at load time)
        if (s.gpu_schedule != "latency" && s.gpu_schedule != "throughput" && s.gpu_schedule != "auto")
            throw std::runtime_error("bio_ik (MI355X): gpu_schedule must be 'latency', 'throughput' or 'auto'");
        if (settings_.devices.empty()) settings_.devices.push_back(0);
        for (int dev : settings_.devices) {
            bioik_model* m = nullptr;
            if (bioik_model_create(&md, dev, &m) != BIOIK_OK) {
                const std::string msg = bioik_last_error();
                release();
                throw std::runtime_error("bio_ik (MI355X): " + msg);
            }
            models_.push_back(m);
        }
        next_query_ = 0;
    }

    std::shared_ptr<Ticket> submit(const Request& rq) {
        const auto t_entry = std::chrono::steady_clock::now();
        if (models_.empty()) throw std::runtime_error("bio_ik (MI355X): plugin not initialised");
        auto tk = std::make_shared<Ticket>();
        const size_t n = rq.seed_states ? rq.seed_states->size() : 0, V = mv_.n_variables;
        tk->n = n, tk->approximate = rq.return_approximate_solution, tk->bio = rq.bio;
        // Goals without a device implementation (std::function callbacks, user subclasses: the reference calls them through Goal::evaluate inside its
        // solver loop, src/problem.cpp:244-257) cannot steer a device search.  The hybrid path: the device searches over the goals it CAN evaluate,
        // `gpu_host_goal_candidates` times per query with independent random streams; wait() scores every candidate with the remaining goals on the host
        // (bio_ik/goal_eval.h) and returns the best.  A weaker coupling than the reference's -- the callbacks choose among candidates instead of
        // shaping them -- documented in DESIGN.md section 7.
        std::vector<const Goal*> device_goals;
        for (const Goal* g : rq.goals) (g->gpuOpcode() >= 0 ? device_goals : tk->host_goals).push_back(g);
        tk->device_goals = device_goals;
        if (!tk->host_goals.empty()) {
            if (device_goals.empty())
                throw std::runtime_error("bio_ik (MI355X): every goal of this request is a host callback (JointFunctionGoal, LinkFunctionGoal or a user-defined Goal): the "
                                         "device search needs at least one goal with a device implementation to steer it");
            if (!mv_.link_frame || mv_.var_max_velocity.size() != mv_.n_variables)
                throw std::runtime_error("bio_ik (MI355X): this build of the plugin cannot evaluate goals on the host (no link_frame in its ModelView)");
            tk->replicas = (size_t)std::max(1, settings_.gpu_host_goal_candidates);
            const double dmax = std::min(settings_.dpos, settings_.dtwist);
            tk->dmax_sq = dmax >= 1e150 ? DBL_MAX : dmax * dmax;
        }
        const size_t K = tk->replicas;
        std::shared_ptr<ProblemSet> set = problemFor(device_goals, rq.fixed_joints);
        tk->problems = set, tk->handles = set->per_device;
        bioik_problem* problem = tk->handles.front();
        const size_t P = (size_t)bioik_problem_param_count(problem);
        tk->active.resize((size_t)bioik_problem_active_variable_count(problem));
        bioik_problem_active_variables(problem, tk->active.data());
        // the seed states over the context / default state (:465-485)
        // (row k K + j: candidate j of query k; K = 1 unless host goals are present)
        tk->seeds.resize(n * K * V);
        for (size_t k = 0; k < n; k++) {
            double* row = &tk->seeds[k * K * V];
            for (size_t v = 0; v < V; v++) row[v] = rq.context[v];
            const std::vector<double>& seed = (*rq.seed_states)[k];
            for (size_t i = 0; i < mv_.group_vars.size(); i++) row[mv_.group_vars[i]] = seed.at(i);
            for (size_t j = 1; j < K; j++) std::copy(row, row + V, row + j * V);
        }
        // per-query goal numbers: the default pose goals move into the model frame (:487-502, :540-546), then every goal writes its own
        tk->params.resize(n * K * P);
        std::vector<double> row;
        if (rq.n_pose_goals > device_goals.size()) throw std::logic_error("bio_ik (MI355X): the default pose goals must lead the goal list");
        for (size_t k = 0; k < n; k++) {
            row.clear();
            for (size_t gi = 0; gi < device_goals.size(); gi++) {
                const Goal* g = device_goals[gi];
                if (gi < rq.n_pose_goals) {
                    double m[7];
                    concat7(rq.base_frame, &rq.tip_poses.at((k * rq.n_pose_goals + gi) * 7), m);
                    auto* goal = static_cast<PoseGoal*>(const_cast<Goal*>(g));
                    goal->setPosition(Vector3(m[0], m[1], m[2]));
                    goal->setOrientation(Quaternion(m[3], m[4], m[5], m[6]));  // normalises (goal_types.h:146)
                }
                g->gpuParams(row);
            }
            for (size_t j = 0; j < K; j++)
                for (size_t i = 0; i < P; i++) tk->params[(k * K + j) * P + i] = row.at(i);
        }
        bioik_solve_params sp;
        bioik_default_solve_params(&sp);
        const SolverMode sm = solverMode(settings_.mode);
        sp.mode = sm.mode;
        sp.fk_mode = settings_.gpu_fk == "linear" ? BIOIK_FK_LINEAR : BIOIK_FK_EXACT;
        sp.schedule = settings_.gpu_schedule == "throughput" ? BIOIK_SCHEDULE_THROUGHPUT : (settings_.gpu_schedule == "auto" ? BIOIK_SCHEDULE_AUTO : BIOIK_SCHEDULE_LATENCY);
        sp.population = settings_.gpu_population, sp.islands = sm.threads > 1 ? sm.threads : settings_.gpu_islands, sp.max_steps = settings_.gpu_max_steps;
        sp.random_seed = (uint64_t)(uint32_t)settings_.random_seed;
        sp.dpos = settings_.dpos, sp.drot = settings_.drot, sp.dtwist = settings_.dtwist;
        sp.no_wipeout = settings_.no_wipeout ? 1 : 0;
        sp.island_sync = (settings_.gpu_island_sync && sp.islands != 1) ? 1 : 0;  // (islands 0 = BIOIK_ISLANDS_AUTO: sized per call, stopping each other)
        const size_t rows = n * K;
        tk->sol.resize(rows * V), tk->fit.resize(rows), tk->suc.resize(rows), tk->steps.resize(rows);
        // problem.timeout = t0 + timeout with t0 taken at entry (:448, :504): what the marshalling above has used is off the budget (one
        // step always runs, ik_parallel.h:160)
        if (rq.timeout > 0.0) {
            const double used = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count();
            sp.timeout = std::max(rq.timeout - used, 1e-6);
        }
        // contiguous shards over the devices, each submitted to its own handle (its own streams): no exchange between shards, and the
        // query-indexed random streams make the result independent of the split
        const size_t W = tk->handles.size();
        tk->tickets.assign(W, 0);
        // gpu_islands: 0 (BIOIK_ISLANDS_AUTO) sizes the islands to the call: resolved ONCE per request, for the largest shard, so that every shard of a request
        // that is cut over several devices runs the same solve (rows = 2049 on two devices would otherwise run 1024 queries on four islands and 1025 on one)
        if (W > 1 && sp.islands <= 0 && rows > 0) {
            int32_t isl = 1, sync = 0;
            if (bioik_resolve_islands(tk->handles[0], &sp, (rows + W - 1) / W, &isl, &sync) == BIOIK_OK) sp.islands = isl, sp.island_sync = sync;
        }
        const uint64_t first = settings_.gpu_reproducible_calls ? 0 : next_query_;
        next_query_ += rows;
        for (size_t r = 0; r < W; r++) {
            const size_t a = r * rows / W, b = (r + 1) * rows / W;
            if (a == b) continue;
            bioik_problem_set_first_query(tk->handles[r], first + a);
            if (bioik_solve_batch_submit(tk->handles[r], &sp, b - a, &tk->seeds[a * V], P ? &tk->params[a * P] : nullptr, &tk->sol[a * V], &tk->fit[a], &tk->suc[a],
                                         &tk->steps[a], &tk->tickets[r]) != BIOIK_OK)
                tk->failed = true;  // device errors never abort the caller: every query of the batch reports NO_IK_SOLUTION
        }
        return tk;
    }

    // Waits for the batch; solutions [n][group variables], ok[k] = accurate solution or an approximate one was asked for (:638-641).
    // Returns true iff every query is ok.
    bool wait(Ticket& tk, std::vector<std::vector<double>>& solutions, std::vector<uint8_t>& ok) {
        const size_t n = tk.n, V = mv_.n_variables, K = tk.replicas;
        for (size_t r = 0; r < tk.handles.size(); r++)
            if (tk.tickets[r] && bioik_solve_batch_wait(tk.handles[r], tk.tickets[r]) != BIOIK_OK) tk.failed = true;
        tk.waited = true;
        solutions.assign(n, std::vector<double>());
        ok.assign(n, 0);
        if (tk.failed) return false;
        bool all_ok = true;
        std::unique_ptr<HostGoalProblem> host;  // the goals the device did not see, evaluated through their own describe() / evaluate()
        std::vector<double> positions(V);
        std::unique_ptr<HostGoalProblem> host_dev;  // the device-capable goals on the host: only for candidates the device accepted and the host rejects (below)
        HostGoalProblem::Model hm;
        if (!tk.host_goals.empty() && n) {
            for (size_t v = 0; v < V; v++)
                hm.info.addVariable(mv_.var_min[v], mv_.var_max[v], mv_.var_bounded[v] != 0, mv_.var_max_velocity[v], mv_.var_revolute[v] != 0,
                                    v < mv_.var_prismatic.size() && mv_.var_prismatic[v] != 0);
            hm.variable_index = mv_.variable_index;
            const ModelView* mv = &mv_;
            hm.link_frame = [mv](const std::string& name, const std::vector<double>& p) {
                double f[7];
                mv->link_frame(name, p.data(), f);
                return Frame(Vector3(f[0], f[1], f[2]), Quaternion(f[3], f[4], f[5], f[6]));
            };
            host.reset(new HostGoalProblem(hm, tk.host_goals, std::vector<int>(tk.active.begin(), tk.active.end()), std::vector<double>(&tk.seeds[0], &tk.seeds[0] + V)));
        }
        for (size_t k = 0; k < n; k++) {
            size_t best = k * K;
            if (host) {
                // ik_parallel.h:220-269 over the candidates: the accepted ones ranked by primary + secondary fitness, else the least primary fitness --
                // with the host goals' weighted costs added to the device's figures, and a candidate accepted only if every PRIMARY host goal also
                // passes the success rule for goal classes the test does not know: weight^2 cost < min(dpos, dtwist)^2 (problem.cpp:327-334)
                host->setInitialGuess(std::vector<double>(&tk.seeds[k * K * V], &tk.seeds[k * K * V] + V));
                double best_fit = DBL_MAX;
                bool best_ok = false;
                for (size_t j = 0; j < K; j++) {
                    const size_t r = k * K + j;
                    double* st = &tk.sol[r * V];
                    postprocess(st, &tk.seeds[r * V], tk.active);
                    positions.assign(st, st + V);
                    const std::vector<double> e = host->evaluateGoals(positions);
                    double prim = 0.0, sec = 0.0;
                    bool pass = tk.suc[r] != 0;
                    for (size_t g = 0; g < e.size(); g++) {
                        const double d = e[g] * host->weightSq(g);
                        if (host->isSecondary(g)) {
                            sec += d;
                        } else {
                            prim += d;
                            if (!(d < tk.dmax_sq)) pass = false;
                        }
                    }
                    // (the device's fitness of an accepted candidate already holds its secondary goals, of a rejected one the primary goals only;
                    // a candidate the host rejects is compared on PRIMARY fitness, like one the device rejected: if the device had accepted it, its figure
                    // is replaced by the primary goals of the device-capable list evaluated here -- the same closed forms, tests/cpp/test_goal_eval.cpp)
                    double device_part = tk.fit[r];
                    if (!pass && tk.suc[r] != 0) {
                        if (!host_dev) host_dev.reset(new HostGoalProblem(hm, tk.device_goals, std::vector<int>(tk.active.begin(), tk.active.end()), std::vector<double>(&tk.seeds[0], &tk.seeds[0] + V)));
                        host_dev->setInitialGuess(std::vector<double>(&tk.seeds[k * K * V], &tk.seeds[k * K * V] + V));
                        const std::vector<double> ed = host_dev->evaluateGoals(positions);
                        device_part = 0.0;
                        for (size_t g = 0; g < ed.size(); g++)
                            if (!host_dev->isSecondary(g)) device_part += ed[g] * host_dev->weightSq(g);
                    }
                    const double total = device_part + prim + (pass ? sec : 0.0);
                    tk.suc[r] = pass ? 1 : 0, tk.fit[r] = total;
                    if ((pass && !best_ok) || (pass == best_ok && total < best_fit)) best = r, best_fit = total, best_ok = pass;
                }
            } else {
                postprocess(&tk.sol[best * V], &tk.seeds[best * V], tk.active);
            }
            const double* st = &tk.sol[best * V];
            for (int gv : mv_.group_vars) solutions[k].push_back(st[gv]);  // map the result to the group's variables (:619-629)
            ok[k] = (tk.suc[best] || tk.approximate) ? 1 : 0;
            all_ok = all_ok && ok[k];
            if (tk.bio && k == n - 1) tk.bio->solution_fitness = tk.fit[best];  // :632-634
        }
        return all_ok;
    }

    // What the reference does to the solver's answer before it hands it out (:580-616), on ONE full variable vector `st` solved from `seed`:
    // the angle wrap of the active revolute variables, then RobotModel::enforcePositionBounds.
    void postprocess(double* st, const double* seed, const std::vector<int32_t>& active) const {
        const size_t V = mv_.n_variables;
        {
            {
            if (!mv_.has_mimic)
                for (int ivar : active) {  // wrap angles (:580-613)
                    if (!mv_.var_revolute[ivar]) continue;
                    double v = st[ivar];
                    const double r = seed[ivar], lo = mv_.var_min[ivar], hi = mv_.var_max[ivar];
                    if (r < v - M_PI || r > v + M_PI) {  // move close to the initial guess
                        v -= r, v /= (2 * M_PI), v += 0.5, v -= std::floor(v), v -= 0.5, v *= (2 * M_PI), v += r;
                    }
                    if (v > hi) v -= std::ceil(std::max(0.0, v - hi) / (2 * M_PI)) * (2 * M_PI);  // wrap at the joint limits
                    if (v < lo) v += std::ceil(std::max(0.0, lo - v) / (2 * M_PI)) * (2 * M_PI);
                    if (v < lo) v = lo;  // clamp at the edges
                    if (v > hi) v = hi;
                    st[ivar] = v;
                }
            if (mv_.enforce_bounds) {
                mv_.enforce_bounds(st);  // RobotModel::enforcePositionBounds (:616)
            } else {
                for (size_t v = 0; v < V; v++) {  // its effect on one-variable joints: clamp bounded variables, wrap continuous revolute ones to (-pi, pi]
                    if (mv_.var_bounded[v]) {
                        st[v] = std::min(std::max(st[v], mv_.var_min[v]), mv_.var_max[v]);
                    } else if (mv_.var_revolute[v] && (st[v] <= -M_PI || st[v] > M_PI)) {
                        st[v] = std::fmod(st[v], 2 * M_PI);
                        if (st[v] <= -M_PI) st[v] += 2 * M_PI;
                        else if (st[v] > M_PI) st[v] -= 2 * M_PI;
                    }
                }
            }
            }
        }
    }
    // the settings a caller may change between calls (budgets, thresholds, seed); the devices stay as initialised
    void updateSettings(const Settings& s) {
        solverMode(s.mode);
        const std::vector<int> devices = settings_.devices;
        settings_ = s;
        settings_.devices = devices;
    }
    const ModelView& modelView() const { return mv_; }
};

}  // namespace core
}  // namespace bio_ik

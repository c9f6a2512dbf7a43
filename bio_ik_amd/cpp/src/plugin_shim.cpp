// src/plugin_shim.cpp — the Python package's face of bio_ik/plugin_core.h.
//
// `bio_ik_amd/plugin.py` (the Python class with the reference plugin's method names, src/kinematics_plugin.cpp:117-671) keeps no copy of
// what the plugin does around the solver call: seed mapping, goal marshalling, the angle wrap and the bounds (kinematics_plugin.cpp:
// 465-641) are core::Engine's, reached through the few C entry points below (ctypes).  The Python side hands over flat arrays: the
// model as a bioik_model_desc plus its name tables, the caller's goals as (opcode, link, variable, weight, secondary, numbers) records
// -- the form every built-in goal of bio_ik_amd/goals.py serialises itself into.  Not part of the drop-in boundary (include/bioik_hip.h).
#include <chrono>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include <bio_ik/bio_ik.h>
#include <bio_ik/plugin_core.h>

namespace {

struct WireGoal : bio_ik::Goal {  // a goal of the Python package, already serialised
    int opcode = -1;
    std::string link, variable;
    std::vector<double> numbers;
    void setSecondary(bool s) { secondary_ = s; }
    int gpuOpcode() const override { return opcode; }
    std::string gpuLinkName() const override { return link; }
    std::string gpuVariableName() const override { return variable; }
    void gpuParams(std::vector<double>& out) const override { out.insert(out.end(), numbers.begin(), numbers.end()); }
};

struct InFlight {
    std::shared_ptr<bio_ik::core::Engine::Ticket> ticket;
    std::vector<std::vector<double>> seed_states;
    std::vector<std::unique_ptr<bio_ik::Goal>> caller_goals;
};

thread_local std::string g_error;

template <class F>
int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
    } catch (...) {
        g_error = "unknown error";
    }
    return 1;
}

}  // namespace

extern "C" {

typedef struct bioik_plugin_settings {  // kinematics.yaml keys (kinematics_plugin.cpp:243-328) + the additive gpu_* keys
    const char* mode;
    const char* gpu_fk;
    const char* gpu_schedule;
    int32_t random_seed, no_wipeout, position_only_ik, gpu_population, gpu_islands, gpu_max_steps, gpu_reproducible_calls, n_devices;
    const int32_t* devices;
    double dpos, drot, dtwist;  // negative dpos / drot: not set (DBL_MAX)
    double rotation_scale, center_joints_weight, avoid_joint_limits_weight, minimal_displacement_weight;
} bioik_plugin_settings;

typedef struct bioik_plugin_goal {
    int32_t opcode, secondary, n_numbers, reserved;
    const char* link;      // NULL / "": none
    const char* variable;  // NULL / "": none
    double weight;
    const double* numbers;  // [n_numbers] the goal's per-query numbers (the same for every query of the batch)
} bioik_plugin_goal;

struct bioik_plugin {
    bio_ik::core::Engine engine;
    std::vector<std::string> link_names, joint_names, variable_names, tip_frames;
    std::vector<std::unique_ptr<bio_ik::Goal>> default_goals;
    std::unordered_map<uint64_t, InFlight> in_flight;
    uint64_t next_ticket = 1;
    std::mutex mutex;
};

static bio_ik::core::Settings coreSettings(const bioik_plugin_settings& s) {
    bio_ik::core::Settings c;
    c.mode = s.mode ? s.mode : "bio2_memetic";
    c.gpu_fk = s.gpu_fk ? s.gpu_fk : "exact";
    c.gpu_schedule = s.gpu_schedule ? s.gpu_schedule : "auto";
    c.random_seed = s.random_seed, c.no_wipeout = s.no_wipeout != 0;
    c.dpos = s.dpos < 0 ? DBL_MAX : s.dpos, c.drot = s.drot < 0 ? DBL_MAX : s.drot, c.dtwist = s.dtwist;
    c.gpu_population = s.gpu_population, c.gpu_islands = s.gpu_islands, c.gpu_max_steps = s.gpu_max_steps;
    c.gpu_reproducible_calls = s.gpu_reproducible_calls != 0;
    c.devices.assign(s.devices, s.devices + s.n_devices);
    return c;
}

static void makeDefaults(bioik_plugin& p, const bioik_plugin_settings& s) {
    bio_ik::core::makeDefaultGoals(p.tip_frames, s.rotation_scale, s.position_only_ik != 0, s.center_joints_weight, s.avoid_joint_limits_weight,
                                   s.minimal_displacement_weight, p.default_goals);
}

const char* bioik_plugin_last_error() { return g_error.c_str(); }

// kinematics_plugin.cpp:191-335 (load).  link_names / joint_names [n_links] (the joint above each link), variable_names [n_variables],
// group_joints [n_group_joints] link indices of the group's active joints in order, tip_frames [n_tips].
int bioik_plugin_create(const bioik_model_desc* md, const char* const* link_names, const char* const* joint_names, const char* const* variable_names,
                        uint32_t n_group_joints, const int32_t* group_joints, uint32_t n_tips, const char* const* tip_frames, const bioik_plugin_settings* s,
                        bioik_plugin** out) {
    return guarded([&] {
        std::unique_ptr<bioik_plugin> p(new bioik_plugin());
        for (uint32_t l = 0; l < md->n_links; l++) p->link_names.push_back(link_names[l]), p->joint_names.push_back(joint_names[l] ? joint_names[l] : "");
        for (uint32_t v = 0; v < md->n_variables; v++) p->variable_names.push_back(variable_names[v]);
        for (uint32_t t = 0; t < n_tips; t++) p->tip_frames.push_back(tip_frames[t]);
        bio_ik::core::ModelView mv;
        mv.n_variables = md->n_variables;
        mv.var_revolute.assign(md->n_variables, 0);
        for (uint32_t l = 0; l < md->n_links; l++) {
            if (md->joint_type[l] == BIOIK_JOINT_REVOLUTE) mv.var_revolute[md->joint_first_variable[l]] = 1;
            mv.has_mimic = mv.has_mimic || md->joint_mimic[l] >= 0;
        }
        mv.var_bounded.assign(md->var_bounded, md->var_bounded + md->n_variables);
        mv.var_min.assign(md->var_min, md->var_min + md->n_variables), mv.var_max.assign(md->var_max, md->var_max + md->n_variables);
        for (uint32_t i = 0; i < n_group_joints; i++) {
            const int j = group_joints[i], t = md->joint_type[j];
            const int nv = t == BIOIK_JOINT_FLOATING ? 7 : (t == BIOIK_JOINT_PLANAR ? 3 : 1);
            for (int vi = 0; vi < nv; vi++) mv.group_vars.push_back(md->joint_first_variable[j] + vi);  // every variable of the joint (:473-484)
            mv.group_joints.push_back(j);
        }
        bioik_plugin* raw = p.get();  // (the look-ups live as long as the engine that holds them)
        auto find = [](const std::vector<std::string>& names, const std::string& n) {
            for (size_t i = 0; i < names.size(); i++)
                if (names[i] == n) return (int)i;
            return -1;
        };
        mv.link_index = [raw, find](const std::string& n) { return find(raw->link_names, n); };
        mv.variable_index = [raw, find](const std::string& n) { return find(raw->variable_names, n); };
        mv.joint_link_index = [raw, find](const std::string& n) { return n.empty() ? -1 : find(raw->joint_names, n); };
        p->engine.initialize(*md, mv, coreSettings(*s));
        makeDefaults(*p, *s);
        *out = p.release();
    });
}

void bioik_plugin_destroy(bioik_plugin* p) { delete p; }

// the keys a caller may change between calls (everything but the devices)
int bioik_plugin_update(bioik_plugin* p, const bioik_plugin_settings* s) {
    return guarded([&] {
        std::lock_guard<std::mutex> lock(p->mutex);
        p->engine.updateSettings(coreSettings(*s));
        makeDefaults(*p, *s);
    });
}

uint32_t bioik_plugin_group_variable_count(const bioik_plugin* p) { return (uint32_t)p->engine.modelView().group_vars.size(); }
void bioik_plugin_group_variables(const bioik_plugin* p, int32_t* out) {
    const auto& gv = p->engine.modelView().group_vars;
    for (size_t i = 0; i < gv.size(); i++) out[i] = gv[i];
}

// kinematics_plugin.cpp:437-578 for n queries: seeds [n][group variables]; tip_poses [n][tips][7] in the frame `base_frame` [7] is the
// global transform of (ignored with `replace`); context [n_variables]; goals: the CALLER's goals (the plugin's defaults are prepended
// here unless `replace`, :550-556).  Nothing is waited for.
int bioik_plugin_submit(bioik_plugin* p, uint64_t n, const double* seeds, const double* tip_poses, const double* base_frame, const double* context,
                        uint32_t n_goals, const bioik_plugin_goal* goals, int32_t replace, uint32_t n_fixed, const char* const* fixed_joints, double timeout,
                        int32_t return_approximate_solution, uint64_t* ticket) {
    return guarded([&] {
        std::lock_guard<std::mutex> lock(p->mutex);
        const bio_ik::core::ModelView& mv = p->engine.modelView();
        InFlight f;
        bio_ik::core::Request rq;
        if (!replace)
            for (auto& g : p->default_goals) rq.goals.push_back(g.get());
        rq.n_pose_goals = replace ? 0 : p->tip_frames.size();
        for (uint32_t i = 0; i < n_goals; i++) {
            auto* g = new WireGoal();
            f.caller_goals.emplace_back(g);
            g->opcode = goals[i].opcode;
            if (goals[i].link) g->link = goals[i].link;
            if (goals[i].variable) g->variable = goals[i].variable;
            g->setWeight(goals[i].weight);
            g->numbers.assign(goals[i].numbers, goals[i].numbers + goals[i].n_numbers);
            g->setSecondary(goals[i].secondary != 0);
            rq.goals.push_back(g);
        }
        for (uint32_t i = 0; i < n_fixed; i++) rq.fixed_joints.push_back(fixed_joints[i]);
        const size_t G = mv.group_vars.size();
        f.seed_states.resize(n);
        for (uint64_t k = 0; k < n; k++) f.seed_states[k].assign(seeds + k * G, seeds + (k + 1) * G);
        rq.seed_states = &f.seed_states;
        if (rq.n_pose_goals) rq.tip_poses.assign(tip_poses, tip_poses + n * rq.n_pose_goals * 7);
        for (int c = 0; c < 7; c++) rq.base_frame[c] = base_frame[c];
        rq.context.assign(context, context + mv.n_variables);
        rq.timeout = timeout, rq.return_approximate_solution = return_approximate_solution != 0;
        f.ticket = p->engine.submit(rq);
        *ticket = p->next_ticket++;
        p->in_flight.emplace(*ticket, std::move(f));
    });
}

// kinematics_plugin.cpp:580-641: solutions [n][group variables], ok [n] (accurate, or approximate ones were asked for), fitness [n]
int bioik_plugin_wait(bioik_plugin* p, uint64_t ticket, double* solutions, uint8_t* ok, double* fitness) {
    return guarded([&] {
        InFlight f;
        {
            std::lock_guard<std::mutex> lock(p->mutex);
            auto it = p->in_flight.find(ticket);
            if (it == p->in_flight.end()) throw std::runtime_error("bio_ik (MI355X): unknown ticket");
            f = std::move(it->second);
            p->in_flight.erase(it);
        }
        std::vector<std::vector<double>> sols;
        std::vector<uint8_t> oks;
        p->engine.wait(*f.ticket, sols, oks);
        const size_t G = p->engine.modelView().group_vars.size();
        for (size_t k = 0; k < f.ticket->n; k++) {
            ok[k] = oks[k];
            fitness[k] = f.ticket->failed ? DBL_MAX : f.ticket->fit[k];
            for (size_t i = 0; i < G; i++) solutions[k * G + i] = sols[k].size() == G ? sols[k][i] : f.seed_states[k][i];
        }
    });
}

// What MoveIt does with a list of poses: ONE searchPositionIK call per pose (kinematics_plugin.cpp:437-655), each with the caller's `timeout`, the next call when the
// last has returned.  n such calls in a row -- submit and wait of one query each -- with the wall time of every call [s]: the figures of bench.py's
// `one_pose_timeouts` leg, taken around the plugin core without the Python face's marshalling in them.  Arrays as for bioik_plugin_submit / _wait.
int bioik_plugin_search_each(bioik_plugin* p, uint64_t n, const double* seeds, const double* tip_poses, const double* base_frame, const double* context, uint32_t n_goals,
                             const bioik_plugin_goal* goals, int32_t replace, uint32_t n_fixed, const char* const* fixed_joints, double timeout,
                             int32_t return_approximate_solution, double* solutions, uint8_t* ok, double* fitness, double* seconds) {
    const size_t G = p->engine.modelView().group_vars.size();
    const size_t T = replace ? 0 : p->tip_frames.size();
    for (uint64_t k = 0; k < n; k++) {
        const auto t0 = std::chrono::steady_clock::now();
        uint64_t ticket = 0;
        int rc = bioik_plugin_submit(p, 1, seeds + k * G, T ? tip_poses + k * T * 7 : nullptr, base_frame, context, n_goals, goals, replace, n_fixed, fixed_joints, timeout,
                                     return_approximate_solution, &ticket);
        if (rc == 0) rc = bioik_plugin_wait(p, ticket, solutions + k * G, ok + k, fitness + k);
        if (rc != 0) return rc;
        seconds[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

// the post-processing of bioik_plugin_wait on its own (:580-616), on full variable vectors: states [n][n_variables] in place
int bioik_plugin_postprocess(const bioik_plugin* p, uint64_t n, const double* seeds, double* states, uint32_t n_active, const int32_t* active) {
    return guarded([&] {
        const std::vector<int32_t> act(active, active + n_active);
        const size_t V = p->engine.modelView().n_variables;
        for (uint64_t k = 0; k < n; k++) p->engine.postprocess(states + k * V, seeds + k * V, act);
    });
}

}  // extern "C"

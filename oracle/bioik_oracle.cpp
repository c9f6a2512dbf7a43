// bioik_oracle.cpp — ORACLE (test infrastructure): C API over the restatement in orc_*.h.
// See bioik_oracle.h for the parity status.  Island driver / best-of selection restate reference
// src/ik_parallel.h:148-270; angle wrapping restates src/kinematics_plugin.cpp:580-616.
#include "bioik_oracle.h"

#include <atomic>
#include <climits>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <thread>

#include "orc_evolution.h"

using namespace orc;

static thread_local std::string g_err;
static int fail(const std::exception& e) {
    g_err = e.what();
    return -1;
}

struct OrcProblem {
    const Model* model;
    Problem problem;
    OrcProblem(const Model* m, const bioik_problem_desc& d) : model(m), problem(m, d) {}
};

static void vars_from_genes(const Problem& p, const double* seed, const double* genes, std::vector<double>& vars) {
    vars.assign(seed, seed + p.model->vars.size());
    for (size_t i = 0; i < p.active_variables.size(); i++) vars[p.active_variables[i]] = genes[i];
}

template <class E>
static void dump_state(E& ik, double* species_genes, double* species_fitness, double* solution, double* solution_fitness) {
    size_t D = ik.D();
    for (size_t s = 0; s < 2; s++) {
        for (size_t i = 0; i < 2; i++) {
            std::memcpy(species_genes + ((s * 2 + i) * 2 + 0) * D, ik.species[s].individuals[i].genes.data(), D * sizeof(double));
            std::memcpy(species_genes + ((s * 2 + i) * 2 + 1) * D, ik.species[s].individuals[i].gradients.data(), D * sizeof(double));
        }
        species_fitness[s] = ik.species[s].fitness;
    }
    std::memcpy(solution, ik.solution.data(), ik.solution.size() * sizeof(double));
    *solution_fitness = ik.solution_fitness;
}

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }
void orc_set_trig_mode(int mode) { trig_mode() = mode ? 1 : 0; }
void orc_shared_sincos(size_t n, const double* x, double* s, double* c) {
    for (size_t i = 0; i < n; i++) bioik_sincos(x[i], s + i, c + i);
}
int orc_get_trig_mode(void) { return trig_mode(); }
unsigned long long orc_debug_unbounded_candidates(void) { return unbounded_candidates().load(); }
void orc_set_quirk_mode(int mode) { quirk_mode() = mode ? 1 : 0; }

void* orc_model_create(const bioik_model_desc* desc) {
    try {
        return new Model(*desc);
    } catch (const std::exception& e) {
        fail(e);
        return nullptr;
    }
}
void orc_model_destroy(void* model) { delete (Model*)model; }

void* orc_problem_create(void* model, const bioik_problem_desc* desc) {
    try {
        return new OrcProblem((const Model*)model, *desc);
    } catch (const std::exception& e) {
        fail(e);
        return nullptr;
    }
}
void orc_problem_destroy(void* problem) { delete (OrcProblem*)problem; }

int orc_problem_info(void* problem, int32_t* out4) {
    const Problem& p = ((OrcProblem*)problem)->problem;
    out4[0] = (int32_t)p.active_variables.size();
    out4[1] = (int32_t)p.tip_link_indices.size();
    out4[2] = p.param_count;
    out4[3] = (int32_t)p.model->vars.size();
    return 0;
}
int orc_problem_active_variables(void* problem, int32_t* out) {
    const Problem& p = ((OrcProblem*)problem)->problem;
    for (size_t i = 0; i < p.active_variables.size(); i++) out[i] = (int32_t)p.active_variables[i];
    return 0;
}
int orc_problem_tip_links(void* problem, int32_t* out) {
    const Problem& p = ((OrcProblem*)problem)->problem;
    for (size_t i = 0; i < p.tip_link_indices.size(); i++) out[i] = p.tip_link_indices[i];
    return 0;
}
int orc_model_robot_info(void* model, double* out) {
    const Model& m = *(const Model*)model;
    for (size_t v = 0; v < m.vars.size(); v++) {
        const VarInfo& i = m.vars[v];
        double* o = out + v * 6;
        o[0] = i.clip_min;
        o[1] = i.clip_max;
        o[2] = i.span;
        o[3] = i.min;
        o[4] = i.max;
        o[5] = i.max_velocity_rcp;
    }
    return 0;
}
int orc_problem_velocity_weights(void* problem, double* out) {
    const Problem& p = ((OrcProblem*)problem)->problem;
    for (size_t i = 0; i < p.minimal_displacement_factors.size(); i++) out[i] = p.minimal_displacement_factors[i];
    return 0;
}

// ---- L1 ----
void orc_quat_mul_vec(const double* q4, const double* v3, double* out3) {
    Vec3 r;
    quat_mul_vec(Quat{q4[0], q4[1], q4[2], q4[3]}, Vec3{v3[0], v3[1], v3[2]}, r);
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
}
void orc_quat_mul_quat(const double* p4, const double* q4, double* out4) {
    Quat r;
    quat_mul_quat(Quat{p4[0], p4[1], p4[2], p4[3]}, Quat{q4[0], q4[1], q4[2], q4[3]}, r);
    out4[0] = r.x;
    out4[1] = r.y;
    out4[2] = r.z;
    out4[3] = r.w;
}
void orc_frame_concat(const double* a7, const double* b7, double* out7) {
    Frame r;
    concat(frame_from7(a7), frame_from7(b7), r);
    frame_to7(r, out7);
}
void orc_frame_invert(const double* a7, double* out7) {
    Frame r;
    invert(frame_from7(a7), r);
    frame_to7(r, out7);
}
void orc_frame_change(const double* a7, const double* b7, const double* c7, double* out7) {
    Frame r;
    change(frame_from7(a7), frame_from7(b7), frame_from7(c7), r);
    frame_to7(r, out7);
}
void orc_normalize_fast(double* q4) {
    Quat q = {q4[0], q4[1], q4[2], q4[3]};
    normalize_fast(q);
    q4[0] = q.x;
    q4[1] = q.y;
    q4[2] = q.z;
    q4[3] = q.w;
}
void orc_frame_twist(const double* a7, const double* b7, double* out6) { frame_twist(frame_from7(a7), frame_from7(b7), out6); }

// reference src/utils.h:348-367, driven as in test/utest.cpp:83-111
void orc_linear_int_distribution_hist(uint32_t seed, uint32_t n, uint32_t iters, double* hist) {
    std::mt19937 rng(seed);
    std::uniform_int_distribution<size_t> base(0, n);
    for (uint32_t i = 0; i < n; i++) hist[i] = 0;
    for (uint32_t it = 0; it < iters; it++) {
        while (true) {
            size_t v = base(rng) + base(rng);
            if (v < n) {
                hist[n - v - 1] += 1;
                break;
            }
        }
    }
}

// ---- L2 ----
int orc_fk(void* problem, size_t n, const double* vars, double* tip_frames, double* global_frames) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        RobotFK fk(p.model);
        fk.initialize(p.tip_link_indices);
        size_t V = p.model->vars.size(), T = p.tip_link_indices.size(), L = p.model->links.size();
        std::vector<double> v(V);
        for (size_t k = 0; k < n; k++) {
            v.assign(vars + k * V, vars + (k + 1) * V);
            fk.apply_configuration(v);
            for (size_t t = 0; t < T; t++) frame_to7(fk.tip_frames[t], tip_frames + (k * T + t) * 7);
            if (global_frames)
                for (size_t l = 0; l < L; l++) frame_to7(fk.global_frames[l], global_frames + (k * L + l) * 7);
        }
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
int orc_fk_genes(void* problem, size_t n, const double* seed, const double* genes, double* tip_frames) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        RobotFK fk(p.model);
        fk.initialize(p.tip_link_indices);
        size_t D = p.active_variables.size(), T = p.tip_link_indices.size();
        std::vector<double> v;
        for (size_t k = 0; k < n; k++) {
            vars_from_genes(p, seed, genes + k * D, v);
            fk.apply_configuration(v);
            for (size_t t = 0; t < T; t++) frame_to7(fk.tip_frames[t], tip_frames + (k * T + t) * 7);
        }
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
int orc_jacobian(void* problem, const double* seed, const double* base_genes, double* jac) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        RobotFK fk(p.model);
        fk.initialize(p.tip_link_indices);
        std::vector<double> v;
        vars_from_genes(p, seed, base_genes, v);
        fk.apply_configuration(v);
        fk.compute_jacobian(p.active_variables);
        std::memcpy(jac, fk.approx_jacobian.data(), fk.approx_jacobian.size() * sizeof(double));
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
int orc_approximator(void* problem, const double* seed, const double* base_genes, double* tip_frames, double* deltas, int32_t* mask) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        RobotFK fk(p.model);
        fk.initialize(p.tip_link_indices);
        std::vector<double> v;
        vars_from_genes(p, seed, base_genes, v);
        fk.apply_configuration(v);
        fk.initialize_mutation_approximator(p.active_variables);
        size_t D = p.active_variables.size(), T = p.tip_link_indices.size();
        for (size_t t = 0; t < T; t++) {
            frame_to7(fk.tip_frames[t], tip_frames + t * 7);
            for (size_t i = 0; i < D; i++) {
                frame_to7(fk.approx_frames[t][p.active_variables[i]], deltas + (t * D + i) * 7);
                if (mask) mask[t * D + i] = fk.approx_mask[t][p.active_variables[i]];
            }
        }
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
int orc_approx_eval(void* problem, const double* seed, const double* base_genes, size_t n, const double* genes, double* frames) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        RobotFK fk(p.model);
        fk.initialize(p.tip_link_indices);
        std::vector<double> v;
        vars_from_genes(p, seed, base_genes, v);
        fk.apply_configuration(v);
        fk.initialize_mutation_approximator(p.active_variables);
        size_t D = p.active_variables.size(), T = p.tip_link_indices.size();
        std::vector<Frame> out(T);
        for (size_t k = 0; k < n; k++) {
            const double* g = genes + k * D;
            fk.compute_approximate_mutations(1, &g, out.data());
            for (size_t t = 0; t < T; t++) frame_to7(out[t], frames + (k * T + t) * 7);
        }
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}

// ---- L4 ----
int orc_fitness_frames(void* problem, const double* seed, const double* goal_params, const double* frames, const double* genes,
                       double* primary, double* secondary) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        size_t T = p.tip_link_indices.size();
        std::vector<Frame> f(T), nullf(T, Frame{{0, 0, 0}, {0, 0, 0, 0}});
        for (size_t t = 0; t < T; t++) f[t] = frame_from7(frames + 7 * t);
        Query q{seed, goal_params};
        *primary = p.compute_goal_fitness(p.goals, q, f.data(), genes);
        *secondary = p.compute_goal_fitness(p.secondary_goals, q, nullf.data(), genes);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
int orc_fitness(void* problem, int fk_mode, size_t n, const double* seed, const double* goal_params, const double* base_genes,
                const double* genes, double* primary, double* secondary) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        RobotFK fk(p.model);
        fk.initialize(p.tip_link_indices);
        size_t D = p.active_variables.size(), T = p.tip_link_indices.size();
        std::vector<double> v;
        std::vector<Frame> out(T), nullf(T, Frame{{0, 0, 0}, {0, 0, 0, 0}});
        Query q{seed, goal_params};
        if (fk_mode == BIOIK_FK_LINEAR) {
            vars_from_genes(p, seed, base_genes, v);
            fk.apply_configuration(v);
            fk.initialize_mutation_approximator(p.active_variables);
        }
        for (size_t k = 0; k < n; k++) {
            const double* g = genes + k * D;
            if (fk_mode == BIOIK_FK_LINEAR) {
                fk.compute_approximate_mutations(1, &g, out.data());
            } else {
                vars_from_genes(p, seed, g, v);
                fk.apply_configuration(v);
                out = fk.tip_frames;
            }
            primary[k] = p.compute_goal_fitness(p.goals, q, out.data(), g);
            secondary[k] = p.compute_goal_fitness(p.secondary_goals, q, nullf.data(), g);
        }
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
int orc_check(void* problem, const bioik_solve_params* params, size_t n, const double* seed, const double* goal_params,
              const double* genes, int32_t* ok) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        RobotFK fk(p.model);
        fk.initialize(p.tip_link_indices);
        size_t D = p.active_variables.size();
        std::vector<double> v;
        Query q{seed, goal_params};
        double dpos = normalize_threshold(params->dpos), drot = normalize_threshold(params->drot), dtwist = normalize_threshold(params->dtwist);
        for (size_t k = 0; k < n; k++) {
            const double* g = genes + k * D;
            vars_from_genes(p, seed, g, v);
            fk.apply_configuration(v);
            ok[k] = p.check_solution(q, fk.tip_frames.data(), g, dpos, drot, dtwist) ? 1 : 0;
        }
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
void orc_pose_twist(const double* goal7, const double* tip7, double* out6) { kdl_pose_twist(frame_from7(goal7), frame_from7(tip7), out6); }

// ---- counter RNG ----
void orc_philox2x32(uint32_t key, uint32_t c0, uint32_t c1, uint32_t* out2) { philox2x32_10(key, c0, c1, out2); }
void orc_philox4x32(const uint32_t* key2, const uint32_t* ctr4, uint32_t* out4) { philox4x32_10(key2, ctr4, out4); }
uint32_t orc_child_word(uint32_t key, uint32_t ctr1, uint32_t child, uint32_t w) { return child_word_of(child_stream(key, ctr1), child, w); }
double orc_counter_gauss32(uint32_t word) { return counter_gauss_from32(word); }
double orc_counter_uniform(uint32_t key, uint32_t c0, uint32_t c1) {
    uint32_t o[2];
    philox2x32_10(key, c0, c1, o);
    return counter_uniform_from(o[0], o[1]);
}
uint32_t orc_query_key(uint64_t seed, uint64_t query, uint32_t island) { return query_key(seed, query, island); }
// the children a species' generation walks when secondary goals pre-select them (ik_evolution_2.cpp:366-378), as the solver's own random source draws the
// count: what tests/test_evaluation_count.py holds the host-side restatement of bench.py against
uint32_t orc_preselect_children(uint64_t seed, uint64_t query, uint32_t island, uint32_t step, uint32_t generation, uint32_t species_id, uint32_t lambda) {
    CounterRandom r;
    r.key = query_key(seed, query, island);
    r.set_context(step, generation, species_id);
    return (uint32_t)(r.preselect_count(2, lambda) - 2);
}

// ---- L3 ----
// the restated reference random sources, probed in the order reproduce() / step() consume them (cf. oracle/ref_driver.cpp)
int orc_reference_random_probe(int seed, size_t n_gauss, double* gauss, size_t n_index, uint64_t* index16, size_t n_fast, double* fast, size_t n_rng,
                               double* rng_uniform) {
    ReferenceRandom r((uint32_t)seed);
    const double* g = r.fast_random_gauss_n(n_gauss);
    for (size_t i = 0; i < n_gauss; i++) gauss[i] = g[i];
    for (size_t i = 0; i < n_index; i++) index16[i] = r.fast_random_index(16);
    for (size_t i = 0; i < n_fast; i++) fast[i] = r.fast_random();
    for (size_t i = 0; i < n_rng; i++) rng_uniform[i] = r.random();
    return 0;
}

int orc_reproduce_counter(void* problem, int population, uint32_t rng_key, int species, uint32_t generation, const double* parents,
                          double* children_genes, double* children_gradients) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        bioik_solve_params sp;
        std::memset(&sp, 0, sizeof(sp));
        sp.mode = BIOIK_MODE_BIO2_MEMETIC;
        sp.population = population;
        CounterRandom rng;
        rng.key = rng_key;
        Evolution2<CounterRandom> ik(&p, rng, sp);
        size_t D = p.active_variables.size(), V = p.model->vars.size();
        std::vector<double> seed(V, 0.0);
        // a neutral query: reproduce() does not touch goals
        std::vector<double> params((size_t)std::max(1, p.param_count), 0.0);
        // initialise buffers without running FK on garbage: use zeros as seed (only sizes matter here)
        Query q{seed.data(), params.data()};
        ik.initialize(q);
        std::vector<Individual> pop(2);
        for (int i = 0; i < 2; i++) {
            pop[i].genes.assign(parents + (i * 2 + 0) * D, parents + (i * 2 + 0) * D + D);
            pop[i].gradients.assign(parents + (i * 2 + 1) * D, parents + (i * 2 + 1) * D + D);
        }
        ik.rng.generation = generation;
        ik.rng.species = (uint32_t)species;
        ik.reproduce(pop);
        for (int c = 0; c < population; c++) {
            std::memcpy(children_genes + (size_t)c * D, ik.children[2 + c].genes.data(), D * sizeof(double));
            std::memcpy(children_gradients + (size_t)c * D, ik.children[2 + c].gradients.data(), D * sizeof(double));
        }
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}

struct OrcSolver {
    int rng_mode;
    std::unique_ptr<Evolution2<ReferenceRandom>> ref;
    std::unique_ptr<Evolution2<CounterRandom>> ctr;
    std::unique_ptr<GradientDescent<ReferenceRandom>> gd_ref;  // BIOIK_MODE_GD_C, BIOIK_MODE_GD, BIOIK_MODE_GD_R
    std::unique_ptr<GradientDescent<CounterRandom>> gd_ctr;
    std::unique_ptr<JacobianSolver<CounterRandom>> jac;        // BIOIK_MODE_JAC (thread 0 draws no random number)
    std::vector<double> seed, params;
};

void* orc_solver_create(void* problem, const bioik_solve_params* params, int rng_mode, uint32_t rng_key, const double* seed,
                        const double* goal_params) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        auto* s = new OrcSolver();
        s->rng_mode = rng_mode;
        s->seed.assign(seed, seed + p.model->vars.size());
        s->params.assign(goal_params, goal_params + p.param_count);
        s->params.push_back(0.0);
        Query q{s->seed.data(), s->params.data()};
        if (params->mode == BIOIK_MODE_GD_C || params->mode == BIOIK_MODE_GD || params->mode == BIOIK_MODE_GD_R) {
            if (rng_mode == ORC_RNG_REFERENCE) {
                s->gd_ref.reset(new GradientDescent<ReferenceRandom>(&p, *params, gradient_if_stuck(params->mode), ReferenceRandom(rng_key)));
                s->gd_ref->initialize(q);
            } else {
                CounterRandom r;
                r.key = rng_key;
                s->gd_ctr.reset(new GradientDescent<CounterRandom>(&p, *params, gradient_if_stuck(params->mode), r));
                s->gd_ctr->initialize(q);
            }
        } else if (params->mode == BIOIK_MODE_JAC) {
            s->jac.reset(new JacobianSolver<CounterRandom>(&p, *params));
            s->jac->initialize(q);
        } else if (rng_mode == ORC_RNG_REFERENCE) {
            s->ref.reset(new Evolution2<ReferenceRandom>(&p, ReferenceRandom(rng_key), *params));
            s->ref->initialize(q);
        } else {
            CounterRandom r;
            r.key = rng_key;
            s->ctr.reset(new Evolution2<CounterRandom>(&p, r, *params));
            s->ctr->initialize(q);
        }
        return s;
    } catch (const std::exception& e) {
        fail(e);
        return nullptr;
    }
}
void orc_solver_destroy(void* solver) { delete (OrcSolver*)solver; }
int orc_solver_step(void* solver) {
    try {
        auto* s = (OrcSolver*)solver;
        if (s->ref) s->ref->step();
        if (s->ctr) s->ctr->step();
        if (s->gd_ref) s->gd_ref->step();
        if (s->gd_ctr) s->gd_ctr->step();
        if (s->jac) s->jac->step();
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
int orc_solver_state(void* solver, double* species_genes, double* species_fitness, double* solution, double* solution_fitness) {
    auto* s = (OrcSolver*)solver;
    if (s->ref) dump_state(*s->ref, species_genes, species_fitness, solution, solution_fitness);
    if (s->ctr) dump_state(*s->ctr, species_genes, species_fitness, solution, solution_fitness);
    if (s->gd_ref || s->gd_ctr || s->jac) {  // point solvers: only the solution (getSolution(), ik_gradient.cpp:160, 287)
        const std::vector<double>& v = s->gd_ref ? s->gd_ref->get_solution() : (s->gd_ctr ? s->gd_ctr->get_solution() : s->jac->get_solution());
        for (size_t i = 0; i < v.size(); i++) solution[i] = v[i];
        *solution_fitness = 0.0;
    }
    return 0;
}
int orc_solver_check(void* solver, int32_t* success, double* fitness) {
    auto* s = (OrcSolver*)solver;
    bool ok = false;
    double f = 0;
    if (s->ref) s->ref->check(ok, f);
    if (s->ctr) s->ctr->check(ok, f);
    if (s->gd_ref) s->gd_ref->check(ok, f);
    if (s->gd_ctr) s->gd_ctr->check(ok, f);
    if (s->jac) s->jac->check(ok, f);
    *success = ok ? 1 : 0;
    *fitness = f;
    return 0;
}

// Analysis vehicle (tools/fork_simulation.py; nothing of the product): bio2 solves in which a query still unsolved after `fork_step` steps continues as
// `fork_islands` COPIES of its solver state that differ in their random streams from then on (copy 0 keeps the query's stream) and stop when one of them passes.
// out_steps[k]: steps until the first copy passed (max_steps: none did); out_plain[k]: the same query without forking.
int orc_fork_simulation(void* problem, const bioik_solve_params* params, size_t n, const double* seeds, const double* goal_params, int fork_step, int fork_islands,
                        int32_t* out_steps, int32_t* out_plain, int n_threads, uint64_t first_query_index) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        const size_t V = p.model->vars.size(), P = (size_t)p.param_count;
        std::atomic<size_t> next(0);
        auto worker = [&]() {
            Problem local = p;
            std::vector<double> zero_params(1, 0.0);
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= n) break;
                Query q{seeds + k * V, P ? goal_params + k * P : zero_params.data()};
                CounterRandom r;
                r.key = query_key(params->random_seed, first_query_index + k, 0u);
                Evolution2<CounterRandom> ik(&local, r, *params);
                ik.initialize(q);
                int steps = 0;
                bool ok = false;
                double f;
                std::vector<Evolution2<CounterRandom>> copies;
                int plain = params->max_steps;
                while (steps < params->max_steps) {
                    if (steps == fork_step && copies.empty() && fork_islands > 1) {
                        for (int i = 1; i < fork_islands; i++) {
                            copies.push_back(ik);
                            copies.back().rng.key = query_key(params->random_seed, first_query_index + k, (uint32_t)i);
                        }
                    }
                    ik.step();
                    steps++;
                    ik.check(ok, f);
                    if (ok && plain == params->max_steps) plain = steps;
                    bool any = ok;
                    for (auto& c : copies) {
                        c.step();
                        bool ok2;
                        double f2;
                        c.check(ok2, f2);
                        any = any || ok2;
                    }
                    if (any) break;
                }
                out_steps[k] = steps;
                // the plain run, continued if a copy ended the forked one first
                while (plain == params->max_steps && steps < params->max_steps) {
                    ik.step();
                    steps++;
                    ik.check(ok, f);
                    if (ok) plain = steps;
                }
                out_plain[k] = plain;
            }
        };
        std::vector<std::thread> th;
        for (int i = 0; i < (n_threads > 1 ? n_threads : 1); i++) th.emplace_back(worker);
        for (auto& t : th) t.join();
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}

int orc_solve_batch(void* problem, const bioik_solve_params* params, int rng_mode, size_t n, const double* seeds,
                    const double* goal_params, double* solutions, double* fitness, int32_t* success, int32_t* steps, int n_threads,
                    double timeout_s, uint64_t first_query_index) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        size_t V = p.model->vars.size();
        size_t P = (size_t)p.param_count;
        int islands = params->islands > 0 ? params->islands : 1;
        if (rng_mode == ORC_RNG_REFERENCE) ReferenceRandom::get_buffers((uint32_t)params->random_seed);  // build tables once, outside the threads
        std::atomic<size_t> next(0);
        std::atomic<int> failed(0);
        std::string err;
        std::mutex err_mtx;
        auto worker = [&]() {
            try {
                Problem local = p;  // GoalContext is mutated per evaluation in the reference -> per-thread copy (ik_base.h:156)
                std::vector<double> zero_params(1, 0.0);
                for (;;) {
                    size_t k = next.fetch_add(1);
                    if (k >= n) break;
                    Query q{seeds + k * V, P ? goal_params + k * P : zero_params.data()};
                    // ik_parallel.h:220-269 best-of over islands
                    IslandResult best;
                    double best_fitness = DBL_MAX;
                    bool have = false;
                    std::vector<IslandResult> rs;
                    for (int isl = 0; isl < islands; isl++) {
                        if (rng_mode == ORC_RNG_REFERENCE) {
                            // the reference clones the solver including its RNG state (utils.h:423): identical islands.
                            rs.push_back(run_island(&local, ReferenceRandom((uint32_t)params->random_seed), *params, q, timeout_s, isl));
                        } else {
                            CounterRandom r;
                            r.key = query_key(params->random_seed, first_query_index + k, (uint32_t)isl);
                            rs.push_back(run_island(&local, r, *params, q, timeout_s, isl));
                        }
                    }
                    size_t best_index = 0;
                    // bioik_solve_params::island_sync ("any island succeeds => all stop", ik_parallel.h:102, 160-178, in lock step): the islands that
                    // passed after the least number of steps are the candidates; the others would have been stopped at that step without having passed
                    int least_steps = INT_MAX;
                    if (params->island_sync)
                        for (size_t i = 0; i < rs.size(); i++)
                            if (rs[i].success && rs[i].steps < least_steps) least_steps = rs[i].steps;
                    for (size_t i = 0; i < rs.size(); i++) {
                        if (rs[i].success && (!params->island_sync || rs[i].steps == least_steps)) {
                            double f = rs[i].fitness;
                            if (!local.secondary_goals.empty()) {
                                std::vector<Frame> nullf(local.tip_link_indices.size(), Frame{{0, 0, 0}, {0, 0, 0, 0}});
                                std::vector<double> act(local.active_variables.size());
                                for (size_t a = 0; a < act.size(); a++) act[a] = rs[i].solution[local.active_variables[a]];
                                f += local.compute_goal_fitness(local.secondary_goals, q, nullf.data(), act.data());
                            }
                            if (f < best_fitness) best_fitness = f, best_index = i, have = true;
                        }
                    }
                    if (!have) {
                        for (size_t i = 0; i < rs.size(); i++)
                            if (rs[i].fitness < best_fitness) best_fitness = rs[i].fitness, best_index = i;
                    }
                    const IslandResult& r = rs[best_index];
                    std::memcpy(solutions + k * V, r.solution.data(), V * sizeof(double));
                    fitness[k] = best_fitness;
                    success[k] = r.success ? 1 : 0;
                    steps[k] = r.steps;
                }
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lock(err_mtx);
                err = e.what();
                failed = 1;
            }
        };
        if (n_threads <= 1) {
            worker();
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_threads; t++) th.emplace_back(worker);
            for (auto& t : th) t.join();
        }
        if (failed) {
            g_err = err;
            return -1;
        }
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}

// kinematics_plugin.cpp:580-616
int orc_wrap_angles(void* problem, const double* seed, double* state) {
    try {
        const Problem& p = ((OrcProblem*)problem)->problem;
        const Model& m = *p.model;
        for (size_t ivar : p.active_variables) {
            double v = state[ivar];
            if (m.is_revolute(ivar) && m.mimic_joints.empty()) {
                double r = seed[ivar];
                double lo = m.vars[ivar].min;
                double hi = m.vars[ivar].max;
                if (r < v - M_PI || r > v + M_PI) {
                    v -= r;
                    v /= (2 * M_PI);
                    v += 0.5;
                    v -= std::floor(v);
                    v -= 0.5;
                    v *= (2 * M_PI);
                    v += r;
                }
                if (v > hi) v -= std::ceil(std::max(0.0, v - hi) / (2 * M_PI)) * (2 * M_PI);
                if (v < lo) v += std::ceil(std::max(0.0, lo - v) / (2 * M_PI)) * (2 * M_PI);
                if (v < lo) v = lo;
                if (v > hi) v = hi;
            }
            state[ivar] = v;
        }
        // :616 RobotModel::enforcePositionBounds — clamp bounded variables into [min,max]
        // (MoveIt wraps continuous revolute joints into [-pi,pi]; restated for revolute unbounded variables)
        for (size_t v = 0; v < m.vars.size(); v++) {
            const VarInfo& info = m.vars[v];
            if (info.clip_max != DBL_MAX) {
                if (state[v] < info.min) state[v] = info.min;
                if (state[v] > info.max) state[v] = info.max;
            } else if (m.is_revolute(v)) {
                double& x = state[v];
                if (x < -M_PI || x > M_PI) {
                    x = std::fmod(x + M_PI, 2.0 * M_PI);
                    if (x < 0.0) x += 2.0 * M_PI;
                    x -= M_PI;
                }
            }
        }
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}

}  // extern "C"

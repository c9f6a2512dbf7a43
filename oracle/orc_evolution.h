// orc_evolution.h — ORACLE (test infrastructure): the bio2 / bio2_memetic solver.
//
// Restates reference src/ik_evolution_2.cpp:45-657 (IKEvolution2<memetic>) on top of src/ik_base.h:128-210
// (IKBase fitness helpers).  Generalisations, all with the reference value as default:
//   * `lambda` children per species per generation (reference: 16, ik_evolution_2.cpp:138)
//   * `fk_mode` LINEAR = the reference's linearised phenotypes; EXACT = exact FK per child, and the memetic
//     phase re-linearises at the current elite (deviation D2)
//   * RNG back-end as a template parameter (orc_rng.h)
// Reference quirks handled: Q1 stale masked tips (fixed in orc_model.h), Q2 species.fitness
// read uninitialised on the first step (:612) -> +infinity, Q3 secondary goals see null tip frames
// (ik_base.h:163: uninitialised in the reference; zero frames here), Q4 std::sort tie order (:376, :617)
// -> stable order, Q5 the line search's step on a flat model: all three support values equal make the quadratic step 0 / 0 (:503-507) and the linear one a division
// by 0 (:547-548); the candidate's genes are NaN, RobotInfo::clip lets a NaN through (utils.h:328-333: two comparisons that are both false), and a goal that takes
// max(0, .) of its error hides it -- the NaN candidate can be ACCEPTED and returned as the solution (found by tools/robot_fuzz_hostsim.py: 3 of 600 random robots).
// The device takes a candidate with a NaN gene for no candidate and stops the search there -- what the reference does whenever the NaN is NOT hidden (its fitness is
// NaN and fails the comparison): quirk_mode 0 (the default, what the device is compared with) does the same, quirk_mode 1 is the literal reference.
// Q7 (round 6) the same line search on a model without curvature but with a slope: v / 0 is an INFINITE step, RobotInfo::clip puts a joint WITHOUT limits at
// +-DBL_MAX (robot_info.h:109-113: clip_max = DBL_MAX), the linear model is evaluated at 1.8e308 and -- where a goal hides the overflow, e.g. the angle of a
// ConeGoal: acos(NaN) under max(0, .) -- the candidate is accepted: the reference can return a joint value of 1.8e308 (tools/robot_fuzz_hostsim.py: 4 of 8000 random
// robots).  The device takes a candidate with a gene of magnitude >= 1e300 for no candidate (BIOIK_CANDIDATE_BOUND); quirk_mode 0 does the same, mode 1 is literal.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <vector>

#include "orc_gradient.h"
#include "orc_problem.h"
#include "orc_rng.h"

#ifndef ORC_CANDIDATE_BOUND
#define ORC_CANDIDATE_BOUND 1e300  // (bioik_kernels.h: BIOIK_CANDIDATE_BOUND)
#endif

namespace orc {

struct Individual {
    std::vector<double> genes, gradients;
    double fitness = 0;
    size_t order = 0;  // original child index, for stable ordering
};
struct Species {
    std::vector<Individual> individuals;
    double fitness = std::numeric_limits<double>::infinity();
    bool improved = false;
    uint32_t id = 0;
};

template <class Rng>
struct Evolution2 {
    const Problem* problem;
    const Model* model;
    RobotFK fk;
    Rng rng;
    int memetic;  // 0, 'q', 'l'
    int fk_mode;
    size_t lambda;
    bool no_wipeout;
    double dpos, drot, dtwist;
    Query query;
    std::vector<double> initial_guess, solution, temp_joint_variables;
    double solution_fitness = 0;
    std::vector<Species> species;
    std::vector<Individual> children;
    std::vector<Frame> phenotypes, phenotypes2, phenotypes3, null_tip_frames;
    std::vector<size_t> child_indices, quaternion_genes;
    std::vector<const double*> genotypes;
    std::vector<double> genes_min, genes_max, genes_span, gradient, temp, temp_active;
    uint32_t step_index = 0;
    volatile int canceled = 0;

    Evolution2(const Problem* p, Rng r, const bioik_solve_params& sp)
        : problem(p), model(p->model), fk(p->model), rng(r) {
        memetic = sp.mode == BIOIK_MODE_BIO2 ? 0 : (sp.mode == BIOIK_MODE_BIO2_MEMETIC_L ? 'l' : 'q');
        fk_mode = sp.fk_mode;
        lambda = sp.population > 0 ? (size_t)sp.population : 16;
        no_wipeout = sp.no_wipeout != 0;
        dpos = normalize_threshold(sp.dpos);
        drot = normalize_threshold(sp.drot);
        dtwist = normalize_threshold(sp.dtwist);
    }

    size_t D() const { return problem->active_variables.size(); }
    size_t T() const { return problem->tip_link_indices.size(); }

    // ---- ik_base.h:163-207 ----
    double secondary_fitness(const double* genes) { return problem->compute_goal_fitness(problem->secondary_goals, query, null_tip_frames.data(), genes); }
    // (quirk Q5, see the header: the literal clip lets a NaN through; the default mode marks the candidate instead -- it is then no candidate, as on the device)
    bool candidate_has_nan = false;
    double line_search_clip(double p, size_t var) {
        if (quirk_mode() == 0 && p != p) {
            candidate_has_nan = true;
            return p;
        }
        const double c = model->clip(p, var);
        // (quirk Q7: a step without bound clips a joint WITHOUT limits to +-DBL_MAX, robot_info.h:109-113; the default mode takes a candidate with a gene of
        // magnitude 1e300 or more for no candidate, as the device does -- mode 1 keeps it and evaluates the linear model there, as the reference does)
        if (std::fabs(c) >= ORC_CANDIDATE_BOUND) {
            unbounded_candidates()++;  // (diagnostics: how often a line search met such a candidate, in either mode)
            if (quirk_mode() == 0) candidate_has_nan = true;
        }
        return c;
    }
    double primary_fitness(const Frame* frames, const double* genes) { return problem->compute_goal_fitness(problem->goals, query, frames, genes); }
    double combined_fitness(const Frame* frames, const double* genes) {
        double ret = 0.0;
        ret += problem->compute_goal_fitness(problem->goals, query, frames, genes);
        ret += problem->compute_goal_fitness(problem->secondary_goals, query, null_tip_frames.data(), genes);
        return ret;
    }
    const double* extract_active(const std::vector<double>& vars) {
        temp_active.resize(D());
        for (size_t i = 0; i < D(); i++) temp_active[i] = vars[problem->active_variables[i]];
        return temp_active.data();
    }
    double compute_fitness_exact(const std::vector<double>& vars) {  // ik_base.h:203-207
        fk.apply_configuration(vars);
        return primary_fitness(fk.tip_frames.data(), extract_active(vars));
    }
    void genes_to_joint_variables(const Individual& ind, std::vector<double>& vars) {  // :101-107
        vars.resize(model->vars.size());
        for (size_t i = 0; i < D(); i++) vars[problem->active_variables[i]] = ind.genes[i];
    }

    // ---- ik_evolution_2.cpp:111-230 ----
    void initialize(const Query& q) {
        query = q;
        fk.initialize(problem->tip_link_indices);
        null_tip_frames.assign(T(), Frame{{0, 0, 0}, {0, 0, 0, 0}});
        quaternion_genes.clear();
        for (size_t igene = 0; igene < D(); igene++) {
            size_t ivar = problem->active_variables[igene];
            const Link& jl = model->links[model->var_joint[ivar]];
            if ((size_t)jl.first_var + 3 != ivar) continue;
            if (jl.type != BIOIK_JOINT_FLOATING) continue;
            quaternion_genes.push_back(igene);
        }
        initial_guess.assign(q.initial_guess, q.initial_guess + model->vars.size());
        solution = initial_guess;
        solution_fitness = compute_fitness_exact(solution);
        temp_joint_variables = initial_guess;
        size_t population_size = 2;
        species.assign(2, Species());
        uint32_t sid = 0;
        for (auto& s : species) {
            s.id = sid++;
            s.individuals.resize(population_size);
            auto& v = s.individuals[0];
            v.genes.resize(D());
            for (size_t i = 0; i < D(); i++) v.genes[i] = initial_guess[problem->active_variables[i]];
            v.gradients.assign(D(), 0);
            for (size_t i = 1; i < s.individuals.size(); i++) s.individuals[i] = s.individuals[0];
        }
        children.assign(population_size + lambda, Individual());
        for (auto& c : children) {
            c.genes.resize(D());
            c.gradients.resize(D());
        }
        genes_min.resize(D());
        genes_max.resize(D());
        genes_span.resize(D());
        for (size_t i = 0; i < D(); i++) {
            const VarInfo& info = model->vars[problem->active_variables[i]];
            genes_min[i] = info.clip_min;
            genes_max[i] = info.clip_max;
            genes_span[i] = info.span;
        }
        step_index = 0;
        canceled = 0;
    }

    // ---- ik_evolution_2.cpp:242-326 ----
    void reproduce(const std::vector<Individual>& population) {
        size_t gene_count = D();
        rng.reproduce_begin(children.size(), gene_count);
        for (size_t child_index = population.size(); child_index < children.size(); child_index++) {
            double mutation_rate = (double)(1 << rng.rate_exponent(child_index)) * (1.0 / (1 << 23));
            const Individual& parent = population[0];
            const Individual& parent2 = population[1];
            double fmix = (child_index % 2 == 0) * 0.2;
            double gradient_factor = (double)(child_index % 3);
            Individual& child = children[child_index];
            for (size_t gi = 0; gi < gene_count; gi++) {
                double r = rng.gauss(child_index, gi);
                double f = mutation_rate * genes_span[gi];
                double gene = parent.genes[gi];
                double parent_gene = gene;
                gene += r * f;
                double parent_gradient = mix(parent.gradients[gi], parent2.gradients[gi], fmix);
                double grad = parent_gradient * gradient_factor;
                gene += grad;
                gene = clamp(gene, genes_min[gi], genes_max[gi]);
                child.genes[gi] = gene;
                child.gradients[gi] = mix(parent_gradient, gene - parent_gene, 0.3);
            }
            rng.child_end(gene_count);
            for (size_t qg : quaternion_genes) {  // :320-324
                Quat qq = {child.genes[qg], child.genes[qg + 1], child.genes[qg + 2], child.genes[qg + 3]};
                normalize_fast(qq);
                child.genes[qg] = qq.x;
                child.genes[qg + 1] = qq.y;
                child.genes[qg + 2] = qq.z;
                child.genes[qg + 3] = qq.w;
            }
            child.order = child_index;
        }
    }

    // phenotype of n genotypes -> out[n*T]; LINEAR: forward_kinematics.h:1175 ; EXACT: :331
    void phenotypes_of(size_t n, const double* const* geno, std::vector<Frame>& out, int mode) {
        out.resize(n * T());
        if (mode == BIOIK_FK_LINEAR) {
            fk.compute_approximate_mutations(n, geno, out.data());
        } else {
            // exact chain walk per individual on a scratch FK so that the approximator's base is untouched
            static thread_local std::vector<double> vars;
            RobotFK& x = exact_fk();
            vars = temp_joint_variables;
            for (size_t m = 0; m < n; m++) {
                for (size_t i = 0; i < D(); i++) vars[problem->active_variables[i]] = geno[m][i];
                x.apply_configuration(vars);
                for (size_t t = 0; t < T(); t++) out[m * T() + t] = x.tip_frames[t];
            }
        }
    }
    std::vector<RobotFK> exact_fk_storage;
    RobotFK& exact_fk() {
        if (exact_fk_storage.empty()) {
            exact_fk_storage.emplace_back(model);
            exact_fk_storage[0].initialize(problem->tip_link_indices);
        }
        return exact_fk_storage[0];
    }

    // ---- ik_evolution_2.cpp:328-646 ----
    void step() {
        for (size_t ispecies = 0; ispecies < species.size(); ispecies++) {
            Species& sp = species[ispecies];
            auto& population = sp.individuals;
            // :341-346
            genes_to_joint_variables(population[0], temp_joint_variables);
            fk.apply_configuration(temp_joint_variables);
            fk.initialize_mutation_approximator(problem->active_variables);

            size_t generation_count = 16;
            if (memetic) generation_count = 8;
            for (size_t generation = 0; generation < generation_count; generation++) {
                if (canceled) break;
                rng.set_context(step_index, (uint32_t)generation, sp.id);
                reproduce(population);
                size_t child_count = children.size();
                // pre-selection by secondary objectives, :366-378
                if (problem->secondary_goals.size()) {
                    child_count = rng.preselect_count(population.size(), lambda);
                    for (size_t ci = population.size(); ci < children.size(); ci++)
                        children[ci].fitness = secondary_fitness(children[ci].genes.data());
                    if (quirk_mode() == 1)
                        std::sort(children.begin() + population.size(), children.end(),
                                  [](const Individual& a, const Individual& b) { return a.fitness < b.fitness; });  // :376 verbatim
                    else
                        std::stable_sort(children.begin() + population.size(), children.end(),
                                         [](const Individual& a, const Individual& b) { return a.fitness < b.fitness; });
                }
                // keep parents, :381-388
                for (size_t i = 0; i < population.size(); i++) {
                    children[i].genes = population[i].genes;
                    children[i].gradients = population[i].gradients;
                }
                // genotype-phenotype mapping, :391-398
                genotypes.resize(child_count);
                for (size_t i = 0; i < child_count; i++) genotypes[i] = children[i].genes.data();
                phenotypes_of(child_count, genotypes.data(), phenotypes, fk_mode);
                // fitness, :401-407
                for (size_t ci = 0; ci < child_count; ci++)
                    children[ci].fitness = primary_fitness(&phenotypes[ci * T()], genotypes[ci]);
                // selection, :410-431
                child_indices.resize(child_count);
                for (size_t i = 0; i < child_count; i++) child_indices[i] = i;
                for (size_t i = 0; i < population.size(); i++) {
                    size_t jmin = i;
                    double fmin = children[child_indices[i]].fitness;
                    for (size_t j = i + 1; j < child_count; j++) {
                        double f = children[child_indices[j]].fitness;
                        if (f < fmin) jmin = j, fmin = f;
                    }
                    std::swap(child_indices[i], child_indices[jmin]);
                }
                for (size_t i = 0; i < population.size(); i++) {
                    std::swap(population[i].genes, children[child_indices[i]].genes);
                    std::swap(population[i].gradients, children[child_indices[i]].gradients);
                }
            }

            // memetic optimisation, :436-570
            if (memetic == 'q' || memetic == 'l') {
                Individual& individual = population[0];
                if (fk_mode == BIOIK_FK_EXACT) {
                    // deviation D2: fresh linearisation at the current elite
                    genes_to_joint_variables(individual, temp_joint_variables);
                    fk.apply_configuration(temp_joint_variables);
                    fk.initialize_mutation_approximator(problem->active_variables);
                }
                gradient.resize(D());
                rng.set_context(step_index, 0, sp.id);
                double dp = 0.0000001;
                if (rng.memetic_negative()) dp = -dp;
                for (size_t generation = 0; generation < 8; generation++) {
                    if (canceled) break;
                    temp = individual.genes;
                    const double* g0 = temp.data();
                    double* gw = temp.data();
                    phenotypes_of(1, &g0, phenotypes2, BIOIK_FK_LINEAR);
                    double f2p = primary_fitness(phenotypes2.data(), g0);
                    double fa = f2p + secondary_fitness(g0);
                    for (size_t i = 0; i < D(); i++) {
                        gw[i] = individual.genes[i] + dp;
                        fk.compute_approximate_mutation1(problem->active_variables[i], +dp, phenotypes2, phenotypes3);
                        double fb = combined_fitness(phenotypes3.data(), g0);
                        gw[i] = individual.genes[i];
                        gradient[i] = fb - fa;
                    }
                    // normalise gradient, :477-482
                    double sum = dp * dp;
                    for (size_t i = 0; i < D(); i++) sum += std::fabs(gradient[i]);
                    double f = 1.0 / sum * dp;
                    for (size_t i = 0; i < D(); i++) gradient[i] *= f;
                    // support points, :485-495
                    for (size_t i = 0; i < D(); i++) gw[i] = individual.genes[i] - gradient[i];
                    phenotypes_of(1, &g0, phenotypes3, BIOIK_FK_LINEAR);
                    double f1 = combined_fitness(phenotypes3.data(), g0);
                    double f2 = fa;
                    for (size_t i = 0; i < D(); i++) gw[i] = individual.genes[i] + gradient[i];
                    phenotypes_of(1, &g0, phenotypes3, BIOIK_FK_LINEAR);
                    double f3 = combined_fitness(phenotypes3.data(), g0);
                    if (memetic == 'q') {  // :498-539
                        double v1 = (f2 - f1);
                        double v2 = (f3 - f2);
                        double v = (v1 + v2) * 0.5;
                        double a = (v1 - v2);
                        double step_size = v / a;
                        candidate_has_nan = false;
                        for (size_t i = 0; i < D(); i++)
                            gw[i] = line_search_clip(individual.genes[i] + gradient[i] * step_size, problem->active_variables[i]);
                        if (candidate_has_nan) break;  // (Q5, default mode)
                        phenotypes_of(1, &g0, phenotypes2, BIOIK_FK_LINEAR);
                        double f4p = primary_fitness(phenotypes2.data(), g0);
                        if (f4p < f2p) {
                            individual.genes = temp;
                            continue;
                        } else {
                            break;
                        }
                    }
                    if (memetic == 'l') {  // :545-568
                        double cost_diff = (f3 - f1) * 0.5;
                        double step_size = f2 / cost_diff;
                        candidate_has_nan = false;
                        for (size_t i = 0; i < D(); i++)
                            gw[i] = line_search_clip(individual.genes[i] - gradient[i] * step_size, problem->active_variables[i]);
                        if (candidate_has_nan) break;  // (Q5, default mode)
                        phenotypes_of(1, &g0, phenotypes2, BIOIK_FK_LINEAR);
                        double f4p = primary_fitness(phenotypes2.data(), g0);
                        if (f4p < f2p) {
                            individual.genes = temp;
                            continue;
                        } else {
                            break;
                        }
                    }
                }
            }
        }

        // species management, :604-645
        for (auto& sp : species) {
            genes_to_joint_variables(sp.individuals[0], temp_joint_variables);
            double fitness = compute_fitness_exact(temp_joint_variables);
            sp.improved = (fitness != sp.fitness);
            sp.fitness = fitness;
        }
        std::stable_sort(species.begin(), species.end(), [](const Species& a, const Species& b) { return a.fitness < b.fitness; });
        for (size_t si = 1; si < species.size(); si++) {
            rng.set_context(step_index, 0, species[si].id);
            bool wipe = rng.wipeout_u() < 0.1;  // evaluated first, as in `fast_random() < 0.1 || !improved`
            wipe = wipe || !species[si].improved;
            if (no_wipeout) wipe = false;
            if (wipe) {
                Individual& ind = species[si].individuals[0];
                for (size_t i = 0; i < ind.genes.size(); i++) {
                    const VarInfo& info = model->vars[problem->active_variables[i]];
                    ind.genes[i] = rng.wipeout_gene(i, info.min, info.max);
                }
                for (auto& v : ind.gradients) v = 0;
                for (size_t i = 0; i < species[si].individuals.size(); i++) species[si].individuals[i] = species[si].individuals[0];
            }
        }
        if (species[0].fitness < solution_fitness) {
            genes_to_joint_variables(species[0].individuals[0], solution);
            solution_fitness = species[0].fitness;
        }
        step_index++;
    }

    const std::vector<double>& get_solution() const { return solution; }

    // ik_parallel.h:173-181: exact FK of getSolution(), checkSolution, computeFitness
    void check(bool& success, double& fitness) {
        fk.apply_configuration(solution);
        const double* act = extract_active(solution);
        success = problem->check_solution(query, fk.tip_frames.data(), act, dpos, drot, dtwist);
        fitness = primary_fitness(fk.tip_frames.data(), act);
    }
};

struct IslandResult {
    std::vector<double> solution;
    bool success = false;
    double fitness = DBL_MAX;
    int steps = 0;
};

// one island of ik_parallel.h:148-190; budget mode (timeout_s<=0): check after every step, at most max_steps steps;
// wall-clock mode: the reference's loop (1 step + up to 3 more while time remains, then check).  Any solver with
// initialize(q) / step() / check(success, fitness) / get_solution().
template <class Solver>
IslandResult run_island_loop(Solver& ik, size_t n_vars, const bioik_solve_params& sp, const Query& q, double timeout_s) {
    using clock = std::chrono::steady_clock;
    IslandResult r;
    r.solution.assign(q.initial_guess, q.initial_guess + n_vars);
    ik.initialize(q);
    auto t_end = clock::now() + std::chrono::duration_cast<clock::duration>(std::chrono::duration<double>(timeout_s > 0 ? timeout_s : 0));
    bool wall = timeout_s > 0;
    for (size_t iteration = 0;; iteration++) {
        if (wall) {
            if (!(clock::now() < t_end) && iteration != 0) break;
        } else {
            if (r.steps >= sp.max_steps) break;
        }
        ik.step();
        r.steps++;
        if (wall)
            for (int it2 = 1; it2 < 4; it2++)
                if (clock::now() < t_end) {
                    ik.step();
                    r.steps++;
                }
        bool success;
        double fitness;
        ik.check(success, fitness);
        r.success = success;
        r.solution = ik.get_solution();
        r.fitness = fitness;
        if (success) break;
    }
    return r;
}

// the solver behind an IKFactory name (src/ik_evolution_2.cpp:652-654, src/ik_gradient.cpp:254-292)
inline int gradient_if_stuck(int mode) { return mode == BIOIK_MODE_GD ? ' ' : (mode == BIOIK_MODE_GD_R ? 'r' : 'c'); }
// `island`: the solver thread this island stands for (ik_parallel.h:127: thread_index)
template <class Rng>
IslandResult run_island(const Problem* problem, Rng rng, const bioik_solve_params& sp, const Query& q, double timeout_s, int island = 0) {
    const size_t nv = problem->model->vars.size();
    if (sp.mode == BIOIK_MODE_GD_C || sp.mode == BIOIK_MODE_GD || sp.mode == BIOIK_MODE_GD_R) {
        GradientDescent<Rng> ik(problem, sp, gradient_if_stuck(sp.mode), rng, island);
        return run_island_loop(ik, nv, sp, q, timeout_s);
    }
    if (sp.mode == BIOIK_MODE_JAC) {
        JacobianSolver<Rng> ik(problem, sp, rng, island);
        return run_island_loop(ik, nv, sp, q, timeout_s);
    }
    Evolution2<Rng> ik(problem, rng, sp);
    return run_island_loop(ik, nv, sp, q, timeout_s);
}

}  // namespace orc

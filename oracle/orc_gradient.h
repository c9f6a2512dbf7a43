// orc_gradient.h — ORACLE (test infrastructure): the gradient-descent and pseudo-inverse-Jacobian solvers.
//
// Restates reference src/ik_gradient.cpp:136-251 (IKGradientDescent<if_stuck, threads>, factory names gd / gd_c / gd_r and their _2 / _4 /
// _8 forms) and :42-133, 269-292 (IKJacobianBase / IKJacobian, factory names jac, jac_2 ... jac_8).  Solver thread 0 starts at the seed,
// the further threads of the _N forms at random configurations (:157-159, :283-285) -- here: the islands 1 ... N - 1 of a query --, and
// gd_r replaces a configuration that a step failed to improve by a random one (:165-170, :233-237).  The random numbers come from
// the solver's generator: the reference's own (ReferenceRandom, whose clones share one state: its threads 1 ... N - 1 all start at
// the SAME point) or the counter-based one the device uses (one stream per query and island).
//
// The reference's `jac` solves its least-squares step with Eigen::JacobiSVD (ik_gradient.cpp:117); Eigen is a third-party library that
// is absent here, so the step is restated from the published definition — the minimum-norm least-squares solution through the
// singular value decomposition, singular values below epsilon * min(rows, cols) * sigma_max treated as zero (Eigen's default
// threshold) — with a one-sided Jacobi SVD (Hestenes).  The solution is unique, the path to it is not: parity with Eigen would be to
// rounding, not to the bit; against the reference's own sources compiled with the stand-in Eigen of oracle/ref_shim it is exact.
#pragma once
#include <cmath>
#include <vector>

#include "orc_problem.h"
#include "orc_rng.h"

namespace orc {

// min-norm least-squares solution x of J x = b, J is rows x cols row-major; the loop order below is part of the definition (the
// device kernel runs the same sequence of operations)
inline void pinv_solve(const double* J, int rows, int cols, const double* b, double* x) {
    // work on A = J (rows >= cols) or A = J^T (rows < cols): p x q with p >= q, columns of A are orthogonalised
    const bool transposed = rows < cols;
    const int p = transposed ? cols : rows, q = transposed ? rows : cols;
    std::vector<double> A((size_t)p * q), V((size_t)q * q, 0.0);
    for (int i = 0; i < p; i++)
        for (int j = 0; j < q; j++) A[(size_t)i * q + j] = transposed ? J[(size_t)j * cols + i] : J[(size_t)i * cols + j];
    for (int j = 0; j < q; j++) V[(size_t)j * q + j] = 1.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
        for (int i = 0; i < q - 1; i++)
            for (int j = i + 1; j < q; j++) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int k = 0; k < p; k++) {
                    const double ai = A[(size_t)k * q + i], aj = A[(size_t)k * q + j];
                    alpha += ai * ai, beta += aj * aj, gamma += ai * aj;
                }
                if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < p; k++) {
                    const double ai = A[(size_t)k * q + i], aj = A[(size_t)k * q + j];
                    A[(size_t)k * q + i] = c * ai - s * aj;
                    A[(size_t)k * q + j] = s * ai + c * aj;
                }
                for (int k = 0; k < q; k++) {
                    const double vi = V[(size_t)k * q + i], vj = V[(size_t)k * q + j];
                    V[(size_t)k * q + i] = c * vi - s * vj;
                    V[(size_t)k * q + j] = s * vi + c * vj;
                }
            }
        if (!rotated) break;
    }
    // A = U S, so A_k = sigma_k u_k.  x = sum_k v_k (u_k . b) / sigma_k  (not transposed), or with the roles of U and V swapped
    std::vector<double> sigma(q);
    double smax = 0.0;
    for (int k = 0; k < q; k++) {
        double n2 = 0.0;
        for (int i = 0; i < p; i++) n2 += A[(size_t)i * q + k] * A[(size_t)i * q + k];
        sigma[k] = std::sqrt(n2);
        if (sigma[k] > smax) smax = sigma[k];
    }
    const double threshold = 2.220446049250313e-16 * (double)(rows < cols ? rows : cols) * smax;  // Eigen: epsilon * diagSize = min(rows, cols)
    for (int i = 0; i < cols; i++) x[i] = 0.0;
    for (int k = 0; k < q; k++) {
        if (!(sigma[k] > threshold)) continue;
        const double inv = 1.0 / (sigma[k] * sigma[k]);
        if (!transposed) {  // J = U S V^T: u_k = A_k / sigma_k (length rows), v_k = column k of V (length cols)
            double d = 0.0;
            for (int i = 0; i < rows; i++) d += A[(size_t)i * q + k] * b[i];
            d *= inv;
            for (int i = 0; i < cols; i++) x[i] += V[(size_t)i * q + k] * d;
        } else {  // J^T = U S V^T, J = V S U^T: x = sum_k u_k (v_k . b) / sigma_k, u_k = A_k / sigma_k (length cols), v_k of length rows
            double d = 0.0;
            for (int i = 0; i < rows; i++) d += V[(size_t)i * q + k] * b[i];
            d *= inv;
            for (int i = 0; i < cols; i++) x[i] += A[(size_t)i * q + k] * d;
        }
    }
}

struct PointSolverBase {
    const Problem* problem;
    const Model* model;
    RobotFK fk;
    double dpos, drot, dtwist;
    Query query;
    std::vector<double> temp_active;
    PointSolverBase(const Problem* p, const bioik_solve_params& sp) : problem(p), model(p->model), fk(p->model) {
        dpos = normalize_threshold(sp.dpos);
        drot = normalize_threshold(sp.drot);
        dtwist = normalize_threshold(sp.dtwist);
    }
    size_t D() const { return problem->active_variables.size(); }
    const double* extract_active(const std::vector<double>& vars) {
        temp_active.resize(D());
        for (size_t i = 0; i < D(); i++) temp_active[i] = vars[problem->active_variables[i]];
        return temp_active.data();
    }
    double compute_fitness(const std::vector<double>& vars) {  // ik_base.h:203-207
        fk.apply_configuration(vars);
        return problem->compute_goal_fitness(problem->goals, query, fk.tip_frames.data(), extract_active(vars));
    }
    void check_vars(const std::vector<double>& vars, bool& success, double& fitness) {  // ik_parallel.h:173-181
        fk.apply_configuration(vars);
        const double* act = extract_active(vars);
        success = problem->check_solution(query, fk.tip_frames.data(), act, dpos, drot, dtwist);
        fitness = problem->compute_goal_fitness(problem->goals, query, fk.tip_frames.data(), act);
    }
};

// random(modelInfo.getMin(vi), modelInfo.getMax(vi)) for every active variable, in their order (:157-159, :165-170, :283-285)
inline void randomize_configuration(ReferenceRandom& rng, const Problem* problem, uint32_t /*count*/, std::vector<double>& x) {
    for (size_t ivar : problem->active_variables) x[ivar] = rng.random(problem->model->vars[ivar].min, problem->model->vars[ivar].max);
}
inline void randomize_configuration(CounterRandom& rng, const Problem* problem, uint32_t count, std::vector<double>& x) {
    for (size_t g = 0; g < problem->active_variables.size(); g++) {
        const size_t ivar = problem->active_variables[g];
        x[ivar] = rng.point_random(g, count, problem->model->vars[ivar].min, problem->model->vars[ivar].max);
    }
}

// ik_gradient.cpp:136-251; if_stuck: ' ' (keep the best), 'c' (always continue), 'r' (random restart)
template <class Rng>
struct GradientDescent : PointSolverBase {
    int if_stuck;
    Rng rng;
    int thread_index;
    bool reset = false;
    uint32_t steps_done = 0;
    std::vector<double> solution, best_solution, gradient, temp;
    GradientDescent(const Problem* p, const bioik_solve_params& sp, int stuck, Rng r = Rng(), int thread = 0)
        : PointSolverBase(p, sp), if_stuck(stuck), rng(r), thread_index(thread) {}
    void initialize(const Query& q) {  // :147-158
        query = q;
        fk.initialize(problem->tip_link_indices);
        solution.assign(q.initial_guess, q.initial_guess + model->vars.size());
        if (thread_index > 0) randomize_configuration(rng, problem, 0, solution);
        best_solution = solution;
        reset = false, steps_done = 0;
    }
    const std::vector<double>& get_solution() const { return best_solution; }
    void step() {  // :162-247
        if (reset) {  // random reset if stuck (:165-170)
            reset = false;
            randomize_configuration(rng, problem, steps_done + 1, solution);
        }
        temp = solution;
        const double jd = 0.0001;
        gradient.assign(solution.size(), 0.0);
        for (size_t ivar : problem->active_variables) {
            temp[ivar] = solution[ivar] - jd;
            double p1 = compute_fitness(temp);
            temp[ivar] = solution[ivar] + jd;
            double p3 = compute_fitness(temp);
            temp[ivar] = solution[ivar];
            gradient[ivar] = p3 - p1;
        }
        double sum = 0.0001;
        for (size_t ivar : problem->active_variables) sum += std::fabs(gradient[ivar]);
        double f = 1.0 / sum * jd;
        for (size_t ivar : problem->active_variables) gradient[ivar] *= f;
        temp = solution;
        for (size_t ivar : problem->active_variables) temp[ivar] = solution[ivar] - gradient[ivar];
        double p1 = compute_fitness(temp);
        for (size_t ivar : problem->active_variables) temp[ivar] = solution[ivar] + gradient[ivar];
        double p3 = compute_fitness(temp);
        double p2 = (p1 + p3) * 0.5;
        double cost_diff = (p3 - p1) * 0.5;
        double joint_diff = p2 / cost_diff;
        if (!std::isfinite(joint_diff)) joint_diff = 0.0;
        for (size_t ivar : problem->active_variables) temp[ivar] = model->clip(solution[ivar] - gradient[ivar] * joint_diff, ivar);
        if (if_stuck == 'c') {
            solution = temp;
        } else if (compute_fitness(temp) < compute_fitness(solution)) {
            solution = temp;
        } else if (if_stuck == 'r') {
            reset = true;  // (:233-237)
        }
        if (compute_fitness(solution) < compute_fitness(best_solution)) best_solution = solution;
        steps_done++;
    }
    void check(bool& success, double& fitness) { check_vars(best_solution, success, fitness); }
};

// ik_gradient.cpp:42-133, 269-292
template <class Rng>
struct JacobianSolver : PointSolverBase {
    Rng rng;
    int thread_index;
    std::vector<double> solution, tip_diffs, joint_diffs;
    std::vector<Frame> tip_objectives;
    JacobianSolver(const Problem* p, const bioik_solve_params& sp, Rng r = Rng(), int thread = 0) : PointSolverBase(p, sp), rng(r), thread_index(thread) {}
    void initialize(const Query& q) {  // :61-67, :278-286
        query = q;
        fk.initialize(problem->tip_link_indices);
        // goal.frame of problem.cpp:152-174: identity, with the position / orientation of Position / Orientation / Pose goals
        tip_objectives.assign(problem->tip_link_indices.size(), identity_frame());
        for (const GoalInfo& g : problem->goals) {
            const double* P = q.params + g.param_offset;
            Frame f = identity_frame();
            if (g.type == BIOIK_GOAL_POSITION) f.pos = {P[0], P[1], P[2]};
            if (g.type == BIOIK_GOAL_ORIENTATION) f.rot = {P[0], P[1], P[2], P[3]};
            if (g.type == BIOIK_GOAL_POSE) f.pos = {P[0], P[1], P[2]}, f.rot = {P[3], P[4], P[5], P[6]};
            tip_objectives[g.tip_index >= 0 ? (size_t)g.tip_index : 0] = f;  // (goal_info.tip_index = 0 for goals without a link, problem.cpp:155)
        }
        solution.assign(q.initial_guess, q.initial_guess + model->vars.size());
        if (thread_index > 0) randomize_configuration(rng, problem, 0, solution);
    }
    const std::vector<double>& get_solution() const { return solution; }
    void step() {  // optimizeJacobian, :69-132 (translational_scale = rotational_scale = 1)
        const size_t tip_count = problem->tip_link_indices.size(), cols = D();
        tip_diffs.resize(tip_count * 6);
        joint_diffs.resize(cols);
        fk.apply_configuration(solution);
        for (size_t itip = 0; itip < tip_count; itip++) frame_twist(fk.tip_frames[itip], tip_objectives[itip], &tip_diffs[itip * 6]);
        fk.compute_jacobian(problem->active_variables);
        pinv_solve(fk.approx_jacobian.data(), (int)(tip_count * 6), (int)cols, tip_diffs.data(), joint_diffs.data());
        size_t icol = 0;
        for (size_t ivar : problem->active_variables) {
            double v = solution[ivar] + joint_diffs[icol];
            if (!std::isfinite(v)) continue;  // (:126: the column index is not advanced, as in the reference)
            v = model->clip(v, ivar);
            solution[ivar] = v;
            icol++;
        }
    }
    void check(bool& success, double& fitness) { check_vars(solution, success, fitness); }
};

}  // namespace orc

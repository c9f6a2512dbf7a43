// bio_ik/goal_eval.h — goals evaluated on the HOST: the counterpart of the reference's `Problem` for `Goal::describe` /
// `Goal::evaluate` (src/problem.cpp:72-228 builds the GoalContexts, :244-257 sums weight^2 * evaluate).
//
// The MI355X solver evaluates the built-in goals inside its kernels; this class is what runs every goal through its virtual
// `evaluate()` — built-in or user-defined — at ONE configuration: to score or filter returned solutions with goals that have no
// device opcode (JointFunctionGoal, LinkFunctionGoal, a user's subclass), and for the tests that hold the kernels' goal costs against
// the closed forms of bio_ik/goal_types.h.  It is not a solver; the plugin core (bio_ik/plugin_core.h) uses it to re-score the candidates of a
// device search with the goals the device cannot evaluate (the hybrid path for callback goals).
#pragma once
#include <functional>
#include <stdexcept>

#include "goal.h"

namespace bio_ik {

class HostGoalProblem {
public:
    struct Model {  // what is needed of a robot model, whichever type the caller has (moveit::core::RobotModel, bio_ik::RobotModel ...)
        RobotInfo info;                                                                  // one entry per robot variable
        std::function<int(const std::string&)> variable_index;                           // robot variable by name, -1: unknown
        std::function<Frame(const std::string&, const std::vector<double>&)> link_frame;  // global frame of a link at a full variable vector
    };

private:
    struct Entry {
        const Goal* goal;
        GoalContext context;
        double weight_sq;
    };
    Model model_;
    std::vector<Entry> entries_;
    std::vector<std::string> tip_names_;  // the problem's tip links, in the order the goals name them (problem.cpp:144-160)
    std::vector<size_t> active_;          // robot variable of every active variable (gene order)
    mutable std::vector<Frame> frames_;
    mutable std::vector<double> genes_;

public:
    // active_variables: robot variable index per gene (bioik_problem_active_variables); initial_guess: full variable vector (the seed)
    HostGoalProblem(const Model& model, const std::vector<const Goal*>& goals, const std::vector<int>& active_variables, const std::vector<double>& initial_guess)
        : model_(model) {
        for (int v : active_variables) active_.push_back((size_t)v);
        double rcp_sum = 0.0;  // velocity weights (problem.cpp:206-225)
        for (size_t v : active_) rcp_sum += model_.info.getMaxVelocityRcp(v);
        std::vector<double> weights;
        for (size_t v : active_) weights.push_back(rcp_sum > 0 ? model_.info.getMaxVelocityRcp(v) / rcp_sum : 1.0 / active_.size());
        std::vector<size_t> tip_indices;
        entries_.reserve(goals.size());
        for (const Goal* g : goals) {
            entries_.push_back(Entry{g, GoalContext(), 0.0});
            GoalContext& c = entries_.back().context;
            g->describe(c);
            entries_.back().weight_sq = c.goal_weight_ * c.goal_weight_;
            for (auto& name : c.goal_link_names_) {
                size_t t = 0;
                while (t < tip_names_.size() && tip_names_[t] != name) t++;
                if (t == tip_names_.size()) tip_names_.push_back(name);
                c.goal_link_indices_.push_back(t);
            }
            for (auto& name : c.goal_variable_names_) {
                const int ivar = model_.variable_index(name);
                if (ivar < 0) throw std::runtime_error("joint variable not found: " + name);  // problem.cpp:125
                ssize_t idx = -1 - (ssize_t)ivar;  // a variable that is not active reads the initial guess (goal.h:70-77)
                for (size_t i = 0; i < active_.size(); i++)
                    if (active_[i] == (size_t)ivar) idx = (ssize_t)i;
                c.goal_variable_indices_.push_back(idx);
            }
            c.problem_active_variables_ = active_;
            c.initial_guess_ = initial_guess;
            c.velocity_weights_ = weights;
            c.robot_info_ = &model_.info;
        }
        for (size_t t = 0; t < tip_names_.size(); t++) tip_indices.push_back(t);
        for (auto& e : entries_) e.context.problem_tip_link_indices_ = tip_indices;
        frames_.resize(tip_names_.size());
    }
    const std::vector<std::string>& getTipNames() const { return tip_names_; }
    // another query of the same goal structure: what goals read of variables that are not active comes from its seed (goal.h:70-77)
    void setInitialGuess(const std::vector<double>& initial_guess) {
        for (auto& e : entries_) e.context.initial_guess_ = initial_guess;
    }
    size_t goalCount() const { return entries_.size(); }
    bool isSecondary(size_t i) const { return entries_[i].context.goal_secondary_; }
    double weightSq(size_t i) const { return entries_[i].weight_sq; }

    // the goals' unweighted costs at `positions` (a full variable vector), in goal order
    std::vector<double> evaluateGoals(const std::vector<double>& positions) const {
        genes_.resize(active_.size());
        for (size_t i = 0; i < active_.size(); i++) genes_[i] = positions[active_[i]];
        for (size_t t = 0; t < tip_names_.size(); t++) frames_[t] = model_.link_frame(tip_names_[t], positions);
        std::vector<double> out;
        for (auto& e : entries_) {
            GoalContext& c = const_cast<GoalContext&>(e.context);
            c.active_variable_positions_ = genes_.data();
            c.tip_link_frames_ = frames_.data();
            out.push_back(e.goal->evaluate(e.context));
        }
        return out;
    }
    // sum of weight^2 * cost over the primary (secondary = false) or the secondary goals: Problem::computeGoalFitness (problem.cpp:244-257)
    double computeGoalFitness(const std::vector<double>& positions, bool secondary = false) const {
        const std::vector<double> e = evaluateGoals(positions);
        double sum = 0.0;
        for (size_t i = 0; i < entries_.size(); i++)
            if (entries_[i].context.goal_secondary_ == secondary) sum += e[i] * entries_[i].weight_sq;
        return sum;
    }
};

}  // namespace bio_ik

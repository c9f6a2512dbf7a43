// bioik_compile.h — host side of the boundary: flat robot model + problem template -> DevProblem joint program.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bioik_hip.h"
#include "bioik_types.h"

namespace bioik {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct Frame {
    double p[3];
    double q[4];  // x y z w
};

// what RobotJointEvaluator / RobotInfo read from moveit::core::RobotModel
// (reference src/forward_kinematics.h:192-213, include/bio_ik/robot_info.h:70-106)
struct HostModel {
    struct Link {
        int parent, type, first_var, var_count, mimic;
        Frame origin;
        double axis[3];
        double mimic_factor, mimic_offset;
        double mass, center[3];  // urdf inertial (BalanceGoal)
    };
    struct Var {
        double clip_min, clip_max, span, vmin, vmax, max_velocity_rcp;
        int joint;
    };
    std::vector<Link> links;
    std::vector<Var> vars;
    explicit HostModel(const bioik_model_desc& d);
};

// Problem::initialize (reference src/problem.cpp:72-228) + RobotFK::initialize (src/forward_kinematics.h:253-330),
// compiled into the device joint program.
struct HostProblem {
    const HostModel* model;
    std::vector<int> active_variables;  // robot variable per gene
    std::vector<int> tip_links;         // link per public tip
    int param_count = 0;
    DevProblem dev;
    HostProblem(const HostModel* m, const bioik_problem_desc& d);
};

int goal_param_count(int type);
DevSolveParams normalize_params(const bioik_solve_params& p, uint64_t first_query, size_t n_queries = 0, size_t resident_units = 2048);  // resident_units: the (query, island) workgroups of the latency schedule's kernel the device holds at once (eight per CU)  // (n_queries: what BIOIK_ISLANDS_AUTO sizes the islands to; 0: one island)

}  // namespace bioik

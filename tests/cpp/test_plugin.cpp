// C++ driver of the plugin mirror (bio_ik_amd/cpp): FK -> IK -> FK round trip through searchPositionIK /
// searchPositionIKBatch on a PR2-like right arm, error-code conventions, user goals through BioIKKinematicsQueryOptions.
// Linked against libbioik_hip.so on a GPU box, or against the host simulator of the kernels in the CPU suite.
#include <cstdio>
#include <random>

#include <bio_ik/kinematics_plugin.h>

#include "pr2_arm_fixture.h"

using namespace bio_ik_kinematics_plugin;

// the caller's wall-clock budget [s] (honoured on the device, ik_parallel.h:160).  The reference's yaml default is 5 ms; the host
// simulator of the CPU suite executes a step in far more than that and is built with a generous value.
#ifndef TEST_TIMEOUT
#define TEST_TIMEOUT 0.25
#endif

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                   \
        }                                                               \
    } while (0)

struct SecondaryLinkFunctionGoal : bio_ik::LinkFunctionGoal {  // (a callback goal made secondary the way a user of the reference would: in a subclass)
    SecondaryLinkFunctionGoal(const std::string& link, const std::function<double(const bio_ik::Vector3&, const bio_ik::Quaternion&)>& f) : LinkFunctionGoal(link, f) { secondary_ = true; }
};

int main() {
    bio_ik::RobotModel rm = pr2Arm();
    BioIKKinematicsPlugin plugin;
    BioIKParams params;
    params.gpu_population = 16, params.gpu_fk = "linear", params.gpu_max_steps = 60, params.random_seed = 3;
    params.gpu_reproducible_calls = true;  // (the determinism check below; by default the random streams advance from call to call)
#ifdef TEST_ISLANDS
    params.gpu_islands = TEST_ISLANDS;  // (the host simulator's runs: the default -- sixteen islands for a single pose, BIOIK_ISLANDS_AUTO -- takes the fibres a minute)
#endif
    CHECK(plugin.initialize(rm, "right_arm", "torso_lift_link", {"r_wrist_roll_link"}, 0.0, params));
    CHECK(plugin.getJointNames().size() == 7 && plugin.getJointNames()[0] == "r_shoulder_pan_joint");
    CHECK(plugin.getLinkNames().size() == 1 && plugin.supportsGroup(nullptr));
    {
        std::vector<geometry_msgs::Pose> poses;
        CHECK(!plugin.getPositionFK({}, {}, poses));
    }
    // targets: FK of random valid configurations, expressed in the base frame of the group (reference README.md:404-447)
    std::mt19937 rng(5);
    auto uniform = [&](double lo, double hi) { return std::uniform_real_distribution<double>(lo, hi)(rng); };
    const int n = 2;
    std::vector<std::vector<geometry_msgs::Pose>> poses(n);
    std::vector<std::vector<double>> seeds(n);
    std::vector<int> gv;
    for (auto& name : plugin.getJointNames()) gv.push_back(rm.variableIndex(name));
    double base[7], inv[7];
    rm.linkTransform(rm.linkIndex("torso_lift_link"), rm.defaultPositions(), base);
    double iq[4] = {-base[3], -base[4], -base[5], base[6]}, np[3] = {-base[0], -base[1], -base[2]}, ip[3];
    bio_ik::RobotModel::rotate(iq, np, ip);
    for (int c = 0; c < 3; c++) inv[c] = ip[c];
    for (int c = 0; c < 4; c++) inv[3 + c] = iq[c];
    std::vector<std::vector<double>> want(n, std::vector<double>(7));
    for (int k = 0; k < n; k++) {
        std::vector<double> target = rm.defaultPositions();
        for (int v : gv) target[v] = uniform(rm.var_min[v], rm.var_max[v]);
        double tip[7], rel[7];
        rm.linkTransform(rm.linkIndex("r_wrist_roll_link"), target, tip);
        for (int c = 0; c < 7; c++) want[k][c] = tip[c];
        bio_ik::RobotModel::concat(inv, tip, rel);
        geometry_msgs::Pose p;
        p.position.x = rel[0], p.position.y = rel[1], p.position.z = rel[2];
        p.orientation.x = rel[3], p.orientation.y = rel[4], p.orientation.z = rel[5], p.orientation.w = rel[6];
        poses[k].push_back(p);
        for (int v : gv) seeds[k].push_back(std::min(std::max(target[v] + uniform(-0.2, 0.2), rm.var_min[v]), rm.var_max[v]));
    }
    std::vector<std::vector<double>> sols;
    std::vector<moveit_msgs::MoveItErrorCodes> codes;
    CHECK(plugin.searchPositionIKBatch(poses, seeds, sols, codes));
    for (int k = 0; k < n; k++) {
        CHECK(codes[k].val == moveit_msgs::MoveItErrorCodes::SUCCESS && sols[k].size() == 7);
        std::vector<double> st = rm.defaultPositions();
        for (size_t i = 0; i < gv.size(); i++) st[gv[i]] = sols[k][i];
        double got[7];
        rm.linkTransform(rm.linkIndex("r_wrist_roll_link"), st, got);
        double dp = 0, dot = 0;
        for (int c = 0; c < 3; c++) dp += (got[c] - want[k][c]) * (got[c] - want[k][c]);
        for (int c = 3; c < 7; c++) dot += got[c] * want[k][c];
        CHECK(std::sqrt(dp) < 1e-4 && 2 * std::acos(std::min(1.0, std::fabs(dot))) < 1e-3);  // north-star tolerance
    }
    // single-query form + callback semantics (kinematics_plugin.cpp:644-649)
    std::vector<double> solution;
    moveit_msgs::MoveItErrorCodes code;
    CHECK(plugin.searchPositionIK(poses[0][0], seeds[0], TEST_TIMEOUT, solution, code) && code.val == moveit_msgs::MoveItErrorCodes::SUCCESS);
    CHECK(solution == sols[0]);  // deterministic: same query, same stream
    IKCallbackFn reject = [](const geometry_msgs::Pose&, const std::vector<double>&, moveit_msgs::MoveItErrorCodes& e) { e.val = moveit_msgs::MoveItErrorCodes::NO_IK_SOLUTION; };
    CHECK(!plugin.searchPositionIK(poses[0][0], seeds[0], TEST_TIMEOUT, solution, reject, code));
    // unreachable goal: NO_IK_SOLUTION unless approximate solutions are allowed (:638-641); a tiny budget keeps it short
    BioIKKinematicsPlugin quick;
    params.gpu_max_steps = 2;
    CHECK(quick.initialize(rm, "right_arm", "torso_lift_link", {"r_wrist_roll_link"}, 0.0, params));
    geometry_msgs::Pose far;
    far.position.x = far.position.y = far.position.z = 5.0;
    CHECK(!quick.searchPositionIK(far, seeds[0], TEST_TIMEOUT, solution, code) && code.val == moveit_msgs::MoveItErrorCodes::NO_IK_SOLUTION);
    bio_ik::KinematicsQueryOptions approx;
    approx.return_approximate_solution = true;
    CHECK(quick.searchPositionIK(far, seeds[0], TEST_TIMEOUT, solution, code, approx) && solution.size() == 7);
    // user goals replacing the defaults (:540-556)
    bio_ik::BioIKKinematicsQueryOptions opts;
    opts.replace = true;
    opts.return_approximate_solution = true;
    opts.goals.emplace_back(new bio_ik::PositionGoal("r_wrist_roll_link", bio_ik::Vector3(0.5, -0.2, 0.9)));
    opts.goals.emplace_back(new bio_ik::MinimalDisplacementGoal(0.1));
    CHECK(quick.searchPositionIK(std::vector<geometry_msgs::Pose>(), seeds[0], TEST_TIMEOUT, std::vector<double>(), solution, IKCallbackFn(), code, opts));
    CHECK(opts.solution_fitness >= 0.0);

    // Goals without a device implementation (LinkFunctionGoal, JointFunctionGoal, a user's subclass: the reference calls them inside its solver loop,
    // problem.cpp:244-257): the hybrid path of plugin_core.h -- the device searches over the goals it can evaluate, gpu_host_goal_candidates times per
    // query with independent random streams, and the host scores the candidates with the callback goals added.
    {
        BioIKKinematicsPlugin hybrid;
        BioIKParams hp = params;
        hp.gpu_max_steps = 60, hp.gpu_host_goal_candidates = 4, hp.gpu_reproducible_calls = true;
#ifdef TEST_ISLANDS
        hp.gpu_islands = TEST_ISLANDS;
#endif
        CHECK(hybrid.initialize(rm, "right_arm", "torso_lift_link", {"r_wrist_roll_link"}, 0.0, hp));
        // the candidates themselves: with reproducible calls, candidate j of query 0 draws from stream j -- as query j of a plain batch of four copies does
        std::vector<std::vector<geometry_msgs::Pose>> p4(4, poses[0]);
        std::vector<std::vector<double>> s4(4, seeds[0]), cand;
        std::vector<moveit_msgs::MoveItErrorCodes> c4;
        CHECK(hybrid.searchPositionIKBatch(p4, s4, cand, c4));
        auto elbow_z = [&](const std::vector<double>& sol) {
            std::vector<double> st = rm.defaultPositions();
            for (size_t i = 0; i < gv.size(); i++) st[gv[i]] = sol[i];
            double f[7];
            rm.linkTransform(rm.linkIndex("r_elbow_flex_link"), st, f);
            return f[2];
        };
        size_t lowest = 0, highest = 0;
        for (size_t j = 1; j < 4; j++) {
            if (elbow_z(cand[j]) < elbow_z(cand[lowest])) lowest = j;
            if (elbow_z(cand[j]) > elbow_z(cand[highest])) highest = j;
        }
        CHECK(elbow_z(cand[highest]) - elbow_z(cand[lowest]) > 1e-6);  // (a 7-joint arm reaches a pose along a one-parameter family: the streams find different members)
        for (int sign : {+1, -1}) {  // a SECONDARY callback goal that wants the elbow low (+1) or high (-1): it decides which candidate is returned
            bio_ik::BioIKKinematicsQueryOptions o;
            o.goals.emplace_back(new SecondaryLinkFunctionGoal("r_elbow_flex_link", [sign](const bio_ik::Vector3& p, const bio_ik::Quaternion&) { return (sign * p.z() + 3.0) * (sign * p.z() + 3.0); }));
            CHECK(hybrid.searchPositionIK(poses[0], seeds[0], TEST_TIMEOUT, std::vector<double>(), solution, IKCallbackFn(), code, o) && code.val == moveit_msgs::MoveItErrorCodes::SUCCESS);
            CHECK(solution == cand[sign > 0 ? lowest : highest]);
            CHECK(o.solution_fitness > 0.0);
        }
        {  // a PRIMARY callback goal gates success by the rule for goal classes the success test does not know: weight^2 cost < min(dpos, dtwist)^2 (problem.cpp:327-334)
            bio_ik::BioIKKinematicsQueryOptions o;
            o.goals.emplace_back(new bio_ik::JointFunctionGoal({"r_elbow_flex_joint"}, [](std::vector<double>& v) { v[0] = v[0] + 1.0; }));  // cost 1: never met
            CHECK(!hybrid.searchPositionIK(poses[0], seeds[0], TEST_TIMEOUT, std::vector<double>(), solution, IKCallbackFn(), code, o) && code.val == moveit_msgs::MoveItErrorCodes::NO_IK_SOLUTION);
            o.return_approximate_solution = true;
            CHECK(hybrid.searchPositionIK(poses[0], seeds[0], TEST_TIMEOUT, std::vector<double>(), solution, IKCallbackFn(), code, o) && solution.size() == 7);
            bio_ik::BioIKKinematicsQueryOptions met;
            met.goals.emplace_back(new bio_ik::JointFunctionGoal({"r_elbow_flex_joint"}, [](std::vector<double>&) {}));  // cost 0: always met
            CHECK(hybrid.searchPositionIK(poses[0], seeds[0], TEST_TIMEOUT, std::vector<double>(), solution, IKCallbackFn(), code, met) && code.val == moveit_msgs::MoveItErrorCodes::SUCCESS);
        }
        {  // nothing but callbacks: there is no goal the device search could follow -- a configuration error with a message
            bio_ik::BioIKKinematicsQueryOptions o;
            o.replace = true;
            o.goals.emplace_back(new bio_ik::LinkFunctionGoal("r_wrist_roll_link", [](const bio_ik::Vector3& p, const bio_ik::Quaternion&) { return p.z(); }));
            bool threw = false;
            try {
                hybrid.searchPositionIK(std::vector<geometry_msgs::Pose>(), seeds[0], TEST_TIMEOUT, std::vector<double>(), solution, IKCallbackFn(), code, o);
            } catch (const std::runtime_error& e) {
                threw = std::string(e.what()).find("at least one goal with a device implementation") != std::string::npos;
            }
            CHECK(threw);
        }
        {  // a batch that is submitted and never waited for: the ticket's destructor lets the solves finish before their arrays go (no use after free)
            std::vector<std::vector<geometry_msgs::Pose>> pp(2, poses[0]);
            std::vector<std::vector<double>> ss(2, seeds[0]);
            { auto pending = hybrid.searchPositionIKBatchAsync(pp, ss); }
            CHECK(hybrid.searchPositionIKBatch(pp, ss, cand, c4));
        }
    }
    std::printf("ok\n");
    return 0;
}

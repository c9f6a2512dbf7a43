// The closed forms of bio_ik/goal_types.h (`Goal::evaluate`, run on the host through bio_ik/goal_eval.h) against the goal costs the
// solver kernels compute (`bioik_eval_fitness`, exact FK): every built-in goal type with a device opcode, primary and secondary, on
// random configurations of the PR2-like right arm.  Then the goals that exist on the host only (JointFunctionGoal, LinkFunctionGoal, a
// user-defined subclass): describe / evaluate through the same GoalContext interface as the reference's.
#include <cstdio>
#include <random>

#include <bio_ik/bio_ik.h>
#include <bio_ik/goal_eval.h>

#include "pr2_arm_fixture.h"

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                   \
        }                                                               \
    } while (0)

struct ElbowHeightGoal : bio_ik::Goal {  // a user's own goal class, written against the reference's interface
    void describe(bio_ik::GoalContext& context) const override {
        Goal::describe(context);
        context.addLink("r_elbow_flex_link");
        context.addVariable("r_wrist_roll_joint");
    }
    double evaluate(const bio_ik::GoalContext& context) const override {
        const double dz = context.getLinkFrame().getPosition().z() - 1.0, w = context.getVariablePosition();
        return dz * dz + 0.01 * w * w;
    }
};

int main() {
    using namespace bio_ik;
    RobotModel rm = pr2Arm();
    const std::string tip = "r_wrist_roll_link", elbow = "r_elbow_flex_link";
    std::vector<std::unique_ptr<Goal>> goals;
    goals.emplace_back(new PoseGoal(tip, Vector3(0.5, -0.3, 0.9), Quaternion(0.1, 0.2, 0.3, 0.9), 1.0));
    goals.emplace_back(new PositionGoal(elbow, Vector3(0.3, -0.3, 0.8), 0.8));
    goals.emplace_back(new OrientationGoal(elbow, Quaternion(0.0, 0.3, 0.1, 0.9), 0.6));
    goals.emplace_back(new LookAtGoal(tip, Vector3(1, 0, 0), Vector3(1.0, 0.5, 0.7), 0.5));
    goals.emplace_back(new MaxDistanceGoal(tip, Vector3(0.2, 0.2, 0.2), 0.3, 1.1));
    goals.emplace_back(new MinDistanceGoal(tip, Vector3(0.25, 0.2, 0.5), 0.9, 0.9));
    goals.emplace_back(new LineGoal(elbow, Vector3(0.0, 0.0, 0.5), Vector3(0.0, 1.0, 0.5), 0.4));
    goals.emplace_back(new PlaneGoal(elbow, Vector3(0.0, 0.0, 0.6), Vector3(0.0, 0.2, 1.0), 0.3));
    goals.emplace_back(new SideGoal(tip, Vector3(0, 0, 1), Vector3(0, 1, 0), 0.7));
    goals.emplace_back(new DirectionGoal(tip, Vector3(0, 0, 1), Vector3(1, 0, 0), 0.2));
    goals.emplace_back(new ConeGoal(tip, Vector3(0.1, 0.0, 0.9), 0.5, Vector3(1, 0, 0), Vector3(0, 0, 1), 0.3, 0.6));
    goals.emplace_back(new AvoidJointLimitsGoal(0.3, false));
    goals.emplace_back(new CenterJointsGoal(0.2, false));
    goals.emplace_back(new RegularizationGoal(0.1));
    goals.emplace_back(new JointVariableGoal("r_elbow_flex_joint", -1.0, 0.4));
    goals.emplace_back(new MinimalDisplacementGoal(0.7));                         // secondary by default
    goals.emplace_back(new JointVariableGoal("r_wrist_flex_joint", -0.5, 0.5, true));  // secondary

    // the device side: the same goals serialised through their gpu*() members
    bioik_model* model = nullptr;
    bioik_model_desc md = rm.desc();
    CHECK(bioik_model_create(&md, 0, &model) == BIOIK_OK);
    std::vector<bioik_goal_desc> gd;
    std::vector<double> params;
    for (auto& g : goals) {
        bioik_goal_desc d{g->gpuOpcode(), -1, -1, g->isSecondary() ? 1 : 0, g->getWeight()};
        CHECK(d.type >= 0);
        if (!g->gpuLinkName().empty()) d.link = rm.linkIndex(g->gpuLinkName());
        if (!g->gpuVariableName().empty()) d.variable = rm.variableIndex(g->gpuVariableName());
        gd.push_back(d);
        g->gpuParams(params);
    }
    const JointModelGroup& jmg = rm.groups.at("right_arm");
    bioik_problem_desc pd{};
    pd.struct_size = sizeof(pd);
    pd.n_group_joints = (uint32_t)jmg.active_joints.size(), pd.group_joints = jmg.active_joints.data();
    pd.n_goals = (uint32_t)gd.size(), pd.goals = gd.data();
    bioik_problem* problem = nullptr;
    CHECK(bioik_problem_create(model, &pd, &problem) == BIOIK_OK);
    CHECK((size_t)bioik_problem_param_count(problem) == params.size());
    const int D = bioik_problem_active_variable_count(problem);
    std::vector<int32_t> active(D);
    bioik_problem_active_variables(problem, active.data());

    // the host side: the goals' own describe() / evaluate()
    HostGoalProblem::Model hm;
    for (size_t v = 0; v < rm.variable_names.size(); v++) {
        bool revolute = false, prismatic = false;
        for (size_t l = 0; l < rm.joint_type.size(); l++)
            if (rm.joint_first_variable[l] == (int)v) revolute = rm.joint_type[l] == BIOIK_JOINT_REVOLUTE, prismatic = rm.joint_type[l] == BIOIK_JOINT_PRISMATIC;
        hm.info.addVariable(rm.var_min[v], rm.var_max[v], rm.var_bounded[v], rm.var_max_velocity[v], revolute, prismatic);
    }
    hm.variable_index = [&](const std::string& n) { return rm.variableIndex(n); };
    hm.link_frame = [&](const std::string& n, const std::vector<double>& positions) {
        double f[7];
        rm.linkTransform(rm.linkIndex(n), positions, f);
        return Frame(Vector3(f[0], f[1], f[2]), Quaternion(f[3], f[4], f[5], f[6]));
    };
    std::vector<const Goal*> goal_ptrs;
    for (auto& g : goals) goal_ptrs.push_back(g.get());
    std::mt19937 rng(11);
    auto uniform = [&](double lo, double hi) { return std::uniform_real_distribution<double>(lo, hi)(rng); };
    std::vector<double> seed = rm.defaultPositions();
    for (int v : active) seed[v] = uniform(rm.var_min[v], rm.var_max[v]);
    HostGoalProblem host(hm, goal_ptrs, std::vector<int>(active.begin(), active.end()), seed);
    CHECK(host.getTipNames().size() == 2);

    const int n = 24;
    std::vector<double> genes(n * D), prim(n), sec(n);
    std::vector<std::vector<double>> full(n, seed);
    for (int k = 0; k < n; k++)
        for (int i = 0; i < D; i++) full[k][active[i]] = genes[k * D + i] = uniform(rm.var_min[active[i]], rm.var_max[active[i]]);
    CHECK(bioik_eval_fitness(problem, BIOIK_FK_EXACT, n, seed.data(), params.data(), nullptr, genes.data(), prim.data(), sec.data()) == BIOIK_OK);
    double worst = 0.0;
    for (int k = 0; k < n; k++) {
        const double hp = host.computeGoalFitness(full[k], false), hs = host.computeGoalFitness(full[k], true);
        worst = std::max(worst, std::max(std::fabs(hp - prim[k]) / std::max(1.0, std::fabs(hp)), std::fabs(hs - sec[k]) / std::max(1.0, std::fabs(hs))));
    }
    std::printf("host evaluate() vs device goal costs, %d configurations x %zu goals: worst relative difference %.3g\n", n, goals.size(), worst);
    CHECK(worst < 1e-10);

    // goals that exist on the host only
    {
        std::vector<std::unique_ptr<Goal>> hg;
        hg.emplace_back(new LinkFunctionGoal(tip, [](const Vector3& p, const Quaternion& q) { return p.z() * p.z() + q.w(); }, 2.0));
        hg.emplace_back(new JointFunctionGoal({"r_elbow_flex_joint", "r_wrist_flex_joint"}, [](std::vector<double>& v) { v[0] = -1.0, v[1] = v[1] * 0.5; }, 1.0, true));
        hg.emplace_back(new ElbowHeightGoal());
        for (auto& g : hg) CHECK(g->gpuOpcode() == -1);
        std::vector<const Goal*> hp;
        for (auto& g : hg) hp.push_back(g.get());
        HostGoalProblem h2(hm, hp, std::vector<int>(active.begin(), active.end()), seed);
        const std::vector<double>& x = full[0];
        const std::vector<double> e = h2.evaluateGoals(x);
        const Frame ft = hm.link_frame(tip, x), fe = hm.link_frame(elbow, x);
        const double je = x[rm.variableIndex("r_elbow_flex_joint")], jw = x[rm.variableIndex("r_wrist_flex_joint")], jr = x[rm.variableIndex("r_wrist_roll_joint")];
        CHECK(std::fabs(e[0] - (ft.pos.z() * ft.pos.z() + ft.rot.w())) < 1e-15);
        CHECK(std::fabs(e[1] - ((je + 1.0) * (je + 1.0) + 0.25 * jw * jw)) < 1e-15);
        CHECK(std::fabs(e[2] - ((fe.pos.z() - 1.0) * (fe.pos.z() - 1.0) + 0.01 * jr * jr)) < 1e-15);
        CHECK(std::fabs(h2.computeGoalFitness(x, false) - (4.0 * e[0] + e[2])) < 1e-12 && std::fabs(h2.computeGoalFitness(x, true) - e[1]) < 1e-15);
    }
    // conversions the goal setters accept: anything with x() y() z() (w())
    {
        struct V {
            double x() const { return 1; }
            double y() const { return 2; }
            double z() const { return 3; }
        };
        PositionGoal g;
        g.setPosition(V());
        CHECK(g.getPosition().y() == 2);
    }
    bioik_problem_destroy(problem);
    bioik_model_destroy(model);
    std::printf("ok\n");
    return 0;
}

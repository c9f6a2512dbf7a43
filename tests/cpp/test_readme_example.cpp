// Harness for the reference's own usage example (README.md "PR2 turning a valve", the three code blocks after line 199).  The example
// itself is NOT in this repository: tests/test_readme_example.py cuts it out of /root/reference/README.md when that tree is present and
// passes it in as EXAMPLE_FILE, unchanged except for the `...` placeholder line, which becomes the hook EXAMPLE_AFTER_IK.  What this file
// provides is what the example assumes around it: a MoveIt-loaded PR2 with the link names it uses (stand-in RobotModel, PR2-like numbers),
// a `robot_state`, a `joint_model_group` whose solver instance is the bio_ik plugin created through pluginlib, and the includes a ROS
// program would have.  Passing means: the goal (cost) plugin interface is source compatible for this user code -- tf vectors into the
// goal setters, BioIKKinematicsQueryOptions through RobotState::setFromIK -- and the solves go through the device path.
#include <cmath>
#include <cstdio>
#include <memory>

#include <eigen_stl_containers/eigen_stl_containers.h>
#include <moveit/kinematics_base/kinematics_base.h>
#include <moveit/robot_model/robot_model.h>
#include <moveit/robot_state/robot_state.h>
#include <ros/ros.h>
#include <tf/tf.h>

#define BIOIK_WITH_KINEMATICS_BASE 1
#include <bio_ik/bio_ik.h>

extern "C" void* pluginlib_standin_create(const char* derived_type_name);  // libbio_ik.so, stand-in pluginlib build

static moveit::core::RobotModelPtr pr2WithGrippersAndHead() {
    moveit::core::RobotModelPtr m(new moveit::core::RobotModel());
    m->addLink("base_footprint", "", "world_joint", "fixed", 0, 0, 0, 0, 0, 0, 0, 0, 1);
    m->addLink("base_link", "base_footprint", "base_footprint_joint", "fixed", 0, 0, 0.051, 0, 0, 0, 0, 0, 1);
    m->addLink("torso_lift_link", "base_link", "torso_lift_joint", "prismatic", -0.05, 0, 0.739675, 0, 0, 0, 0, 0, 1, 0.0, 0.33, 0.013);
    m->addLink("head_pan_link", "torso_lift_link", "head_pan_joint", "revolute", -0.01707, 0, 0.38145, 0, 0, 0, 0, 0, 1, -3.007, 3.007, 6.0);
    m->addLink("head_tilt_link", "head_pan_link", "head_tilt_joint", "revolute", 0.068, 0, 0, 0, 0, 0, 0, 1, 0, -0.471, 1.396, 5.0);
    m->addLink("sensor_mount_link", "head_tilt_link", "sensor_mount_frame_joint", "fixed", 0.0232, 0, 0.0645, 0, 0, 0, 0, 0, 1);
    for (const char* side : {"r", "l"}) {
        const std::string s(side);
        const double y = s == "r" ? -0.188 : 0.188;
        m->addLink(s + "_shoulder_pan_link", "torso_lift_link", s + "_shoulder_pan_joint", "revolute", 0, y, 0, 0, 0, 0, 0, 0, 1, s == "r" ? -2.2854 : -0.7146,
                   s == "r" ? 0.7146 : 2.2854, 2.088);
        m->addLink(s + "_shoulder_lift_link", s + "_shoulder_pan_link", s + "_shoulder_lift_joint", "revolute", 0.1, 0, 0, 0, 0, 0, 0, 1, 0, -0.5236, 1.3963, 2.082);
        m->addLink(s + "_upper_arm_roll_link", s + "_shoulder_lift_link", s + "_upper_arm_roll_joint", "revolute", 0, 0, 0, 0, 0, 0, 1, 0, 0, s == "r" ? -3.9 : -0.8,
                   s == "r" ? 0.8 : 3.9, 3.27);
        m->addLink(s + "_elbow_flex_link", s + "_upper_arm_roll_link", s + "_elbow_flex_joint", "revolute", 0.4, 0, 0, 0, 0, 0, 0, 1, 0, -2.3213, 0.0, 3.3);
        m->addLink(s + "_forearm_roll_link", s + "_elbow_flex_link", s + "_forearm_roll_joint", "continuous", 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 3.6);
        m->addLink(s + "_wrist_flex_link", s + "_forearm_roll_link", s + "_wrist_flex_joint", "revolute", 0.321, 0, 0, 0, 0, 0, 0, 1, 0, -2.18, 0.0, 3.078);
        m->addLink(s + "_wrist_roll_link", s + "_wrist_flex_link", s + "_wrist_roll_joint", "continuous", 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 3.6);
        m->addLink(s + "_gripper_palm_link", s + "_wrist_roll_link", s + "_gripper_palm_joint", "fixed", 0, 0, 0, 0, 0, 0, 0, 0, 1);
        // the two fingers open symmetrically (one driven joint, the other mimics it), finger tips 0.17 m in front of the wrist
        m->addLink(s + "_gripper_l_finger_tip_link", s + "_gripper_palm_link", s + "_gripper_l_finger_joint", "prismatic", 0.17, 0.01, 0, 0, 0, 0, 0, 1, 0, 0.0, 0.044, 0.2);
        m->addLink(s + "_gripper_r_finger_tip_link", s + "_gripper_palm_link", s + "_gripper_r_finger_joint", "prismatic", 0.17, -0.01, 0, 0, 0, 0, 0, -1, 0, 0.0, 0.044, 0.2);
        m->setMimic(s + "_gripper_r_finger_joint", s + "_gripper_l_finger_joint", 1.0, 0.0);
    }
    std::vector<std::string> joints = {"torso_lift_joint", "head_pan_joint", "head_tilt_joint"};
    for (const char* side : {"r", "l"})
        for (const char* j : {"_shoulder_pan_joint", "_shoulder_lift_joint", "_upper_arm_roll_joint", "_elbow_flex_joint", "_forearm_roll_joint", "_wrist_flex_joint",
                              "_wrist_roll_joint", "_gripper_l_finger_joint", "_gripper_r_finger_joint"})
            joints.push_back(std::string(side) + j);
    m->addJointsGroup("all", joints, {});
    return m;
}

static int iterations_done = 0;
static bool last_ok = false;
#ifndef EXAMPLE_ITERATIONS
#define EXAMPLE_ITERATIONS 3
#endif
// what stands where the example says "... // check solution validity and actually move the robot"
#define EXAMPLE_AFTER_IK(ok)             \
    last_ok = (ok);                      \
    if (++iterations_done >= EXAMPLE_ITERATIONS || !last_ok) break;

int main() {
    ros::set_param("mode", "bio2_memetic");
    ros::set_param("random_seed", 1);
    ros::set_param("gpu_population", 32);
    ros::set_param("gpu_max_steps", EXAMPLE_MAX_STEPS);
#ifdef TEST_ISLANDS
    ros::set_param("gpu_islands", TEST_ISLANDS);
#endif
    moveit::core::RobotModelPtr robot_model = pr2WithGrippersAndHead();
    std::shared_ptr<kinematics::KinematicsBase> solver(static_cast<kinematics::KinematicsBase*>(pluginlib_standin_create("bio_ik_kinematics_plugin::BioIKKinematicsPlugin")));
    if (!solver || !solver->initialize(*robot_model, "all", "base_footprint", std::vector<std::string>{"r_wrist_roll_link", "l_wrist_roll_link"}, 0.0)) return 2;
    robot_model->getJointModelGroup("all")->setSolverInstance(solver);
    const moveit::core::JointModelGroup* joint_model_group = static_cast<const moveit::core::RobotModel&>(*robot_model).getJointModelGroup("all");
    moveit::core::RobotState robot_state(robot_model);
    robot_state.setToDefaultValues();
    for (const char* side : {"r", "l"}) {  // arms forward, elbows bent: a posture from which the valve is in reach
        robot_state.setVariablePosition(std::string(side) + "_shoulder_lift_joint", 0.3);
        robot_state.setVariablePosition(std::string(side) + "_elbow_flex_joint", -1.2);
        robot_state.setVariablePosition(std::string(side) + "_wrist_flex_joint", -0.4);
    }
    {
#include EXAMPLE_FILE
    }
    if (iterations_done < EXAMPLE_ITERATIONS || !last_ok) {
        std::printf("FAILED: %d iterations, last setFromIK %d\n", iterations_done, (int)last_ok);
        return 1;
    }
    // the hands went where the goals of the last iteration put them (approximate solutions are allowed by the example: loose bound)
    const Eigen::Isometry3d L = robot_state.getGlobalLinkTransform("l_gripper_l_finger_tip_link"), R = robot_state.getGlobalLinkTransform("r_gripper_r_finger_tip_link");
    std::printf("after %d iterations: left finger tip (%.3f %.3f %.3f), right finger tip (%.3f %.3f %.3f)\n", iterations_done, L.translation().x(), L.translation().y(),
                L.translation().z(), R.translation().x(), R.translation().y(), R.translation().z());
    if (std::fabs(L.translation().x() - 0.7) > 0.15 || std::fabs(R.translation().x() - 0.7) > 0.15 || std::fabs(L.translation().z() - 1.0) > 0.2) return 1;
    std::printf("ok\n");
    return 0;
}

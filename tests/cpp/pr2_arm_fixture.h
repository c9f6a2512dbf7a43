// the PR2-like right arm of the C++ tests (bio_ik::RobotModel; SURVEY.md Appendix C numbers)
#pragma once
#include <bio_ik/robot_model.h>

static bio_ik::RobotModel pr2Arm() {
    bio_ik::RobotModel m;
    const double z[3] = {0, 0, 0}, ax[3] = {1, 0, 0}, ay[3] = {0, 1, 0}, az[3] = {0, 0, 1};
    m.addLink("base_footprint", "", "", "fixed", z, z, az);
    m.addLink("base_link", "base_footprint", "base_footprint_joint", "fixed", {0, 0, 0.051}, z, az);
    m.addLink("torso_lift_link", "base_link", "torso_lift_joint", "prismatic", {-0.05, 0, 0.739675}, z, az, 0.0, 0.33, 0.013);
    m.addLink("r_shoulder_pan_link", "torso_lift_link", "r_shoulder_pan_joint", "revolute", {0, -0.188, 0}, z, az, -2.2854, 0.7146, 2.088);
    m.addLink("r_shoulder_lift_link", "r_shoulder_pan_link", "r_shoulder_lift_joint", "revolute", {0.1, 0, 0}, z, ay, -0.5236, 1.3963, 2.082);
    m.addLink("r_upper_arm_roll_link", "r_shoulder_lift_link", "r_upper_arm_roll_joint", "revolute", z, z, ax, -3.9, 0.8, 3.27);
    m.addLink("r_upper_arm_link", "r_upper_arm_roll_link", "r_upper_arm_joint", "fixed", z, z, az);
    m.addLink("r_elbow_flex_link", "r_upper_arm_link", "r_elbow_flex_joint", "revolute", {0.4, 0, 0}, z, ay, -2.3213, 0.0, 3.3);
    m.addLink("r_forearm_roll_link", "r_elbow_flex_link", "r_forearm_roll_joint", "continuous", z, z, ax, 0, 0, 3.6);
    m.addLink("r_forearm_link", "r_forearm_roll_link", "r_forearm_joint", "fixed", z, z, az);
    m.addLink("r_wrist_flex_link", "r_forearm_link", "r_wrist_flex_joint", "revolute", {0.321, 0, 0}, z, ay, -2.18, 0.0, 3.078);
    m.addLink("r_wrist_roll_link", "r_wrist_flex_link", "r_wrist_roll_joint", "continuous", z, z, ax, 0, 0, 3.6);
    m.addChainGroup("right_arm", "torso_lift_link", "r_wrist_roll_link");
    return m;
}


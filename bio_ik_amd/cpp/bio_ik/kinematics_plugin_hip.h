// bio_ik/kinematics_plugin_hip.h — what libbio_ik (MI355X build) exports next to the pluginlib class `bio_ik/BioIKKinematicsPlugin`:
// the additive batched form of searchPositionIK (the reference's interface, src/kinematics_plugin.cpp:437-446, solves one query per call).
#pragma once
#include <memory>
#include <vector>

#include <geometry_msgs/Pose.h>
#include <moveit/kinematics_base/kinematics_base.h>
#include <moveit/robot_state/robot_state.h>
#include <moveit_msgs/MoveItErrorCodes.h>

namespace bio_ik_kinematics_plugin {

// n independent queries sharing one goal structure in ONE device launch.  `solver` must be the plugin instance MoveIt loaded
// (kinematics::KinematicsBase of class bio_ik/BioIKKinematicsPlugin); ik_poses[k] holds the tip poses of query k in the base frame
// (ignored with BioIKKinematicsQueryOptions::replace), ik_seed_states[k] its group variables; `timeout` [s] bounds the whole call.
// Returns true iff every query has an acceptable solution; the per-query verdicts are in error_codes.
bool searchPositionIKBatch(const kinematics::KinematicsBase& solver, const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses,
                           const std::vector<std::vector<double>>& ik_seed_states, double timeout, std::vector<std::vector<double>>& solutions,
                           std::vector<moveit_msgs::MoveItErrorCodes>& error_codes,
                           const kinematics::KinematicsQueryOptions& options = kinematics::KinematicsQueryOptions(),
                           const moveit::core::RobotState* context_state = nullptr);

// The same without waiting: the batch is marshalled and enqueued (transfers and kernels on one of the solver handle's SIX streams per
// device), and the call returns a ticket; searchPositionIKBatchWait blocks until that batch is complete and post-processed.  A caller
// with a stream of batches keeps up to six in flight and gets the throughput the device reaches on a stream of batches -- the slow
// tail of one batch runs behind the bulk of the next.  `ik_seed_states` and `options` must stay alive until the wait
// (the wait writes BioIKKinematicsQueryOptions::solution_fitness).  The marshalling time is taken off `timeout`; the device counts what is
// left from the moment the batch's first workgroup runs, so a batch that queues behind others of the pipeline is NOT cut short by its wait.
// A ticket that is destroyed without having been waited for lets its solves finish first (its arrays are what they write into).
struct BatchTicket {
    struct Impl;
    std::unique_ptr<Impl> impl;
    BatchTicket();
    ~BatchTicket();
    BatchTicket(BatchTicket&&);
    BatchTicket& operator=(BatchTicket&&);
};
BatchTicket searchPositionIKBatchAsync(const kinematics::KinematicsBase& solver, const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses,
                                       const std::vector<std::vector<double>>& ik_seed_states, double timeout,
                                       const kinematics::KinematicsQueryOptions& options = kinematics::KinematicsQueryOptions(),
                                       const moveit::core::RobotState* context_state = nullptr);
bool searchPositionIKBatchWait(const kinematics::KinematicsBase& solver, BatchTicket& ticket, std::vector<std::vector<double>>& solutions,
                               std::vector<moveit_msgs::MoveItErrorCodes>& error_codes);

}  // namespace bio_ik_kinematics_plugin

// TEST STUB (tests/native/ros2_stubs/README.md) -- kinematics::KinematicsBase as far as pick_ik's
// plugin overrides / uses it (moveit_core kinematics_base.h): the virtual interface, storeValues,
// the callback types and the query options.
#pragma once
#include <functional>
#include <geometry_msgs/msg/pose.hpp>
#include <memory>
#include <moveit/robot_model/robot_model.h>
#include <moveit/robot_state/robot_state.h>
#include <moveit_msgs/msg/move_it_error_codes.hpp>
#include <rclcpp/rclcpp.hpp>
#include <string>
#include <vector>
namespace kinematics {
struct KinematicsQueryOptions {
    bool lock_redundant_joints = false;
    bool return_approximate_solution = false;
    int discretization_method = 0;
};
class KinematicsBase {
  public:
    using IKCallbackFn = std::function<void(geometry_msgs::msg::Pose const&, std::vector<double> const&,
                                            moveit_msgs::msg::MoveItErrorCodes&)>;
    using IKCostFn = std::function<double(geometry_msgs::msg::Pose const&, moveit::core::RobotState const&,
                                          moveit::core::JointModelGroup const*, std::vector<double> const&)>;
    virtual ~KinematicsBase() = default;
    virtual bool getPositionIK(geometry_msgs::msg::Pose const& ik_pose, std::vector<double> const& ik_seed_state,
                               std::vector<double>& solution, moveit_msgs::msg::MoveItErrorCodes& error_code,
                               KinematicsQueryOptions const& options = KinematicsQueryOptions()) const = 0;
    virtual bool searchPositionIK(geometry_msgs::msg::Pose const& ik_pose, std::vector<double> const& ik_seed_state,
                                  double timeout, std::vector<double>& solution,
                                  moveit_msgs::msg::MoveItErrorCodes& error_code,
                                  KinematicsQueryOptions const& options = KinematicsQueryOptions()) const = 0;
    virtual bool searchPositionIK(geometry_msgs::msg::Pose const& ik_pose, std::vector<double> const& ik_seed_state,
                                  double timeout, std::vector<double> const& consistency_limits,
                                  std::vector<double>& solution, moveit_msgs::msg::MoveItErrorCodes& error_code,
                                  KinematicsQueryOptions const& options = KinematicsQueryOptions()) const = 0;
    virtual bool searchPositionIK(geometry_msgs::msg::Pose const& ik_pose, std::vector<double> const& ik_seed_state,
                                  double timeout, std::vector<double>& solution, IKCallbackFn const& solution_callback,
                                  moveit_msgs::msg::MoveItErrorCodes& error_code,
                                  KinematicsQueryOptions const& options = KinematicsQueryOptions()) const = 0;
    virtual bool searchPositionIK(geometry_msgs::msg::Pose const& ik_pose, std::vector<double> const& ik_seed_state,
                                  double timeout, std::vector<double> const& consistency_limits,
                                  std::vector<double>& solution, IKCallbackFn const& solution_callback,
                                  moveit_msgs::msg::MoveItErrorCodes& error_code,
                                  KinematicsQueryOptions const& options = KinematicsQueryOptions()) const = 0;
    // the pose-vector forms have default implementations in MoveIt (they forward / refuse)
    virtual bool searchPositionIK(std::vector<geometry_msgs::msg::Pose> const& ik_poses,
                                  std::vector<double> const& ik_seed_state, double timeout,
                                  std::vector<double> const& consistency_limits, std::vector<double>& solution,
                                  IKCallbackFn const& solution_callback, moveit_msgs::msg::MoveItErrorCodes& error_code,
                                  KinematicsQueryOptions const& options = KinematicsQueryOptions(),
                                  moveit::core::RobotState const* context_state = nullptr) const {
        (void)context_state;
        if (ik_poses.size() != 1) return false;
        return searchPositionIK(ik_poses[0], ik_seed_state, timeout, consistency_limits, solution, solution_callback,
                                error_code, options);
    }
    virtual bool searchPositionIK(std::vector<geometry_msgs::msg::Pose> const& ik_poses,
                                  std::vector<double> const& ik_seed_state, double timeout,
                                  std::vector<double> const& consistency_limits, std::vector<double>& solution,
                                  IKCallbackFn const& solution_callback, IKCostFn const& cost_function,
                                  moveit_msgs::msg::MoveItErrorCodes& error_code,
                                  KinematicsQueryOptions const& options = KinematicsQueryOptions(),
                                  moveit::core::RobotState const* context_state = nullptr) const {
        (void)cost_function;
        return searchPositionIK(ik_poses, ik_seed_state, timeout, consistency_limits, solution, solution_callback,
                                error_code, options, context_state);
    }
    virtual bool getPositionFK(std::vector<std::string> const& link_names, std::vector<double> const& joint_angles,
                               std::vector<geometry_msgs::msg::Pose>& poses) const = 0;
    virtual bool initialize(rclcpp::Node::SharedPtr const& node, moveit::core::RobotModel const& robot_model,
                            std::string const& group_name, std::string const& base_frame,
                            std::vector<std::string> const& tip_frames, double search_discretization) = 0;
    virtual std::vector<std::string> const& getJointNames() const = 0;
    virtual std::vector<std::string> const& getLinkNames() const = 0;
    virtual std::string const& getGroupName() const { return group_name_; }
    virtual std::string const& getBaseFrame() const { return base_frame_; }
    virtual std::vector<std::string> const& getTipFrames() const { return tip_frames_; }

  protected:
    // MoveIt copies nothing: it keeps a shared pointer to the model the caller owns; the stub takes
    // the address of the reference it is given (the test keeps the model alive)
    void storeValues(moveit::core::RobotModel const& robot_model, std::string const& group_name,
                     std::string const& base_frame, std::vector<std::string> const& tip_frames,
                     double search_discretization) {
        robot_model_ = moveit::core::RobotModelConstPtr(&robot_model, [](moveit::core::RobotModel const*) {});
        group_name_ = group_name;
        base_frame_ = base_frame;
        tip_frames_ = tip_frames;
        search_discretization_ = search_discretization;
    }
    moveit::core::RobotModelConstPtr robot_model_;
    std::string group_name_, base_frame_;
    std::vector<std::string> tip_frames_;
    double search_discretization_ = 0.0;
};
} // namespace kinematics

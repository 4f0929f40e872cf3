// TEST STUB (tests/native/ros2_stubs/README.md) -- tf2::fromMsg(Pose, Isometry3d&): Translation * Quaterniond (not normalised)
#pragma once
#include <Eigen/Geometry>
#include <geometry_msgs/msg/pose.hpp>
namespace tf2 {
inline void fromMsg(geometry_msgs::msg::Pose const& msg, Eigen::Isometry3d& out) {
    out.t = Eigen::Vector3d(msg.position.x, msg.position.y, msg.position.z);
    out.R = Eigen::Quaterniond(msg.orientation.w, msg.orientation.x, msg.orientation.y, msg.orientation.z).toRotationMatrix();
}
} // namespace tf2

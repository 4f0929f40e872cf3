// TEST STUB (tests/native/ros2_stubs/README.md) -- tf2::fromMsg(Pose, Isometry3d&): Translation * Quaterniond (not normalised);
// tf2::toMsg(Isometry3d) -> Pose
#pragma once
#include <Eigen/Geometry>
#include <geometry_msgs/msg/pose.hpp>
namespace tf2 {
inline void fromMsg(geometry_msgs::msg::Pose const& msg, Eigen::Isometry3d& out) {
    out.t = Eigen::Vector3d(msg.position.x, msg.position.y, msg.position.z);
    out.R = Eigen::Quaterniond(msg.orientation.w, msg.orientation.x, msg.orientation.y, msg.orientation.z).toRotationMatrix();
}
inline geometry_msgs::msg::Pose toMsg(Eigen::Isometry3d const& in) {
    Eigen::Quaterniond const q(in.rotation());
    geometry_msgs::msg::Pose p;
    p.position.x = in.translation().x();
    p.position.y = in.translation().y();
    p.position.z = in.translation().z();
    p.orientation.w = q.w();
    p.orientation.x = q.x();
    p.orientation.y = q.y();
    p.orientation.z = q.z();
    return p;
}
} // namespace tf2

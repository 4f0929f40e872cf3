// TEST STUB (tests/native/ros2_stubs/README.md) -- geometry_msgs::msg::Pose
#pragma once
namespace geometry_msgs::msg {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
} // namespace geometry_msgs::msg

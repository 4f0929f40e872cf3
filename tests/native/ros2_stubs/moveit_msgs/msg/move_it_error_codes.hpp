// TEST STUB (tests/native/ros2_stubs/README.md) -- moveit_msgs::msg::MoveItErrorCodes
#pragma once
#include <cstdint>
namespace moveit_msgs::msg {
struct MoveItErrorCodes {
    static constexpr int32_t SUCCESS = 1;
    static constexpr int32_t FAILURE = 99999;
    static constexpr int32_t NO_IK_SOLUTION = -31;
    int32_t val = 0;
};
} // namespace moveit_msgs::msg

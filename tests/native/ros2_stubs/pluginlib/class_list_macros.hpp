// TEST STUB (tests/native/ros2_stubs/README.md) -- PLUGINLIB_EXPORT_CLASS: records the exported type names
#pragma once
#include <string>
#include <utility>
#include <vector>
namespace pluginlib_stub {
inline std::vector<std::pair<std::string, std::string>>& exported() {
    static std::vector<std::pair<std::string, std::string>> v;
    return v;
}
struct Register {
    Register(char const* derived, char const* base) { exported().emplace_back(derived, base); }
};
} // namespace pluginlib_stub
#define PLUGINLIB_EXPORT_CLASS(Derived, Base)                                      \
    static_assert(std::is_base_of<Base, Derived>::value, "plugin must derive");   \
    static pluginlib_stub::Register pluginlib_stub_register_##__LINE__(#Derived, #Base)

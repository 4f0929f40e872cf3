// TEST STUB (tests/native/ros2_stubs/README.md) -- the slice of rclcpp the plugin shim uses.
// Parameters are typed like rclcpp's: integer (int64), double, bool, string; declaring or reading
// one with another type throws (rclcpp::exceptions::InvalidParameterTypeException there).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <variant>
#include <vector>
namespace rclcpp {
struct Logger {
    std::string name;
};
inline Logger get_logger(std::string const& name) { return Logger{name}; }
inline std::vector<std::string>& stub_log() {
    static std::vector<std::string> v;
    return v;
}
inline void stub_logf(Logger const& l, char const* level, char const* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    stub_log().push_back(std::string(level) + " [" + l.name + "] " + buf);
}
class Node {
  public:
    using SharedPtr = std::shared_ptr<Node>;
    using Value = std::variant<bool, int64_t, double, std::string>;
    template <typename T>
    static Value to_value(T const& v) {
        if constexpr (std::is_same_v<T, bool>) return Value(v);
        else if constexpr (std::is_integral_v<T>) return Value(static_cast<int64_t>(v));
        else if constexpr (std::is_floating_point_v<T>) return Value(static_cast<double>(v));
        else return Value(std::string(v));
    }
    bool has_parameter(std::string const& name) const { return params_.count(name) != 0; }
    // a launch file / yaml would have set these before the plugin asks (test: set_parameter)
    template <typename T>
    void set_parameter(std::string const& name, T const& v) { overrides_[name] = to_value(v); }
    template <typename T>
    T declare_parameter(std::string const& name, T const& def) {
        if (has_parameter(name)) throw std::runtime_error("parameter already declared: " + name);
        Value v = to_value(def);
        auto it = overrides_.find(name);
        if (it != overrides_.end()) {
            if (it->second.index() != v.index())
                throw std::runtime_error("parameter '" + name + "' has invalid type (override vs declared default)");
            v = it->second;
        }
        params_[name] = v;
        T out{};
        get_parameter(name, out);
        return out;
    }
    template <typename T>
    bool get_parameter(std::string const& name, T& out) const {
        auto it = params_.find(name);
        if (it == params_.end()) return false;
        Value const want = to_value(T{});
        if (it->second.index() != want.index())
            throw std::runtime_error("parameter '" + name + "' has invalid type");
        if constexpr (std::is_same_v<T, bool>) out = std::get<bool>(it->second);
        else if constexpr (std::is_integral_v<T>) out = static_cast<T>(std::get<int64_t>(it->second));
        else if constexpr (std::is_floating_point_v<T>) out = static_cast<T>(std::get<double>(it->second));
        else out = std::get<std::string>(it->second);
        return true;
    }

  private:
    std::map<std::string, Value> params_, overrides_;
};
} // namespace rclcpp
#define RCLCPP_ERROR(logger, ...) ::rclcpp::stub_logf(logger, "ERROR", __VA_ARGS__)
#define RCLCPP_WARN(logger, ...) ::rclcpp::stub_logf(logger, "WARN", __VA_ARGS__)
#define RCLCPP_INFO(logger, ...) ::rclcpp::stub_logf(logger, "INFO", __VA_ARGS__)

// TEST STUB (tests/native/ros2_stubs/README.md) -- the slice of moveit::core's robot model the
// plugin shim uses, plus a tiny builder (add_link) for the test to describe a kinematic tree.
#pragma once
#include <Eigen/Geometry>
#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <vector>
namespace moveit::core {
struct VariableBounds {
    double min_position_ = 0, max_position_ = 0;
    bool position_bounded_ = false;
    double min_velocity_ = 0, max_velocity_ = 0;
    bool velocity_bounded_ = false;
};
class LinkModel;
class JointModel {
  public:
    enum JointType { UNKNOWN, REVOLUTE, PRISMATIC, PLANAR, FLOATING, FIXED };
    virtual ~JointModel() = default;
    std::string const& getName() const { return name_; }
    JointType getType() const { return type_; }
    JointModel const* getMimic() const { return mimic_; }
    double getMimicFactor() const { return mimic_factor_; }
    double getMimicOffset() const { return mimic_offset_; }
    size_t getVariableCount() const { return type_ == FIXED ? 0 : (type_ == PLANAR ? 3 : (type_ == FLOATING ? 7 : 1)); }
    std::vector<VariableBounds> const& getVariableBounds() const { return bounds_; }
    // moveit_core: zero when the bounds allow it, else the middle of the range (a floating joint: unit quaternion)
    void getVariableDefaultPositions(double* v) const {
        for (size_t i = 0; i < bounds_.size(); ++i) {
            auto const& b = bounds_[i];
            v[i] = (!b.position_bounded_ || (b.min_position_ <= 0.0 && b.max_position_ >= 0.0)) ? 0.0
                                                                                               : 0.5 * (b.min_position_ + b.max_position_);
        }
        if (type_ == FLOATING) v[6] = 1.0;
    }
    void computeTransform(double const* v, Eigen::Isometry3d& J) const {
        J = Eigen::Isometry3d::Identity();
        if (type_ == REVOLUTE) {
            double const c = std::cos(v[0]), s = std::sin(v[0]), t = 1 - c, x = axis_.x(), y = axis_.y(), z = axis_.z();
            J.R(0, 0) = t * x * x + c; J.R(0, 1) = t * x * y - z * s; J.R(0, 2) = t * x * z + y * s;
            J.R(1, 0) = t * x * y + z * s; J.R(1, 1) = t * y * y + c; J.R(1, 2) = t * y * z - x * s;
            J.R(2, 0) = t * x * z - y * s; J.R(2, 1) = t * y * z + x * s; J.R(2, 2) = t * z * z + c;
        } else if (type_ == PRISMATIC) {
            J.t = Eigen::Vector3d(axis_.x() * v[0], axis_.y() * v[0], axis_.z() * v[0]);
        } else if (type_ == PLANAR) {
            J.t = Eigen::Vector3d(v[0], v[1], 0.0);
            J.R(0, 0) = std::cos(v[2]); J.R(0, 1) = -std::sin(v[2]);
            J.R(1, 0) = std::sin(v[2]); J.R(1, 1) = std::cos(v[2]);
        } else if (type_ == FLOATING) {
            J.t = Eigen::Vector3d(v[0], v[1], v[2]);
            J.R = Eigen::Quaterniond(v[6], v[3], v[4], v[5]).toRotationMatrix();
        }
    }
    std::string name_;
    JointType type_ = FIXED;
    JointModel const* mimic_ = nullptr;
    double mimic_factor_ = 1.0, mimic_offset_ = 0.0;
    std::vector<VariableBounds> bounds_;
    Eigen::Vector3d axis_{0, 0, 1};
};
class RevoluteJointModel : public JointModel {
  public:
    Eigen::Vector3d const& getAxis() const { return axis_; }
};
class PrismaticJointModel : public JointModel {
  public:
    Eigen::Vector3d const& getAxis() const { return axis_; }
};
class FixedJointModel : public JointModel {};
class PlanarJointModel : public JointModel {}; // variables x, y, theta: Translation(x, y, 0) * Rz(theta)
// variables trans_x trans_y trans_z rot_x rot_y rot_z rot_w: Translation(t) * Quaterniond(w, x, y, z)
class FloatingJointModel : public JointModel {};
class LinkModel {
  public:
    std::string const& getName() const { return name_; }
    LinkModel const* getParentLinkModel() const { return parent_; }
    JointModel const* getParentJointModel() const { return joint_.get(); }
    Eigen::Isometry3d const& getJointOriginTransform() const { return origin_; }
    std::string name_;
    LinkModel const* parent_ = nullptr;
    std::unique_ptr<JointModel> joint_;
    Eigen::Isometry3d origin_;
};
class JointModelGroup {
  public:
    bool hasJointModel(std::string const& name) const {
        return std::any_of(joints_.begin(), joints_.end(), [&](auto const* j) { return j->getName() == name; });
    }
    std::vector<JointModel const*> const& getActiveJointModels() const { return active_; }
    std::string name_;
    std::vector<JointModel const*> joints_, active_;
};
class RobotModel {
  public:
    std::string const& getModelFrame() const { return links_.front()->getName(); }
    LinkModel const* getLinkModel(std::string const& name) const {
        for (auto const& l : links_)
            if (l->getName() == name) return l.get();
        return nullptr;
    }
    JointModelGroup const* getJointModelGroup(std::string const& name) const {
        for (auto const& g : groups_)
            if (g->name_ == name) return g.get();
        return nullptr;
    }
    // ---- test-side builder ----
    LinkModel* add_root(std::string const& name) {
        links_.push_back(std::make_unique<LinkModel>());
        links_.back()->name_ = name;
        links_.back()->joint_ = std::make_unique<FixedJointModel>();
        links_.back()->joint_->name_ = name + "_root_joint";
        return links_.back().get();
    }
    LinkModel* add_link(std::string const& name, std::string const& parent, std::string const& joint_name,
                        JointModel::JointType type, Eigen::Isometry3d const& origin, Eigen::Vector3d const& axis,
                        VariableBounds const& bounds) {
        auto link = std::make_unique<LinkModel>();
        link->name_ = name;
        link->parent_ = getLinkModel(parent);
        link->origin_ = origin;
        if (type == JointModel::REVOLUTE) link->joint_ = std::make_unique<RevoluteJointModel>();
        else if (type == JointModel::PRISMATIC) link->joint_ = std::make_unique<PrismaticJointModel>();
        else if (type == JointModel::PLANAR) link->joint_ = std::make_unique<PlanarJointModel>();
        else if (type == JointModel::FLOATING) link->joint_ = std::make_unique<FloatingJointModel>();
        else link->joint_ = std::make_unique<FixedJointModel>();
        link->joint_->name_ = joint_name;
        link->joint_->type_ = type;
        link->joint_->axis_ = axis;
        if (type == JointModel::PLANAR) {
            VariableBounds theta; // MoveIt: theta is not position-bounded
            theta.min_position_ = -3.14159265358979323846;
            theta.max_position_ = 3.14159265358979323846;
            theta.max_velocity_ = bounds.max_velocity_;
            link->joint_->bounds_ = {bounds, bounds, theta};
        } else if (type == JointModel::FLOATING) {
            VariableBounds rot; // MoveIt: the quaternion components live in [-1, 1]
            rot.position_bounded_ = true;
            rot.min_position_ = -1.0;
            rot.max_position_ = 1.0;
            rot.max_velocity_ = bounds.max_velocity_;
            link->joint_->bounds_ = {bounds, bounds, bounds, rot, rot, rot, rot};
        } else if (type != JointModel::FIXED) {
            link->joint_->bounds_ = {bounds};
        }
        links_.push_back(std::move(link));
        return links_.back().get();
    }
    JointModelGroup* add_group(std::string const& name, std::vector<std::string> const& joint_names) {
        auto g = std::make_unique<JointModelGroup>();
        g->name_ = name;
        for (auto const& jn : joint_names)
            for (auto const& l : links_)
                if (l->joint_->getName() == jn) {
                    g->joints_.push_back(l->joint_.get());
                    if (l->joint_->getType() != JointModel::FIXED && !l->joint_->getMimic()) g->active_.push_back(l->joint_.get());
                }
        groups_.push_back(std::move(g));
        return groups_.back().get();
    }
    std::vector<std::unique_ptr<LinkModel>> links_;
    std::vector<std::unique_ptr<JointModelGroup>> groups_;
};
using RobotModelConstPtr = std::shared_ptr<RobotModel const>;
} // namespace moveit::core

// Stand-in for include/robotoc/robot/se3.hpp (which is `using SE3 = pinocchio::SE3`; Pinocchio is absent).
// TEST INFRASTRUCTURE ONLY (oracle/_ref).  Only what ContactStatus / ImpactStatus store: a rotation and a translation.
#ifndef ROBOTOC_SE3_HPP_
#define ROBOTOC_SE3_HPP_
#include "Eigen/Core"
namespace robotoc {
class SE3 {
 public:
  SE3() : R_(Eigen::Matrix3d::Identity()), p_(Eigen::Vector3d::Zero()) {}
  SE3(const Eigen::Matrix3d& R, const Eigen::Vector3d& p) : R_(R), p_(p) {}
  static SE3 Identity() { return SE3(); }
  const Eigen::Matrix3d& rotation() const { return R_; }
  const Eigen::Vector3d& translation() const { return p_; }
  Eigen::Matrix3d& rotation() { return R_; }
  Eigen::Vector3d& translation() { return p_; }
  void setRandom() { R_.setIdentity(); p_.setRandom(); }
  static SE3 Random() { SE3 s; s.setRandom(); return s; }
  bool isApprox(const SE3& o, double prec = 1e-12) const { return R_.isApprox(o.R_, prec) && p_.isApprox(o.p_, prec); }
  friend std::ostream& operator<<(std::ostream& os, const SE3& s) { return os << s.R_ << "\n" << s.p_.transpose(); }
 private:
  Eigen::Matrix3d R_;
  Eigen::Vector3d p_;
};
}  // namespace robotoc
#endif

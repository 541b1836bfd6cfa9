// Stand-in for include/robotoc/robot/robot.hpp.  TEST INFRASTRUCTURE ONLY (oracle/_ref).
//
// The real Robot wraps pinocchio::Model / Data (third-party, absent from the image).  The reference's Riccati,
// core and dynamics sources use it for (a) dimensions and (b) a handful of rigid-body routines.  This stand-in
// provides (a) exactly, computeMJtJinv (robot.hxx:642-684) as the same formula on dense LLT factorisations (the
// reference uses Pinocchio's sparse Cholesky of M: same matrix, different elimination order), the Euclidean
// configuration arithmetic of fixed-base robots, and ABORTS in everything that needs Pinocchio's kinematics
// (RNEA, Baumgarte, frame Jacobians, SE3 integration): the linearize* halves of the dynamics sources compile but
// are never called by oracle/ref_shim/ref_capi.cpp.
#ifndef ROBOTOC_ROBOT_HPP_
#define ROBOTOC_ROBOT_HPP_

#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "Eigen/Core"
#include "Eigen/LU"

#include "robotoc/robot/contact_model_info.hpp"
#include "robotoc/robot/robot_model_info.hpp"
#include "robotoc/robot/se3.hpp"
#include "robotoc/robot/contact_status.hpp"
#include "robotoc/robot/impact_status.hpp"
#include "robotoc/utils/aligned_vector.hpp"

namespace robotoc {

class Robot {
 public:
  using Vector6d = Eigen::Matrix<double, 6, 1>;

  // dimv generalized velocities, dimu actuated joints, one ContactType per contact frame
  Robot(const int dimv, const int dimu, const std::vector<ContactType>& contact_types,
        const double contact_inv_damping = 0.0)
      : dimv_(dimv), dimu_(dimu), contact_types_(contact_types), contact_inv_damping_(contact_inv_damping), max_dimf_(0) {
    for (const auto t : contact_types_) max_dimf_ += (t == ContactType::PointContact) ? 3 : 6;
    for (size_t i = 0; i < contact_types_.size(); ++i) contact_frame_names_.push_back("contact_" + std::to_string(i));
  }
  Robot() : dimv_(0), dimu_(0), contact_inv_damping_(0.0), max_dimf_(0) {}

  int dimq() const { return hasFloatingBase() ? dimv_ + 1 : dimv_; }
  int dimv() const { return dimv_; }
  int dimu() const { return dimu_; }
  int max_dimf() const { return max_dimf_; }
  int dim_passive() const { return dimv_ - dimu_; }
  bool hasFloatingBase() const { return dimv_ != dimu_; }
  int maxNumContacts() const { return (int)contact_types_.size(); }
  int maxNumPointContacts() const {
    int n = 0;
    for (const auto t : contact_types_) n += (t == ContactType::PointContact);
    return n;
  }
  int maxNumSurfaceContacts() const { return maxNumContacts() - maxNumPointContacts(); }
  ContactType contactType(const int i) const { return contact_types_[i]; }
  std::vector<ContactType> contactTypes() const { return contact_types_; }
  std::vector<std::string> contactFrameNames() const { return contact_frame_names_; }
  ContactStatus createContactStatus() const { return ContactStatus(contact_types_, contact_frame_names_); }
  ImpactStatus createImpactStatus() const { return ImpactStatus(contact_types_, contact_frame_names_); }

  // Robot::computeMJtJinv (include/robotoc/robot/robot.hxx:642-684): inverse of [[M, J^T], [J, 0]]
  //   Minv, JMinvJt = J Minv J^T (+ contact_inv_damping on its diagonal), its inverse by LLT, then the blocks
  //   [[Minv - Minv J^T S^-1 J Minv,  Minv J^T S^-1], [S^-1 J Minv, -S^-1]].
  template <typename MatrixType1, typename MatrixType2, typename MatrixType3>
  void computeMJtJinv(const Eigen::MatrixBase<MatrixType1>& M, const Eigen::MatrixBase<MatrixType2>& J,
                      const Eigen::MatrixBase<MatrixType3>& MJtJinv) {
    const int dimv = dimv_, dimf = (int)J.rows();
    Eigen::MatrixBase<MatrixType3>& out = const_cast<Eigen::MatrixBase<MatrixType3>&>(MJtJinv);
    Eigen::LLT<Eigen::MatrixXd> lltM(M);
    const Eigen::MatrixXd Minv = lltM.solve(Eigen::MatrixXd::Identity(dimv, dimv));
    if (dimf == 0) {
      out.topLeftCorner(dimv, dimv) = Minv;
      return;
    }
    const Eigen::MatrixXd MinvJt = lltM.solve(Eigen::MatrixXd(J.transpose()));
    Eigen::MatrixXd S = J * MinvJt;
    for (int i = 0; i < dimf; ++i) S(i, i) += contact_inv_damping_;
    Eigen::LLT<Eigen::MatrixXd> lltS(S);
    const Eigen::MatrixXd Sinv = lltS.solve(Eigen::MatrixXd::Identity(dimf, dimf));
    const Eigen::MatrixXd SinvJMinv = Sinv * MinvJt.transpose();
    out.topLeftCorner(dimv, dimv) = Minv - MinvJt * SinvJMinv;
    out.topRightCorner(dimv, dimf) = SinvJMinv.transpose();
    out.bottomLeftCorner(dimf, dimv) = SinvJMinv;
    out.bottomRightCorner(dimf, dimf) = -Sinv;
  }

  // fixed-base (Euclidean) configuration arithmetic; floating bases need Pinocchio's SE3 maps
  template <typename A, typename B, typename C>
  void subtractConfiguration(const Eigen::MatrixBase<A>& qf, const Eigen::MatrixBase<B>& q0,
                             const Eigen::MatrixBase<C>& qdiff) const {
    if (hasFloatingBase()) {
      const_cast<Eigen::MatrixBase<C>&>(qdiff) = pop("subtractConfiguration");   // injected, in call order
      return;
    }
    const_cast<Eigen::MatrixBase<C>&>(qdiff) = qf - q0;
  }
  template <typename A, typename C>
  void integrateConfiguration(const Eigen::MatrixBase<A>& v, const double h, const Eigen::MatrixBase<C>& q) const {
    if (hasFloatingBase()) {
      const_cast<Eigen::MatrixBase<C>&>(q) = pop("integrateConfiguration");   // injected
      return;
    }
    const_cast<Eigen::MatrixBase<C>&>(q) += h * v;
  }
  template <typename A>
  void normalizeConfiguration(const Eigen::MatrixBase<A>&) const {}

  // ---- what the constraint components read (src/constraints/*.cpp): joint limits, and the frame kinematics of the contact
  //      frames -- INJECTED by the test (frame rotation in the world, LOCAL frame Jacobian 6 x dimv, i.e. what
  //      pinocchio::getFrameJacobian(..., LOCAL, ...) returns), so that the reference's composition of them is what runs ----
  void setJointLimits(const Eigen::VectorXd& q_min, const Eigen::VectorXd& q_max, const Eigen::VectorXd& v_max, const Eigen::VectorXd& u_max) {
    q_min_ = q_min, q_max_ = q_max, v_max_ = v_max, u_max_ = u_max;
  }
  Eigen::VectorXd lowerJointPositionLimit() const { return q_min_; }
  Eigen::VectorXd upperJointPositionLimit() const { return q_max_; }
  Eigen::VectorXd jointVelocityLimit() const { return v_max_; }
  Eigen::VectorXd jointEffortLimit() const { return u_max_; }
  std::vector<int> contactFrames() const {
    std::vector<int> f;
    for (int i = 0; i < maxNumContacts(); ++i) f.push_back(i);   // frame id = contact index
    return f;
  }
  void setFrameKinematics(const int frame_id, const Eigen::Matrix3d& R_world, const Eigen::MatrixXd& J_local) {
    if ((int)frame_R_.size() <= frame_id) frame_R_.resize(frame_id + 1), frame_J_.resize(frame_id + 1);
    frame_R_[frame_id] = R_world, frame_J_[frame_id] = J_local;
  }
  template <typename A>
  void updateFrameKinematics(const Eigen::MatrixBase<A>&) {}   // the injected kinematics stand
  const Eigen::Matrix3d& frameRotation(const int frame_id) const { return frame_R_.at(frame_id); }
  template <typename MatrixType>
  void getFrameJacobian(const int frame_id, const Eigen::MatrixBase<MatrixType>& J) {
    const_cast<Eigen::MatrixBase<MatrixType>&>(J) = frame_J_.at(frame_id);
  }
  // robot.hxx:264-271
  template <typename Vector3dType>
  void transformFromLocalToWorld(const int frame_id, const Eigen::Vector3d& vec_local, const Eigen::MatrixBase<Vector3dType>& vec_world) const {
    const_cast<Eigen::MatrixBase<Vector3dType>&>(vec_world).noalias() = frameRotation(frame_id) * vec_local;
  }
  // robot.hxx:274-287, statement for statement
  template <typename Vector3dType, typename MatrixType>
  void getJacobianTransformFromLocalToWorld(const int frame_id, const Eigen::MatrixBase<Vector3dType>& vec_world,
                                            const Eigen::MatrixBase<MatrixType>& J) {
    const_cast<Eigen::MatrixBase<MatrixType>&>(J).setZero();
    getFrameJacobian(frame_id, const_cast<Eigen::MatrixBase<MatrixType>&>(J));
    for (int i = 0; i < dimv_; ++i) {
      const_cast<Eigen::MatrixBase<MatrixType>&>(J).template topRows<3>().col(i).noalias() =
          J.template bottomRows<3>().col(i).cross(vec_world.template head<3>());
    }
  }

  // ---- inverse dynamics and its partial derivatives: INJECTED per call by the test (computed by this repository's CPU
  //      restatement at the same (q, v, a)), so that the reference's composition around them is what runs ----
  void setInverseDynamics(const Eigen::VectorXd& ID, const Eigen::MatrixXd& dIDdq, const Eigen::MatrixXd& dIDdv, const Eigen::MatrixXd& dIDda) {
    id_ = ID, did_dq_ = dIDdq, did_dv_ = dIDdv, did_da_ = dIDda;
    has_id_ = true;
  }
  // the injected quantities stand.  Horizon mode (ref_ocp_capi.cpp): every stage's evalKKT opens with updateKinematics(q, v[, a])
  // -- that call pops the stage's frame kinematics (one matrix per contact: R row-major 3 x 3 stacked on the 6 x nv LOCAL
  // Jacobian, i.e. (3 + 6) ... packed as [9 + 6 nv] x 1); the one-argument call of the switching constraint pops nothing
  template <typename A>
  void updateKinematics(const Eigen::MatrixBase<A>&) {}
  template <typename A, typename B, typename... Rest>
  void updateKinematics(const Eigen::MatrixBase<A>&, const Eigen::MatrixBase<B>&, const Rest&...) {
    auto it = fifo_->find("frames");
    if (it == fifo_->end() || it->second.empty()) return;
    for (int c = 0; c < maxNumContacts(); ++c) {
      const Eigen::MatrixXd m = pop("frames");
      Eigen::Matrix3d R;
      for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) R(r, k) = m(3 * r + k, 0);
      Eigen::MatrixXd J(6, dimv_);
      for (int j = 0; j < dimv_; ++j)
        for (int r = 0; r < 6; ++r) J(r, j) = m(9 + r + 6 * j, 0);
      setFrameKinematics(c, R, J);
    }
  }
  bool queued(const char* key) const {
    auto it = fifo_->find(key);
    return it != fifo_->end() && !it->second.empty();
  }
  template <typename A, typename B, typename C, typename D>
  void RNEA(const Eigen::MatrixBase<A>&, const Eigen::MatrixBase<B>&, const Eigen::MatrixBase<C>&, const Eigen::MatrixBase<D>& tau) {
    if (queued("ID")) {
      const_cast<Eigen::MatrixBase<D>&>(tau) = pop("ID");
      return;
    }
    if (!has_id_) unavailable("RNEA");
    const_cast<Eigen::MatrixBase<D>&>(tau) = id_;
  }
  template <typename A, typename B, typename C, typename D, typename E, typename F>
  void RNEADerivatives(const Eigen::MatrixBase<A>&, const Eigen::MatrixBase<B>&, const Eigen::MatrixBase<C>&,
                       const Eigen::MatrixBase<D>& dq, const Eigen::MatrixBase<E>& dv, const Eigen::MatrixBase<F>& da) {
    if (queued("dIDdq")) {
      const_cast<Eigen::MatrixBase<D>&>(dq) = pop("dIDdq");
      const_cast<Eigen::MatrixBase<E>&>(dv) = pop("dIDdv");
      const_cast<Eigen::MatrixBase<F>&>(da) = pop("dIDda");
      return;
    }
    if (!has_id_) unavailable("RNEADerivatives");
    const_cast<Eigen::MatrixBase<D>&>(dq) = did_dq_;
    const_cast<Eigen::MatrixBase<E>&>(dv) = did_dv_;
    const_cast<Eigen::MatrixBase<F>&>(da) = did_da_;
  }

  // ---- floating-base / contact quantities as FIFO injections: the test pushes, in the order the reference's stage code
  //      asks for them, what this repository's CPU restatement computes (inject(name, matrix)); each call pops one ----
  void inject(const std::string& key, const Eigen::MatrixXd& m) { (*fifo_)[key].push_back(m); }
  void clearInjections() { fifo_->clear(); }
  size_t pendingOf(const std::string& key) const { return (*fifo_)[key].size(); }
  size_t pending() const {
    size_t n = 0;
    for (const auto& kv : *fifo_) n += kv.second.size();
    return n;
  }
  Eigen::MatrixXd pop(const char* key) const {
    auto& q = (*fifo_)[key];
    if (q.empty()) unavailable(key);
    Eigen::MatrixXd m = q.front();
    q.pop_front();
    return m;
  }
#define RTOC_POP1(name, key)                                                         \
  template <typename A, typename B, typename C>                                      \
  void name(const A&, const B&, const Eigen::MatrixBase<C>& out) const {             \
    const_cast<Eigen::MatrixBase<C>&>(out) = pop(key);                               \
  }
  RTOC_POP1(dSubtractConfiguration_dqf, "dSubtractConfiguration_dqf")
  RTOC_POP1(dSubtractConfiguration_dq0, "dSubtractConfiguration_dq0")
#undef RTOC_POP1
  template <typename... Args>
  void setContactForces(const Args&...) {}
  template <typename A, typename C>
  void computeBaumgarteResidual(const A&, const Eigen::MatrixBase<C>& res) const { const_cast<Eigen::MatrixBase<C>&>(res) = pop("baumgarteResidual"); }
  template <typename A, typename B, typename C, typename D>
  void computeBaumgarteDerivatives(const A&, const Eigen::MatrixBase<B>& dq, const Eigen::MatrixBase<C>& dv, const Eigen::MatrixBase<D>& da) {
    const_cast<Eigen::MatrixBase<B>&>(dq) = pop("baumgarte_dq");
    const_cast<Eigen::MatrixBase<C>&>(dv) = pop("baumgarte_dv");
    const_cast<Eigen::MatrixBase<D>&>(da) = pop("baumgarte_da");
  }

  // switching constraint (src/dynamics/switching_constraint.cpp): the integrated configuration is not used by the stand-in;
  // position residual / derivative of the impacting contacts at it are injected.  dIntegrateTransport restates the wrapper
  // of robot.hxx:59-92 on top of Pinocchio's documented semantics (dIntegrateTransport left-multiplies by dIntegrate): the
  // reference passes the TRANSPOSED Jacobian, so Jout^T = dIntegrate Jin^T.  dIntegrate_dq / _dv themselves are injected.
  template <typename A, typename B, typename C>
  void integrateConfiguration(const Eigen::MatrixBase<A>& q, const Eigen::MatrixBase<B>& v, const double h, const Eigen::MatrixBase<C>& q_int) const {
    if (hasFloatingBase()) {
      const_cast<Eigen::MatrixBase<C>&>(q_int) = pop("integrateConfiguration");
      return;
    }
    const_cast<Eigen::MatrixBase<C>&>(q_int) = q + h * v;
  }
  template <typename A, typename C>
  void computeContactPositionResidual(const A&, const Eigen::MatrixBase<C>& res) const { const_cast<Eigen::MatrixBase<C>&>(res) = pop("contactPositionResidual"); }
  template <typename A, typename C>
  void computeContactPositionDerivative(const A&, const Eigen::MatrixBase<C>& J) const { const_cast<Eigen::MatrixBase<C>&>(J) = pop("contactPositionDerivative"); }
  template <typename A, typename B, typename C, typename D>
  void dIntegrateTransport_dq(const Eigen::MatrixBase<A>&, const Eigen::MatrixBase<B>&, const Eigen::MatrixBase<C>& Jin, const Eigen::MatrixBase<D>& Jout) const {
    const Eigen::MatrixXd T = pop("dIntegrate_dq");
    const_cast<Eigen::MatrixBase<D>&>(Jout) = Eigen::MatrixXd(T * Eigen::MatrixXd(Jin.transpose())).transpose();
  }
  template <typename A, typename B, typename C, typename D>
  void dIntegrateTransport_dv(const Eigen::MatrixBase<A>&, const Eigen::MatrixBase<B>&, const Eigen::MatrixBase<C>& Jin, const Eigen::MatrixBase<D>& Jout) const {
    const Eigen::MatrixXd T = pop("dIntegrate_dv");
    const_cast<Eigen::MatrixBase<D>&>(Jout) = Eigen::MatrixXd(T * Eigen::MatrixXd(Jin.transpose())).transpose();
  }

  // impact variants: the same injected inverse dynamics (dID/dv slot = dID/d(dv)), FIFO for the contact-velocity rows
  template <typename... Args>
  void setImpactForces(const Args&...) {}
  template <typename A, typename B, typename D>
  void RNEAImpact(const Eigen::MatrixBase<A>&, const Eigen::MatrixBase<B>&, const Eigen::MatrixBase<D>& res) {
    if (queued("ID")) {
      const_cast<Eigen::MatrixBase<D>&>(res) = pop("ID");
      return;
    }
    if (!has_id_) unavailable("RNEAImpact");
    const_cast<Eigen::MatrixBase<D>&>(res) = id_;
  }
  template <typename A, typename B, typename D, typename E>
  void RNEAImpactDerivatives(const Eigen::MatrixBase<A>&, const Eigen::MatrixBase<B>&, const Eigen::MatrixBase<D>& dq, const Eigen::MatrixBase<E>& ddv) {
    if (queued("dIDdq")) {
      const_cast<Eigen::MatrixBase<D>&>(dq) = pop("dIDdq");
      (void)pop("dIDdv");
      const_cast<Eigen::MatrixBase<E>&>(ddv) = pop("dIDda");
      return;
    }
    if (!has_id_) unavailable("RNEAImpactDerivatives");
    const_cast<Eigen::MatrixBase<D>&>(dq) = did_dq_;
    const_cast<Eigen::MatrixBase<E>&>(ddv) = did_da_;
  }
  template <typename A, typename C>
  void computeImpactVelocityResidual(const A&, const Eigen::MatrixBase<C>& res) const { const_cast<Eigen::MatrixBase<C>&>(res) = pop("impactVelocityResidual"); }
  template <typename A, typename B, typename C>
  void computeImpactVelocityDerivatives(const A&, const Eigen::MatrixBase<B>& dq, const Eigen::MatrixBase<C>& dv) {
    const_cast<Eigen::MatrixBase<B>&>(dq) = pop("impactVelocity_dq");
    const_cast<Eigen::MatrixBase<C>&>(dv) = pop("impactVelocity_dv");
  }

  // ---- everything below needs Pinocchio: present so that the reference sources compile, never called ----
#define RTOC_NEEDS_PINOCCHIO(name)                    \
  template <typename... Args>                         \
  void name(const Args&...) const {                   \
    unavailable(#name);                               \
  }
#undef RTOC_NEEDS_PINOCCHIO

 private:
  static void unavailable(const char* what) {
    std::fprintf(stderr, "oracle/_ref: Robot::%s needs Pinocchio, which is absent from this image\n", what);
    std::abort();
  }
  void needFixedBase(const char* what) const {
    if (hasFloatingBase()) unavailable(what);
  }
  int dimv_, dimu_;
  Eigen::VectorXd q_min_, q_max_, v_max_, u_max_;
  Eigen::VectorXd id_;
  Eigen::MatrixXd did_dq_, did_dv_, did_da_;
  bool has_id_ = false;
  std::shared_ptr<std::map<std::string, std::deque<Eigen::MatrixXd>>> fifo_ = std::make_shared<std::map<std::string, std::deque<Eigen::MatrixXd>>>();
  std::vector<Eigen::Matrix3d> frame_R_;
  std::vector<Eigen::MatrixXd> frame_J_;
  std::vector<ContactType> contact_types_;
  std::vector<std::string> contact_frame_names_;
  double contact_inv_damping_;
  int max_dimf_;
};

}  // namespace robotoc
#endif  // ROBOTOC_ROBOT_HPP_

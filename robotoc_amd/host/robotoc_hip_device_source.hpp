// robotoc_hip_device_source.hpp -- a StageDataSource that linearises ON THE DEVICE.
//
// The OCPSolver shell (robotoc_hip_solver.hpp) takes the Pinocchio / cost half of evalKKT from a StageDataSource.  For an
// OCP whose cost is a ConfigurationSpaceCost and whose Constraints object holds joint limits and friction cones
// (examples/anymal/trot.cpp:131-146), that half exists on the device as well (include/rtoc_robot.h: rtoc_contact_eval_kkt =
// cost + inequality rows + state equation on the manifold + rigid-body linearisation + switching constraints): with this
// source nothing of OCPSolver::updateSolution (src/solver/ocp_solver.cpp:111-145) runs on the host -- the solution stays
// resident in RTOC_BUF_SOL, the host sees KKT errors and, on request, the iterate.
#ifndef ROBOTOC_HIP_DEVICE_SOURCE_HPP_
#define ROBOTOC_HIP_DEVICE_SOURCE_HPP_

#include "../../include/rtoc_robot.h"
#include "robotoc_hip_solver.hpp"

namespace robotoc {

class ConfigurationCostSource : public StageDataSource {
 public:
  // grid: the discretisation (TimeDiscretization of the contact sequence); active[i]: contact mask of grid point i;
  // contact_positions: [grid point][contact][3]; s0: initial guess
  ConfigurationCostSource(const rtoc_robot_model& model, const rtoc_configuration_cost& cost, const std::vector<GridInfo>& grid,
                          const std::vector<unsigned>& active, const std::vector<double>& contact_positions, const Solution& s0)
      : model_(model), cost_(cost), td_(grid), active_(active), cpos_(contact_positions), s0_(s0) {
    if (active.size() != grid.size() || contact_positions.size() != grid.size() * static_cast<size_t>(model.ncontacts) * 3)
      throw std::invalid_argument("[ConfigurationCostSource] one contact mask and ncontacts positions per grid point");
  }
  // The same OCP described like the reference describes it: contact sequence + horizon (OCP::contact_sequence, T, N) instead of
  // a ready-made grid, optionally with STOConstraints (an OCP with an STO problem: PhaseBased discretisation, ocp_solver.cpp:46-48).
  // The source then owns the discretisation: OCPSolver::discretize(t) re-runs TimeDiscretization::discretize at the sequence's
  // current event times (mesh refinement).  s0: initial guess over the N + 1 + lifts + 2 impacts grid points of discretize(t = 0).
  ConfigurationCostSource(const rtoc_robot_model& model, const rtoc_configuration_cost& cost, const ContactSequence& contact_sequence,
                          const double T, const int N, const Solution& s0, const std::shared_ptr<STOConstraints>& sto_constraints = nullptr)
      : model_(model), cost_(cost), s0_(s0), cs_(new ContactSequence(contact_sequence)), sto_(sto_constraints), T_(T), N_(N) {
    if (!(T > 0.0)) throw std::out_of_range("[OCPSolver] invalid argument: ocp.T must be positive!");
    if (N <= 0) throw std::out_of_range("[OCPSolver] invalid argument: ocp.N must be positive!");
    if (contact_sequence.numContacts() != model.ncontacts) throw std::invalid_argument("[ConfigurationCostSource] the contact sequence is for another robot");
    discretize(0.0);
  }
  ContactSequence* contactSequence() override { return cs_.get(); }
  const STOConstraints* stoConstraints() const override { return sto_.get(); }
  double horizonLength() const override { return T_; }
  const std::vector<unsigned>* contactMasks() const override { return &active_; }
  int maxGridPoints() const override {
    return cs_ ? N_ + 1 + cs_->numLiftEvents() + 2 * cs_->numImpactEvents() : td_.size();
  }
  bool discretize(const double t) override {
    if (!cs_) return false;
    td_ = robotoc::discretize(*cs_, T_, N_, t, static_cast<bool>(sto_));
    contactSchedule(*cs_, td_, active_, cpos_, &crot_);
    scheduled_ = false;
    return true;
  }
  RobotDims robot() const override {
    const bool ff = model_.type[0] == RTOC_JOINT_FREE_FLYER;
    int dimf = 0;
    for (int k = 0; k < model_.ncontacts; ++k) dimf += model_.contact_type[k] == RTOC_CONTACT_SURFACE ? 6 : 3;
    RobotDims r;
    r.dimv = model_.nv, r.dimu = ff ? model_.nv - 6 : model_.nv, r.dim_passive = ff ? 6 : 0, r.max_dimf = dimf;
    return r;
  }
  // ---- the Constraints object (before the solver is constructed) ----
  // JointPosition{Lower,Upper}Limit, JointVelocity{Lower,Upper}Limit, JointTorques{Lower,Upper}Limit in the order
  // examples/anymal/trot.cpp:134-146 adds them; every vector has dimu entries (the actuated joints)
  void setJointLimits(const std::vector<double>& q_min, const std::vector<double>& q_max, const std::vector<double>& v_max,
                      const std::vector<double>& u_max) {
    const RobotDims r = robot();
    const size_t nu = static_cast<size_t>(r.dimu);
    if (q_min.size() != nu || q_max.size() != nu || v_max.size() != nu || u_max.size() != nu)
      throw std::invalid_argument("[ConfigurationCostSource] joint limits: dimu entries each");
    rows_.clear(), bounds_.clear();
    const int vars[3] = {RTOC_VAR_Q, RTOC_VAR_V, RTOC_VAR_U}, levels[3] = {2, 1, 0};
    for (int k = 0; k < 3; ++k)
      for (int sign = -1; sign <= 1; sign += 2)
        for (int j = 0; j < r.dimu; ++j) {
          rtoc_box_row w;
          w.var = vars[k], w.index = k == 2 ? j : r.dim_passive + j, w.sign = sign, w.level = levels[k];
          rows_.push_back(w);
          // g = sign z - bound <= 0: upper z - z_max, lower z_min - z
          bounds_.push_back(k == 0 ? (sign > 0 ? q_max[j] : -q_min[j]) : (k == 1 ? v_max[j] : u_max[j]));
        }
  }
  // FrictionCone with ContactStatus::frictionCoefficient mu[k]; impact_cone: ImpactFrictionCone as well
  void setFrictionCone(const std::vector<double>& mu, const bool impact_cone = false) {
    if (mu.size() != static_cast<size_t>(model_.ncontacts)) throw std::invalid_argument("[ConfigurationCostSource] one friction coefficient per contact");
    mu_ = mu, impact_cone_ = impact_cone;
  }
  void setBarrierParam(const double barrier_param, const double fraction_to_boundary_rule) {
    barrier_ = barrier_param, ftb_ = fraction_to_boundary_rule;
  }
  int ncMax() const override {
    const int n = static_cast<int>(rows_.size()) + (mu_.empty() ? 0 : 5 * model_.ncontacts);
    return (n + 7) & ~7;
  }
  const TimeDiscretization& timeDiscretization() const override { return td_; }
  void configure(rtoc_ctx* ctx) override {
    chk(rtoc_set_robot_model(ctx, &model_), "rtoc_set_robot_model");
    chk(rtoc_set_configuration_cost(ctx, &cost_), "rtoc_set_configuration_cost");
    if (!rows_.empty()) {
      chk(rtoc_set_constraint_rows(ctx, rows_.data(), static_cast<int>(rows_.size())), "rtoc_set_constraint_rows");
      chk(rtoc_set_constraint_bounds(ctx, bounds_.data(), static_cast<int>(bounds_.size()), barrier_, ftb_), "rtoc_set_constraint_bounds");
    }
    if (!mu_.empty()) {
      const int cd = model_.contact_type[0] == RTOC_CONTACT_SURFACE ? 6 : 3;
      chk(rtoc_set_friction_cones(ctx, model_.ncontacts, cd), "rtoc_set_friction_cones");
      chk(rtoc_set_friction_coefficients(ctx, mu_.data(), model_.ncontacts), "rtoc_set_friction_coefficients");
      chk(rtoc_set_option(ctx, RTOC_OPT_IMPACT_CONES, impact_cone_ ? 1 : 0), "rtoc_set_option");
      chk(rtoc_set_barrier_param(ctx, barrier_, ftb_), "rtoc_set_barrier_param");
    }
    scheduled_ = false;
  }
  void initConstraints(rtoc_ctx* ctx, const Solution&) override {
    schedule(ctx);
    if (!rows_.empty() || !mu_.empty()) chk(rtoc_contact_init_constraints(ctx), "rtoc_contact_init_constraints");
  }
  void setInitialState(rtoc_ctx* ctx, const Vec& q, const Vec& v) override {
    std::vector<double> x0(q.size() + v.size());
    for (int i = 0; i < q.size(); ++i) x0[i] = q(i);
    for (int i = 0; i < v.size(); ++i) x0[q.size() + i] = v(i);
    chk(rtoc_set_initial_state(ctx, x0.data(), 1), "rtoc_set_initial_state");
  }
  void linearize(rtoc_ctx* ctx, const Solution&) override {
    schedule(ctx);
    chk(rtoc_contact_eval_kkt(ctx), "rtoc_contact_eval_kkt");
  }
  // computed on the device with the linearisation (state_equation.cpp:99-109, Fqq_prev_inv correction included)
  bool initialStateDirectionOnDevice() const override { return true; }
  bool needsHostSolution() const override { return false; }
  void initialStateDirection(const Vec&, const Vec&, const Solution&, Vec&) const override {}
  void initialSolution(Solution& s) const override {
    for (size_t i = 0; i < s.size() && i < s0_.size(); ++i) s[i] = s0_[i];
  }

 private:
  void schedule(rtoc_ctx* ctx) {
    if (scheduled_) return;  // needs the grid, which the solver sets after configure()
    chk(rtoc_set_contact_schedule(ctx, active_.data(), cpos_.data(), crot_.empty() ? nullptr : crot_.data()), "rtoc_set_contact_schedule");
    scheduled_ = true;
  }
  static void chk(const int rc, const char* what) {
    if (rc != RTOC_OK) throw std::runtime_error(std::string("[ConfigurationCostSource] ") + what + ": " + rtoc_error_string(rc));
  }
  rtoc_robot_model model_;
  rtoc_configuration_cost cost_;
  TimeDiscretization td_;
  std::vector<unsigned> active_;
  std::vector<double> cpos_;
  std::vector<double> crot_;   // [grid point][contact][9] or empty (surface contacts: ContactSequence rotations)
  Solution s0_;
  std::vector<rtoc_box_row> rows_;
  std::vector<double> bounds_, mu_;
  bool impact_cone_ = false;
  double barrier_ = 1.0e-3, ftb_ = 0.995;
  bool scheduled_ = false;
  std::shared_ptr<ContactSequence> cs_;
  std::shared_ptr<STOConstraints> sto_;
  double T_ = 0.0;
  int N_ = 0;
};

}  // namespace robotoc
#endif

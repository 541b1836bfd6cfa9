// ref_contact_stage_capi.cpp -- IntermediateStage::evalKKT (src/ocp/intermediate_stage.cpp:84-148) of a floating-base robot in
// contact, run by the REFERENCE'S OWN sources.  TEST INFRASTRUCTURE ONLY (oracle/_ref/librtoc_ref.so, oracle/Makefile.ref).
//
// Reference code that runs here: IntermediateStage, ContactSequence / ContactStatus, CostFunction + ConfigurationSpaceCost (the
// SE(3) branch), Constraints (joint limits + FrictionCone), linearizeStateEquation / correctLinearizeStateEquation with
// SE3JacobianInverse, linearizeContactDynamics / condenseContactDynamics, the evalKKT tail scalings.  NOT reference code: Eigen
// (mini_eigen.hpp) and every Pinocchio quantity -- configuration differences and their Jacobians, inverse dynamics with
// contact forces and its partial derivatives, Baumgarte residual and derivatives, frame rotations and LOCAL Jacobians -- which
// the caller injects (computed by this repository's CPU rigid-body restatement), in the order the stage asks for them.
#include <memory>
#include <vector>

#include "robotoc/constraints/constraints.hpp"
#include "robotoc/constraints/contact_wrench_cone.hpp"
#include "robotoc/constraints/friction_cone.hpp"
#include "robotoc/constraints/joint_position_lower_limit.hpp"
#include "robotoc/constraints/joint_position_upper_limit.hpp"
#include "robotoc/constraints/joint_torques_lower_limit.hpp"
#include "robotoc/constraints/joint_torques_upper_limit.hpp"
#include "robotoc/constraints/joint_velocity_lower_limit.hpp"
#include "robotoc/constraints/joint_velocity_upper_limit.hpp"
#include "robotoc/cost/configuration_space_cost.hpp"
#include "robotoc/cost/cost_function.hpp"
#include "robotoc/ocp/impact_stage.hpp"
#include "robotoc/ocp/intermediate_stage.hpp"
#include "robotoc/ocp/terminal_stage.hpp"
#include "robotoc/planner/contact_sequence.hpp"

using namespace robotoc;

namespace {
std::unique_ptr<Robot> g_robot;
Eigen::MatrixXd mat(const double* p, int r, int c) {   // column-major in
  Eigen::MatrixXd m(r, c);
  for (int j = 0; j < c; ++j)
    for (int i = 0; i < r; ++i) m(i, j) = p[i + (size_t)j * r];
  return m;
}
Eigen::VectorXd vec(const double* p, int n) {
  Eigen::VectorXd v(n);
  for (int i = 0; i < n; ++i) v(i) = p[i];
  return v;
}
}  // namespace

extern "C" {

int ref_stage_begin(int nv, int nu, int ncontacts) {
  g_robot.reset(new Robot(nv, nu, std::vector<ContactType>(ncontacts, ContactType::PointContact)));
  return 0;
}
int ref_stage_begin_surface(int nv, int nu, int ncontacts) {
  g_robot.reset(new Robot(nv, nu, std::vector<ContactType>(ncontacts, ContactType::SurfaceContact)));
  return 0;
}
int ref_stage_inject(const char* key, const double* data, int rows, int cols) {
  if (!g_robot) return 1;
  g_robot->inject(key, mat(data, rows, cols));
  return 0;
}
int ref_stage_frame(int c, const double* R_rowmajor, const double* J_local /*6 x nv col-major*/) {
  if (!g_robot) return 1;
  Eigen::Matrix3d R;
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) R(r, k) = R_rowmajor[3 * r + k];
  g_robot->setFrameKinematics(c, R, mat(J_local, 6, g_robot->dimv()));
  return 0;
}
int ref_stage_inverse_dynamics(const double* ID, const double* dq, const double* dv, const double* da) {
  if (!g_robot) return 1;
  const int nv = g_robot->dimv();
  g_robot->setInverseDynamics(vec(ID, nv), mat(dq, nv, nv), mat(dv, nv, nv), mat(da, nv, nv));
  return 0;
}

// cost: [12][nv + 1] = q_ref (nq), v_ref, u_ref, q / v / a / u weights, terminal q / v weights, impact q / v / dv weights
// sol / sol_next: q (nq), v, a, u (nu), f ([ncontacts][3] by contact index), lmd, gmm, beta, mu ([ncontacts][3]), nu_passive (6)
// slack / dual: rows of the six joint-limit components (nu each), then 5 per contact by contact index
// out: Qxx (nx^2), Qxu (nx nu), Quu (nu^2), Fxx (nx^2), Fvu (nv nu), lx (nx), lu (nu), Fx (nx), hx (nx), hu (nu), fx (nx),
//      [Qtt, Qtt_prev, h, kkt_error] -- all column-major, after IntermediateStage::evalKKT
int ref_contact_stage_eval_kkt(unsigned active, const double* contact_pos, const double* mu, double dt, int stage, int num_grids_in_phase,
                               const double* cost, const double* limits, double barrier, double tau, const double* q_prev,
                               const double* sol, const double* sol_next, const double* slack, const double* dual, double* out) {
  if (!g_robot) return 1;
  Robot& robot = *g_robot;
  const int nv = robot.dimv(), nu = robot.dimu(), nq = nv + 1, nc = robot.maxNumContacts(), nx = 2 * nv;
  const int M = nv + 1;
  auto config = std::make_shared<ConfigurationSpaceCost>(robot);
  config->set_q_ref(vec(cost, nq)), config->set_v_ref(vec(cost + M, nv)), config->set_u_ref(vec(cost + 2 * M, nu));
  config->set_q_weight(vec(cost + 3 * M, nv)), config->set_v_weight(vec(cost + 4 * M, nv)), config->set_a_weight(vec(cost + 5 * M, nv));
  config->set_u_weight(vec(cost + 6 * M, nu)), config->set_q_weight_terminal(vec(cost + 7 * M, nv)), config->set_v_weight_terminal(vec(cost + 8 * M, nv));
  config->set_q_weight_impact(vec(cost + 9 * M, nv)), config->set_v_weight_impact(vec(cost + 10 * M, nv)), config->set_dv_weight_impact(vec(cost + 11 * M, nv));
  auto cf = std::make_shared<CostFunction>();
  cf->add("config_cost", config);
  auto constraints = std::make_shared<Constraints>(barrier, tau);
  robot.setJointLimits(vec(limits, nu), vec(limits + nu, nu), vec(limits + 2 * nu, nu), vec(limits + 3 * nu, nu));
  constraints->add("joint_position_lower", std::make_shared<JointPositionLowerLimit>(robot));
  constraints->add("joint_position_upper", std::make_shared<JointPositionUpperLimit>(robot));
  constraints->add("joint_velocity_lower", std::make_shared<JointVelocityLowerLimit>(robot));
  constraints->add("joint_velocity_upper", std::make_shared<JointVelocityUpperLimit>(robot));
  constraints->add("joint_torques_lower", std::make_shared<JointTorquesLowerLimit>(robot));
  constraints->add("joint_torques_upper", std::make_shared<JointTorquesUpperLimit>(robot));
  constraints->add("friction_cone", std::make_shared<FrictionCone>(robot));
  ContactStatus cs = robot.createContactStatus();
  for (int c = 0; c < nc; ++c) {
    if ((active >> c) & 1u) cs.activateContact(c);
    cs.setFrictionCoefficient(c, mu[c]);
    cs.setContactPlacement(c, Eigen::Vector3d(vec(contact_pos + 3 * c, 3)));
  }
  auto seq = std::make_shared<ContactSequence>(robot);
  seq->init(cs);
  IntermediateStage st(cf, constraints, seq);
  GridInfo gi;
  gi.type = GridType::Intermediate;
  gi.dt = dt, gi.stage = stage, gi.phase = 0, gi.num_grids_in_phase = num_grids_in_phase, gi.t = dt * stage;
  auto load = [&](const double* p, SplitSolution& s, bool full) {
    s.setContactStatus(cs);
    s.q = vec(p, nq), p += nq;
    s.v = vec(p, nv), p += nv;
    if (!full) {
      s.lmd = vec(p, nv), s.gmm = vec(p + nv, nv);
      return;
    }
    s.a = vec(p, nv), p += nv;
    s.u = vec(p, nu), p += nu;
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 3; ++k) s.f[c](k) = p[3 * c + k];
    p += 3 * nc;
    s.lmd = vec(p, nv), p += nv;
    s.gmm = vec(p, nv), p += nv;
    s.beta = vec(p, nv), p += nv;
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 3; ++k) s.mu[c](k) = p[3 * c + k];
    p += 3 * nc;
    s.nu_passive = vec(p, 6);
    s.set_f_stack(), s.set_mu_stack();
  };
  SplitSolution s(robot), sn(robot);
  load(sol, s, true), load(sol_next, sn, false);
  OCPData data = st.createData(robot);
  st.initConstraints(robot, gi, s, data);   // sets the stage mask; then the caller's slack / dual
  {
    std::vector<ConstraintComponentData*> comp;
    for (auto& c : data.constraints_data.position_level_data) comp.push_back(&c);
    for (auto& c : data.constraints_data.velocity_level_data) comp.push_back(&c);
    for (auto& c : data.constraints_data.acceleration_level_data) comp.push_back(&c);
    int o = 0;
    for (size_t k = 0; k < comp.size(); ++k) {
      const int n = k < 6 ? nu : 5 * nc;
      for (int r = 0; r < n; ++r) comp[k]->slack(r) = slack[o + r], comp[k]->dual(r) = dual[o + r];
      o += n;
    }
  }
  SplitKKTMatrix km(robot);
  SplitKKTResidual kr(robot);
  st.evalKKT(robot, gi, vec(q_prev, nq), s, sn, data, km, kr);
  if (robot.pending() != 0) return 2;   // injected more than the stage asked for: the call order assumed by the test is off
  double* o = out;
  auto putm = [&](const Eigen::MatrixXd& m) {
    for (int j = 0; j < m.cols(); ++j)
      for (int i = 0; i < m.rows(); ++i) *o++ = m(i, j);
  };
  putm(km.Qxx), putm(km.Qxu), putm(km.Quu), putm(km.Fxx), putm(km.Fvu);
  putm(kr.lx), putm(kr.lu), putm(kr.Fx), putm(km.hx), putm(km.hu), putm(km.fx);
  *o++ = km.Qtt, *o++ = km.Qtt_prev, *o++ = kr.h, *o++ = data.performance_index.kkt_error;
  (void)nx;
  return 0;
}

// The same for SURFACE contacts (six rows each: wrench f, multipliers mu; desired placement = position + rotation) with the
// ContactWrenchCone of a 2X x 2Y sole instead of the friction cone (cone_kind 2) or the FrictionCone on the first three
// wrench components (cone_kind 1); slack / dual: joint-limit rows, then 17 (5) rows per contact by contact index.
// cost: [12][nv + 1] = q_ref (nq), v_ref, u_ref, q / v / a / u weights, terminal q / v weights, impact q / v / dv weights
// sol / sol_next: q (nq), v, a, u (nu), f ([ncontacts][3] by contact index), lmd, gmm, beta, mu ([ncontacts][3]), nu_passive (6)
// slack / dual: rows of the six joint-limit components (nu each), then 5 per contact by contact index
// out: Qxx (nx^2), Qxu (nx nu), Quu (nu^2), Fxx (nx^2), Fvu (nv nu), lx (nx), lu (nu), Fx (nx), hx (nx), hu (nu), fx (nx),
//      [Qtt, Qtt_prev, h, kkt_error] -- all column-major, after IntermediateStage::evalKKT
int ref_contact_stage_eval_kkt_surface(int cone_kind, double X, double Y, const double* contact_rot, unsigned active, const double* contact_pos, const double* mu, double dt, int stage, int num_grids_in_phase,
                               const double* cost, const double* limits, double barrier, double tau, const double* q_prev,
                               const double* sol, const double* sol_next, const double* slack, const double* dual, double* out) {
  if (!g_robot) return 1;
  Robot& robot = *g_robot;
  const int nv = robot.dimv(), nu = robot.dimu(), nq = nv + 1, nc = robot.maxNumContacts(), nx = 2 * nv;
  const int M = nv + 1;
  auto config = std::make_shared<ConfigurationSpaceCost>(robot);
  config->set_q_ref(vec(cost, nq)), config->set_v_ref(vec(cost + M, nv)), config->set_u_ref(vec(cost + 2 * M, nu));
  config->set_q_weight(vec(cost + 3 * M, nv)), config->set_v_weight(vec(cost + 4 * M, nv)), config->set_a_weight(vec(cost + 5 * M, nv));
  config->set_u_weight(vec(cost + 6 * M, nu)), config->set_q_weight_terminal(vec(cost + 7 * M, nv)), config->set_v_weight_terminal(vec(cost + 8 * M, nv));
  config->set_q_weight_impact(vec(cost + 9 * M, nv)), config->set_v_weight_impact(vec(cost + 10 * M, nv)), config->set_dv_weight_impact(vec(cost + 11 * M, nv));
  auto cf = std::make_shared<CostFunction>();
  cf->add("config_cost", config);
  auto constraints = std::make_shared<Constraints>(barrier, tau);
  robot.setJointLimits(vec(limits, nu), vec(limits + nu, nu), vec(limits + 2 * nu, nu), vec(limits + 3 * nu, nu));
  constraints->add("joint_position_lower", std::make_shared<JointPositionLowerLimit>(robot));
  constraints->add("joint_position_upper", std::make_shared<JointPositionUpperLimit>(robot));
  constraints->add("joint_velocity_lower", std::make_shared<JointVelocityLowerLimit>(robot));
  constraints->add("joint_velocity_upper", std::make_shared<JointVelocityUpperLimit>(robot));
  constraints->add("joint_torques_lower", std::make_shared<JointTorquesLowerLimit>(robot));
  constraints->add("joint_torques_upper", std::make_shared<JointTorquesUpperLimit>(robot));
  if (cone_kind == 2) constraints->add("contact_wrench_cone", std::make_shared<ContactWrenchCone>(robot, X, Y));
  else constraints->add("friction_cone", std::make_shared<FrictionCone>(robot));
  ContactStatus cs = robot.createContactStatus();
  for (int c = 0; c < nc; ++c) {
    if ((active >> c) & 1u) cs.activateContact(c);
    cs.setFrictionCoefficient(c, mu[c]);
    {
      Eigen::Matrix3d Rc;
      for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) Rc(r, k) = contact_rot[9 * c + 3 * r + k];
      cs.setContactPlacement(c, Eigen::Vector3d(vec(contact_pos + 3 * c, 3)), Rc);
    }
  }
  auto seq = std::make_shared<ContactSequence>(robot);
  seq->init(cs);
  IntermediateStage st(cf, constraints, seq);
  GridInfo gi;
  gi.type = GridType::Intermediate;
  gi.dt = dt, gi.stage = stage, gi.phase = 0, gi.num_grids_in_phase = num_grids_in_phase, gi.t = dt * stage;
  auto load = [&](const double* p, SplitSolution& s, bool full) {
    s.setContactStatus(cs);
    s.q = vec(p, nq), p += nq;
    s.v = vec(p, nv), p += nv;
    if (!full) {
      s.lmd = vec(p, nv), s.gmm = vec(p + nv, nv);
      return;
    }
    s.a = vec(p, nv), p += nv;
    s.u = vec(p, nu), p += nu;
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 6; ++k) s.f[c](k) = p[6 * c + k];
    p += 6 * nc;
    s.lmd = vec(p, nv), p += nv;
    s.gmm = vec(p, nv), p += nv;
    s.beta = vec(p, nv), p += nv;
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 6; ++k) s.mu[c](k) = p[6 * c + k];
    p += 6 * nc;
    s.nu_passive = vec(p, 6);
    s.set_f_stack(), s.set_mu_stack();
  };
  SplitSolution s(robot), sn(robot);
  load(sol, s, true), load(sol_next, sn, false);
  OCPData data = st.createData(robot);
  st.initConstraints(robot, gi, s, data);   // sets the stage mask; then the caller's slack / dual
  {
    std::vector<ConstraintComponentData*> comp;
    for (auto& c : data.constraints_data.position_level_data) comp.push_back(&c);
    for (auto& c : data.constraints_data.velocity_level_data) comp.push_back(&c);
    for (auto& c : data.constraints_data.acceleration_level_data) comp.push_back(&c);
    int o = 0;
    for (size_t k = 0; k < comp.size(); ++k) {
      const int n = k < 6 ? nu : (cone_kind == 2 ? 17 : 5) * nc;
      for (int r = 0; r < n; ++r) comp[k]->slack(r) = slack[o + r], comp[k]->dual(r) = dual[o + r];
      o += n;
    }
  }
  SplitKKTMatrix km(robot);
  SplitKKTResidual kr(robot);
  st.evalKKT(robot, gi, vec(q_prev, nq), s, sn, data, km, kr);
  if (robot.pending() != 0) return 2;   // injected more than the stage asked for: the call order assumed by the test is off
  double* o = out;
  auto putm = [&](const Eigen::MatrixXd& m) {
    for (int j = 0; j < m.cols(); ++j)
      for (int i = 0; i < m.rows(); ++i) *o++ = m(i, j);
  };
  putm(km.Qxx), putm(km.Qxu), putm(km.Quu), putm(km.Fxx), putm(km.Fvu);
  putm(kr.lx), putm(kr.lu), putm(kr.Fx), putm(km.hx), putm(km.hu), putm(km.fx);
  *o++ = km.Qtt, *o++ = km.Qtt_prev, *o++ = kr.h, *o++ = data.performance_index.kkt_error;
  (void)nx;
  return 0;
}

// IntermediateStage::evalKKT on the grid point two ahead of a touch-down (GridInfo::switching_constraint): cost, state
// equation, contact dynamics, linearizeSwitchingConstraint (src/dynamics/switching_constraint.cpp:26-70) and their
// condensation; no inequality rows (an empty Constraints object).  active: the contacts of the phase, impact: those that
// touch down.  sol as in ref_contact_stage_eval_kkt plus xi (rows of the impacting contacts) at the end.
// out: Qxx, Qxu, Quu, Fxx, Fvu, lx, lu, Fx, Phix (ns x nx), Phiu (ns x nu), Phit (ns), P (ns), hx, hu, [Qtt, h]
int ref_contact_stage_eval_kkt3(unsigned active, unsigned impact, const double* contact_pos, const double* mu, double dt, double dt_next, int stage,
                                int num_grids_in_phase, const double* cost, const double* q_prev, const double* sol, const double* sol_next,
                                double* out) {
  if (!g_robot) return 1;
  Robot& robot = *g_robot;
  const int nv = robot.dimv(), nu = robot.dimu(), nq = nv + 1, nc = robot.maxNumContacts();
  const int M = nv + 1;
  auto config = std::make_shared<ConfigurationSpaceCost>(robot);
  config->set_q_ref(vec(cost, nq)), config->set_v_ref(vec(cost + M, nv)), config->set_u_ref(vec(cost + 2 * M, nu));
  config->set_q_weight(vec(cost + 3 * M, nv)), config->set_v_weight(vec(cost + 4 * M, nv)), config->set_a_weight(vec(cost + 5 * M, nv));
  config->set_u_weight(vec(cost + 6 * M, nu)), config->set_q_weight_terminal(vec(cost + 7 * M, nv)), config->set_v_weight_terminal(vec(cost + 8 * M, nv));
  auto cf = std::make_shared<CostFunction>();
  cf->add("config_cost", config);
  auto constraints = std::make_shared<Constraints>(1.0e-3, 0.995);
  ContactStatus pre = robot.createContactStatus(), post = robot.createContactStatus();
  int ns = 0;
  for (int c = 0; c < nc; ++c) {
    if ((active >> c) & 1u) pre.activateContact(c);
    if (((active | impact) >> c) & 1u) post.activateContact(c);
    if ((impact >> c) & 1u) ns += 3;
    pre.setFrictionCoefficient(c, mu[c]), post.setFrictionCoefficient(c, mu[c]);
    pre.setContactPlacement(c, Eigen::Vector3d(vec(contact_pos + 3 * c, 3)));
    post.setContactPlacement(c, Eigen::Vector3d(vec(contact_pos + 3 * c, 3)));
  }
  auto seq = std::make_shared<ContactSequence>(robot, 2);
  seq->init(pre);
  seq->push_back(post, dt * (stage + 2));
  IntermediateStage st(cf, constraints, seq);
  GridInfo gi;
  gi.type = GridType::Intermediate;
  gi.dt = dt, gi.dt_next = dt_next, gi.stage = stage, gi.phase = 0, gi.impact_index = -1, gi.num_grids_in_phase = num_grids_in_phase, gi.t = dt * stage;
  gi.switching_constraint = true;
  SplitSolution s(robot), sn(robot);
  s.setContactStatus(pre);
  s.setSwitchingConstraintDimension(ns);
  const double* p = sol;
  s.q = vec(p, nq), p += nq;
  s.v = vec(p, nv), p += nv;
  s.a = vec(p, nv), p += nv;
  s.u = vec(p, nu), p += nu;
  for (int c = 0; c < nc; ++c)
    for (int k = 0; k < 3; ++k) s.f[c](k) = p[3 * c + k];
  p += 3 * nc;
  s.lmd = vec(p, nv), p += nv;
  s.gmm = vec(p, nv), p += nv;
  s.beta = vec(p, nv), p += nv;
  for (int c = 0; c < nc; ++c)
    for (int k = 0; k < 3; ++k) s.mu[c](k) = p[3 * c + k];
  p += 3 * nc;
  s.nu_passive = vec(p, 6), p += 6;
  s.xi_stack() = vec(p, ns);
  s.set_f_stack(), s.set_mu_stack();
  sn.q = vec(sol_next, nq), sn.v = vec(sol_next + nq, nv), sn.lmd = vec(sol_next + nq + nv, nv), sn.gmm = vec(sol_next + nq + 2 * nv, nv);
  OCPData data = st.createData(robot);
  st.initConstraints(robot, gi, s, data);
  SplitKKTMatrix km(robot);
  SplitKKTResidual kr(robot);
  st.evalKKT(robot, gi, vec(q_prev, nq), s, sn, data, km, kr);
  if (robot.pending() != 0) return 2;
  double* o = out;
  auto putm = [&](const Eigen::MatrixXd& m) {
    for (int j = 0; j < m.cols(); ++j)
      for (int i = 0; i < m.rows(); ++i) *o++ = m(i, j);
  };
  putm(km.Qxx), putm(km.Qxu), putm(km.Quu), putm(km.Fxx), putm(km.Fvu), putm(kr.lx), putm(kr.lu), putm(kr.Fx);
  putm(km.Phix()), putm(km.Phiu()), putm(km.Phit()), putm(kr.P()), putm(km.hx), putm(km.hu);
  *o++ = km.Qtt, *o++ = kr.h;
  return 0;
}

// The impact grid (kind 1: ImpactStage::evalKKT, src/ocp/impact_stage.cpp:78-120) and the terminal grid (kind 2:
// TerminalStage::evalKKT, src/ocp/terminal_stage.cpp:70-100) of the same OCP.  active_pre: contacts active before the impact,
// impact: the contacts that touch down (their rows carry the impact forces / multipliers); sol for kind 1: q, v, dv, f ([ncontacts][3]
// by contact index), lmd, gmm, beta, mu; for kind 2: q, v, lmd, gmm.  out: kind 1: Qxx, Fxx, lx, Fx; kind 2: Qxx, lx.
int ref_contact_stage_eval_kkt2(int kind, unsigned active_pre, unsigned impact, const double* contact_pos, const double* mu, const double* cost,
                                const double* limits, double barrier, double tau, const double* q_prev, const double* sol,
                                const double* sol_next, double* out) {
  if (!g_robot) return 1;
  Robot& robot = *g_robot;
  const int nv = robot.dimv(), nu = robot.dimu(), nq = nv + 1, nc = robot.maxNumContacts();
  const int M = nv + 1;
  auto config = std::make_shared<ConfigurationSpaceCost>(robot);
  config->set_q_ref(vec(cost, nq)), config->set_v_ref(vec(cost + M, nv)), config->set_u_ref(vec(cost + 2 * M, nu));
  config->set_q_weight(vec(cost + 3 * M, nv)), config->set_v_weight(vec(cost + 4 * M, nv)), config->set_a_weight(vec(cost + 5 * M, nv));
  config->set_u_weight(vec(cost + 6 * M, nu)), config->set_q_weight_terminal(vec(cost + 7 * M, nv)), config->set_v_weight_terminal(vec(cost + 8 * M, nv));
  config->set_q_weight_impact(vec(cost + 9 * M, nv)), config->set_v_weight_impact(vec(cost + 10 * M, nv)), config->set_dv_weight_impact(vec(cost + 11 * M, nv));
  auto cf = std::make_shared<CostFunction>();
  cf->add("config_cost", config);
  auto constraints = std::make_shared<Constraints>(barrier, tau);   // joint limits + FrictionCone: none of them acts on these grids
  robot.setJointLimits(vec(limits, nu), vec(limits + nu, nu), vec(limits + 2 * nu, nu), vec(limits + 3 * nu, nu));
  constraints->add("joint_position_lower", std::make_shared<JointPositionLowerLimit>(robot));
  constraints->add("joint_torques_upper", std::make_shared<JointTorquesUpperLimit>(robot));
  constraints->add("friction_cone", std::make_shared<FrictionCone>(robot));
  ContactStatus pre = robot.createContactStatus(), post = robot.createContactStatus();
  for (int c = 0; c < nc; ++c) {
    if ((active_pre >> c) & 1u) pre.activateContact(c);
    if (((active_pre | impact) >> c) & 1u) post.activateContact(c);
    pre.setFrictionCoefficient(c, mu[c]), post.setFrictionCoefficient(c, mu[c]);
    pre.setContactPlacement(c, Eigen::Vector3d(vec(contact_pos + 3 * c, 3)));
    post.setContactPlacement(c, Eigen::Vector3d(vec(contact_pos + 3 * c, 3)));
  }
  auto seq = std::make_shared<ContactSequence>(robot, 2);
  seq->init(pre);
  seq->push_back(post, 0.1);
  double* o = out;
  auto putm = [&](const Eigen::MatrixXd& m) {
    for (int j = 0; j < m.cols(); ++j)
      for (int i = 0; i < m.rows(); ++i) *o++ = m(i, j);
  };
  SplitKKTMatrix km(robot);
  SplitKKTResidual kr(robot);
  if (kind == 1) {
    const ImpactStatus& is = seq->impactStatus(0);
    ImpactStage st(cf, constraints, seq);
    GridInfo gi;
    gi.type = GridType::Impact, gi.dt = 0.0, gi.impact_index = 0, gi.phase = 0, gi.t = 0.1;
    SplitSolution s(robot), sn(robot);
    s.setContactStatus(is);
    const double* p = sol;
    s.q = vec(p, nq), p += nq;
    s.v = vec(p, nv), p += nv;
    s.dv = vec(p, nv), p += nv;
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 3; ++k) s.f[c](k) = p[3 * c + k];
    p += 3 * nc;
    s.lmd = vec(p, nv), p += nv;
    s.gmm = vec(p, nv), p += nv;
    s.beta = vec(p, nv), p += nv;
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 3; ++k) s.mu[c](k) = p[3 * c + k];
    s.set_f_stack(), s.set_mu_stack();
    sn.q = vec(sol_next, nq), sn.v = vec(sol_next + nq, nv), sn.lmd = vec(sol_next + nq + nv, nv), sn.gmm = vec(sol_next + nq + 2 * nv, nv);
    OCPData data = st.createData(robot);
    st.initConstraints(robot, gi, s, data);
    st.evalKKT(robot, gi, vec(q_prev, nq), s, sn, data, km, kr);
    putm(km.Qxx), putm(km.Fxx), putm(kr.lx), putm(kr.Fx);
  } else {
    TerminalStage st(cf, constraints, seq);
    GridInfo gi;
    gi.type = GridType::Terminal, gi.dt = 0.0, gi.phase = 1, gi.t = 0.2;
    SplitSolution s(robot);
    s.q = vec(sol, nq), s.v = vec(sol + nq, nv), s.lmd = vec(sol + nq + nv, nv), s.gmm = vec(sol + nq + 2 * nv, nv);
    OCPData data = st.createData(robot);
    st.evalKKT(robot, gi, vec(q_prev, nq), s, data, km, kr);
    putm(km.Qxx), putm(kr.lx);
  }
  if (robot.pending() != 0) return 2;
  return 0;
}

}  // extern "C"

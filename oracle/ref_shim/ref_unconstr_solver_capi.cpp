// ref_unconstr_solver_capi.cpp -- UnconstrOCPSolver::updateSolution (src/solver/unconstr_ocp_solver.cpp:96-118) driven with the
// REFERENCE'S OWN stage, cost, constraint, state-equation, dynamics and Riccati sources.  TEST INFRASTRUCTURE ONLY
// (oracle/_ref/librtoc_ref.so, oracle/Makefile.ref).
//
// Reference code that runs here: UnconstrIntermediateStage / UnconstrTerminalStage (src/unconstr), CostFunction +
// ConfigurationSpaceCost (src/cost), Constraints + the six joint-limit components (src/constraints), the forward-Euler state
// equation (src/dynamics/unconstr_state_equation.cpp), UnconstrDynamics (linearise, condense, expand), UnconstrRiccatiRecursion,
// SplitSolution::integrate.  The loop over the horizon restates UnconstrDirectMultipleShooting / UnconstrOCPSolver (their
// constructors want an OCP with a URDF-built Robot): unconstr_direct_multiple_shooting.cpp:85-200, unconstr_ocp_solver.cpp:96-118.
// NOT reference code: Eigen (mini_eigen.hpp) and Pinocchio -- inverse dynamics and its partial derivatives of every grid
// point are INJECTED (computed by this repository's CPU restatement at the same iterate).
#include <memory>
#include <vector>

#include "robotoc/constraints/constraints.hpp"
#include "robotoc/constraints/joint_position_lower_limit.hpp"
#include "robotoc/constraints/joint_position_upper_limit.hpp"
#include "robotoc/constraints/joint_torques_lower_limit.hpp"
#include "robotoc/constraints/joint_torques_upper_limit.hpp"
#include "robotoc/constraints/joint_velocity_lower_limit.hpp"
#include "robotoc/constraints/joint_velocity_upper_limit.hpp"
#include "robotoc/cost/configuration_space_cost.hpp"
#include "robotoc/cost/cost_function.hpp"
#include "robotoc/riccati/unconstr_riccati_recursion.hpp"
#include "robotoc/unconstr/unconstr_intermediate_stage.hpp"
#include "robotoc/unconstr/unconstr_terminal_stage.hpp"

using namespace robotoc;

extern "C" {

// cost: [9][nv] = q_ref, v_ref, u_ref, q_weight, v_weight, a_weight, u_weight, q_weight_terminal, v_weight_terminal
// limits: NULL (no inequality rows) or [4][nv] = q_min, q_max, v_max, u_max
// sol (in: the iterate, out: the next one): [N + 1][7 nv] = q, v, a, u, lmd, gmm, beta
// rnea: [N][nv + 3 nv^2] = ID, dID/dq, dID/dv, dID/da (column-major) at the iterate
// con (in unless init_constraints, out): [N][2][6 nv] = slack, dual of the rows in the order the components are added
// kkt_out: [N + 1][...] condensed Qxx (4nv^2), Qxu (2nv^2), Qaa (nv^2), lx (2nv), la (nv), Fx (2nv) per grid point (terminal: Qxx, lx)
// out: KKT error (sum of squares, before the update), primal step, dual step
int ref_unconstr_update_solution(int nv, int N, double dt, const double* cost, const double* limits, double barrier, double tau,
                                 const double* q0, const double* v0, double* sol, const double* rnea, double* con, int init_constraints,
                                 double* kkt_out, double* out) {
  Robot robot(nv, nv, {});
  auto V = [&](const double* p) {
    Eigen::VectorXd x(nv);
    for (int i = 0; i < nv; ++i) x(i) = p[i];
    return x;
  };
  auto config = std::make_shared<ConfigurationSpaceCost>(robot);
  config->set_q_ref(V(cost)), config->set_v_ref(V(cost + nv)), config->set_u_ref(V(cost + 2 * nv));
  config->set_q_weight(V(cost + 3 * nv)), config->set_v_weight(V(cost + 4 * nv)), config->set_a_weight(V(cost + 5 * nv));
  config->set_u_weight(V(cost + 6 * nv)), config->set_q_weight_terminal(V(cost + 7 * nv)), config->set_v_weight_terminal(V(cost + 8 * nv));
  auto cf = std::make_shared<CostFunction>();
  cf->add("config_cost", config);
  auto constraints = std::make_shared<Constraints>(barrier, tau);
  if (limits) {
    robot.setJointLimits(V(limits), V(limits + nv), V(limits + 2 * nv), V(limits + 3 * nv));
    constraints->add("joint_position_lower", std::make_shared<JointPositionLowerLimit>(robot));
    constraints->add("joint_position_upper", std::make_shared<JointPositionUpperLimit>(robot));
    constraints->add("joint_velocity_lower", std::make_shared<JointVelocityLowerLimit>(robot));
    constraints->add("joint_velocity_upper", std::make_shared<JointVelocityUpperLimit>(robot));
    constraints->add("joint_torques_lower", std::make_shared<JointTorquesLowerLimit>(robot));
    constraints->add("joint_torques_upper", std::make_shared<JointTorquesUpperLimit>(robot));
  }
  UnconstrIntermediateStage stage(robot, cf, constraints);
  UnconstrTerminalStage terminal(robot, cf, constraints);
  std::vector<GridInfo> grid(N + 1);
  for (int i = 0; i <= N; ++i) grid[i].t = dt * i, grid[i].dt = dt, grid[i].stage = i, grid[i].type = i == N ? GridType::Terminal : GridType::Intermediate;
  std::vector<SplitSolution> s(N + 1, SplitSolution(robot));
  std::vector<SplitDirection> d(N + 1, SplitDirection(robot));
  std::vector<SplitKKTMatrix> km(N + 1, SplitKKTMatrix(robot));
  std::vector<SplitKKTResidual> kr(N + 1, SplitKKTResidual(robot));
  std::vector<UnconstrOCPData> data;
  for (int i = 0; i <= N; ++i) {
    const double* r = sol + (size_t)i * 7 * nv;
    s[i].q = V(r), s[i].v = V(r + nv), s[i].a = V(r + 2 * nv), s[i].u = V(r + 3 * nv);
    s[i].lmd = V(r + 4 * nv), s[i].gmm = V(r + 5 * nv), s[i].beta = V(r + 6 * nv);
    data.push_back(i < N ? stage.createData(robot) : terminal.createData(robot));
  }
  auto comps = [&](UnconstrOCPData& dd) {
    std::vector<ConstraintComponentData*> c;
    for (auto& x : dd.constraints_data.position_level_data) c.push_back(&x);
    for (auto& x : dd.constraints_data.velocity_level_data) c.push_back(&x);
    for (auto& x : dd.constraints_data.acceleration_level_data) c.push_back(&x);
    return c;
  };
  // initConstraints (unconstr_direct_multiple_shooting.cpp:44-57) or the caller's slack / dual
  for (int i = 0; i < N; ++i) {
    stage.initConstraints(robot, grid[i], s[i], data[i]);   // also sets the stage mask of the data
    if (limits && !init_constraints) {
      auto c = comps(data[i]);
      for (size_t k = 0; k < c.size(); ++k)
        for (int r = 0; r < nv; ++r) c[k]->slack(r) = con[((size_t)i * 2 + 0) * 6 * nv + k * nv + r], c[k]->dual(r) = con[((size_t)i * 2 + 1) * 6 * nv + k * nv + r];
    }
  }
  // evalKKT (:85-110)
  double kkt_error = 0.0;
  for (int i = 0; i <= N; ++i) {
    if (i < N) {
      const double* r = rnea + (size_t)i * (nv + 3 * nv * nv);
      Eigen::MatrixXd dq(nv, nv), dv(nv, nv), da(nv, nv);
      for (int c = 0; c < nv; ++c)
        for (int rr = 0; rr < nv; ++rr) dq(rr, c) = r[nv + rr + c * nv], dv(rr, c) = r[nv + nv * nv + rr + c * nv], da(rr, c) = r[nv + 2 * nv * nv + rr + c * nv];
      robot.setInverseDynamics(V(r), dq, dv, da);
      stage.evalKKT(robot, grid[i], s[i], s[i + 1], data[i], km[i], kr[i]);
    } else {
      terminal.evalKKT(robot, grid[i], s[i], data[i], km[i], kr[i]);
    }
    kkt_error += data[i].performance_index.kkt_error;
  }
  if (kkt_out) {
    const int nx = 2 * nv;
    const size_t per = (size_t)nx * nx + (size_t)nx * nv + (size_t)nv * nv + nx + nv + nx;
    for (int i = 0; i <= N; ++i) {
      double* o = kkt_out + (size_t)i * per;
      for (int c = 0; c < nx; ++c)
        for (int r = 0; r < nx; ++r) *o++ = km[i].Qxx(r, c);
      for (int c = 0; c < nv; ++c)
        for (int r = 0; r < nx; ++r) *o++ = i < N ? km[i].Qxu(r, c) : 0.0;
      for (int c = 0; c < nv; ++c)
        for (int r = 0; r < nv; ++r) *o++ = i < N ? km[i].Qaa(r, c) : 0.0;
      for (int r = 0; r < nx; ++r) *o++ = kr[i].lx(r);
      for (int r = 0; r < nv; ++r) *o++ = i < N ? kr[i].la(r) : 0.0;
      for (int r = 0; r < nx; ++r) *o++ = i < N ? kr[i].Fx(r) : 0.0;
    }
  }
  // Riccati recursion, initial state direction, step sizes, update (unconstr_ocp_solver.cpp:100-117)
  OCP ocp;
  ocp.robot = robot;
  ocp.N = N;
  ocp.T = dt * N;
  UnconstrRiccatiRecursion riccati(ocp);
  std::vector<SplitRiccatiFactorization> fact(N + 1, SplitRiccatiFactorization(robot));
  riccati.backwardRiccatiRecursion(km, kr, fact);
  d[0].dq() = V(q0) - s[0].q;
  d[0].dv() = V(v0) - s[0].v;
  riccati.forwardRiccatiRecursion(kr, fact, d);
  double primal = 1.0, dual = 1.0;
  for (int i = 0; i < N; ++i) {
    stage.expandPrimalAndDual(grid[i].dt, km[i], kr[i], data[i], d[i]);
    primal = std::min(primal, stage.maxPrimalStepSize(data[i]));
    dual = std::min(dual, stage.maxDualStepSize(data[i]));
  }
  for (int i = 0; i <= N; ++i) {
    if (i < N) {
      stage.updatePrimal(robot, primal, d[i], s[i], data[i]);
      stage.updateDual(dual, data[i]);
    } else {
      terminal.updatePrimal(robot, primal, d[i], s[i], data[i]);
      terminal.updateDual(dual, data[i]);
    }
  }
  for (int i = 0; i <= N; ++i) {
    double* r = sol + (size_t)i * 7 * nv;
    for (int k = 0; k < nv; ++k)
      r[k] = s[i].q(k), r[nv + k] = s[i].v(k), r[2 * nv + k] = s[i].a(k), r[3 * nv + k] = s[i].u(k), r[4 * nv + k] = s[i].lmd(k),
      r[5 * nv + k] = s[i].gmm(k), r[6 * nv + k] = s[i].beta(k);
    if (limits && i < N) {
      auto c = comps(data[i]);
      for (size_t k = 0; k < c.size(); ++k)
        for (int rr = 0; rr < nv; ++rr) con[((size_t)i * 2 + 0) * 6 * nv + k * nv + rr] = c[k]->slack(rr), con[((size_t)i * 2 + 1) * 6 * nv + k * nv + rr] = c[k]->dual(rr);
    }
  }
  out[0] = kkt_error, out[1] = primal, out[2] = dual;
  return 0;
}

}  // extern "C"

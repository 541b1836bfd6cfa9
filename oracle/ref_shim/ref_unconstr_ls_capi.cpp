// ref_unconstr_ls_capi.cpp -- UnconstrOCPSolver::updateSolution WITH SolverOptions::enable_line_search
// (src/solver/unconstr_ocp_solver.cpp:96-118) run by the REFERENCE'S OWN UnconstrDirectMultipleShooting
// (src/unconstr/unconstr_direct_multiple_shooting.cpp), UnconstrRiccatiRecursion and UnconstrLineSearch
// (src/line_search/unconstr_line_search.cpp:37-67).  TEST INFRASTRUCTURE ONLY (oracle/_ref/librtoc_ref.so, oracle/Makefile.ref).
//
// Unlike ref_unconstr_solver_capi.cpp (which restates the ten lines of UnconstrDirectMultipleShooting's loops around the reference's
// stages), the multiple-shooting object itself runs here: the line search copies it (dms_trial_ = dms) and calls its
// integratePrimalSolution / evalOCP.  NOT reference code: Eigen (mini_eigen.hpp) and Pinocchio -- the inverse dynamics (and, for the
// iterate, its partial derivatives) of every grid point are injected in the order the stages ask for them (FIFO of the Robot stand-in).
// The iteration is split in three calls because the trial iterates' inverse dynamics need the Newton direction:
//   ref_uls_direction   evalKKT .. computeStepSizes; hands back the direction and dms.getEval() of the iterate
//   ref_uls_line_search line_search_.computeStepSize over the injected trials (trial k: N inverse-dynamics vectors at step max rate^k)
//   ref_uls_integrate   dms.integrateSolution with the accepted step
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
#define private public   // (the slacks / duals of the rows are set and read in UnconstrDirectMultipleShooting's own data)
#include "robotoc/unconstr/unconstr_direct_multiple_shooting.hpp"
#undef private
#include "robotoc/line_search/unconstr_line_search.hpp"
#include "robotoc/riccati/unconstr_riccati_recursion.hpp"

using namespace robotoc;

namespace {
struct UlsState {
  int nv, N;
  bool rows;
  aligned_vector<Robot> robots;
  OCP ocp;
  std::unique_ptr<UnconstrDirectMultipleShooting> dms;
  std::unique_ptr<UnconstrRiccatiRecursion> riccati;
  std::vector<GridInfo> td;
  Solution s;
  Direction d;
  KKTMatrix km;
  KKTResidual kr;
  UnconstrRiccatiFactorization fact;
  Eigen::VectorXd q0, v0;
  double primal, dual;
};
std::unique_ptr<UlsState> U;
std::vector<ConstraintComponentData*> comps(UnconstrOCPData& dd) {
  std::vector<ConstraintComponentData*> c;
  for (auto& x : dd.constraints_data.position_level_data) c.push_back(&x);
  for (auto& x : dd.constraints_data.velocity_level_data) c.push_back(&x);
  for (auto& x : dd.constraints_data.acceleration_level_data) c.push_back(&x);
  return c;
}
}  // namespace

extern "C" {

// arguments as ref_unconstr_update_solution (ref_unconstr_solver_capi.cpp); out_dir: [N + 1][4 nv] = dq, dv, da, du;
// out_dslack: [N][6 nv] (the rows' slack directions: a trial iterate moves the slacks too); out: kkt error (sum of squares),
// max primal step, max dual step, getEval().cost, .cost_barrier, .primal_feasibility
int ref_uls_direction(int nv, int N, double dt, const double* cost, const double* limits, double barrier, double tau, const double* q0,
                      const double* v0, const double* sol, const double* rnea, const double* con, int init_constraints, double* out_dir,
                      double* out_dslack, double* out) {
  U.reset(new UlsState());
  UlsState& g = *U;
  g.nv = nv, g.N = N, g.rows = limits != nullptr;
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
  g.robots.push_back(robot);
  g.ocp.robot = robot, g.ocp.cost = cf, g.ocp.constraints = constraints, g.ocp.N = N, g.ocp.T = dt * N;
  g.dms.reset(new UnconstrDirectMultipleShooting(g.ocp, 1));
  g.riccati.reset(new UnconstrRiccatiRecursion(g.ocp));
  g.td.assign(N + 1, GridInfo());
  for (int i = 0; i <= N; ++i) g.td[i].t = dt * i, g.td[i].dt = dt, g.td[i].stage = i, g.td[i].type = i == N ? GridType::Terminal : GridType::Intermediate;
  g.s.assign(N + 1, SplitSolution(robot));
  g.d.assign(N + 1, SplitDirection(robot));
  g.km.assign(N + 1, SplitKKTMatrix(robot));
  g.kr.assign(N + 1, SplitKKTResidual(robot));
  g.fact.assign(N + 1, SplitRiccatiFactorization(robot));
  for (int i = 0; i <= N; ++i) {
    const double* r = sol + (size_t)i * 7 * nv;
    g.s[i].q = V(r), g.s[i].v = V(r + nv), g.s[i].a = V(r + 2 * nv), g.s[i].u = V(r + 3 * nv);
    g.s[i].lmd = V(r + 4 * nv), g.s[i].gmm = V(r + 5 * nv), g.s[i].beta = V(r + 6 * nv);
  }
  g.q0 = V(q0), g.v0 = V(v0);
  g.dms->initConstraints(g.robots, g.td, g.s);   // unconstr_ocp_solver.cpp:92-94
  if (limits && !init_constraints) {
    for (int i = 0; i < N; ++i) {
      auto c = comps(g.dms->data_[i]);
      for (size_t k = 0; k < c.size(); ++k)
        for (int r = 0; r < nv; ++r) c[k]->slack(r) = con[((size_t)i * 2 + 0) * 6 * nv + k * nv + r], c[k]->dual(r) = con[((size_t)i * 2 + 1) * 6 * nv + k * nv + r];
    }
  }
  Robot& rb = g.robots[0];
  for (int i = 0; i < N; ++i) {   // what evalKKT's stages pop, in grid order
    const double* r = rnea + (size_t)i * (nv + 3 * nv * nv);
    Eigen::MatrixXd dq(nv, nv), dv(nv, nv), da(nv, nv);
    for (int c = 0; c < nv; ++c)
      for (int rr = 0; rr < nv; ++rr) dq(rr, c) = r[nv + rr + c * nv], dv(rr, c) = r[nv + nv * nv + rr + c * nv], da(rr, c) = r[nv + 2 * nv * nv + rr + c * nv];
    rb.inject("ID", V(r));
    rb.inject("dIDdq", dq), rb.inject("dIDdv", dv), rb.inject("dIDda", da);
  }
  g.dms->evalKKT(g.robots, g.td, g.q0, g.v0, g.s, g.km, g.kr);                        // unconstr_ocp_solver.cpp:101
  g.riccati->backwardRiccatiRecursion(g.km, g.kr, g.fact);                            // :102-103
  g.dms->computeInitialStateDirection(g.q0, g.v0, g.s, g.d);                          // :104
  g.riccati->forwardRiccatiRecursion(g.kr, g.fact, g.d);                              // :105
  g.dms->computeStepSizes(g.td, g.km, g.kr, g.d);                                     // :106
  g.primal = g.dms->maxPrimalStepSize(), g.dual = g.dms->maxDualStepSize();
  for (int i = 0; i <= N; ++i) {
    double* o = out_dir + (size_t)i * 4 * nv;
    for (int k = 0; k < nv; ++k) o[k] = g.d[i].dq()(k), o[nv + k] = g.d[i].dv()(k), o[2 * nv + k] = i < N ? g.d[i].da()(k) : 0.0, o[3 * nv + k] = i < N ? g.d[i].du(k) : 0.0;
  }
  if (limits && out_dslack) {
    for (int i = 0; i < N; ++i) {
      auto c = comps(g.dms->data_[i]);
      for (size_t k = 0; k < c.size(); ++k)
        for (int r = 0; r < nv; ++r) out_dslack[(size_t)i * 6 * nv + k * nv + r] = c[k]->dslack(r);
    }
  }
  out[0] = g.dms->getEval().kkt_error, out[1] = g.primal, out[2] = g.dual;
  out[3] = g.dms->getEval().cost, out[4] = g.dms->getEval().cost_barrier, out[5] = g.dms->getEval().primal_feasibility;
  rb.clearInjections();
  return 0;
}

// trial_id: [ntrials][N][nv] inverse dynamics of the trial iterates at the steps max, max rate, max rate^2, ... (what evalOCP's stages
// pop); out: accepted step, trials left unconsumed
int ref_uls_line_search(const double* trial_id, int ntrials, double rate, double min_step, double cost_rate, double viol_rate, double* out) {
  if (!U || !U->dms) return 1;
  UlsState& g = *U;
  const int nv = g.nv, N = g.N;
  Robot& rb = g.robots[0];
  for (int k = 0; k < ntrials; ++k)
    for (int i = 0; i < N; ++i) {
      Eigen::VectorXd id(nv);
      for (int r = 0; r < nv; ++r) id(r) = trial_id[((size_t)k * N + i) * nv + r];
      rb.inject("ID", id);
    }
  LineSearchSettings st;
  st.line_search_method = LineSearchMethod::Filter;
  st.step_size_reduction_rate = rate, st.min_step_size = min_step;
  st.filter_cost_reduction_rate = cost_rate, st.filter_constraint_violation_reduction_rate = viol_rate;
  UnconstrLineSearch ls(g.ocp, st);
  ls.clearHistory();
  g.primal = ls.computeStepSize(*g.dms, g.robots, g.td, g.q0, g.v0, g.s, g.d, g.primal);   // unconstr_ocp_solver.cpp:109-111
  out[0] = g.primal;
  out[1] = (double)rb.pendingOf("ID") / (N > 0 ? N : 1);
  rb.clearInjections();
  return 0;
}

// sol_out: [N + 1][7 nv]; con_out: [N][2][6 nv] slack, dual
int ref_uls_integrate(double* sol_out, double* con_out) {
  if (!U || !U->dms) return 1;
  UlsState& g = *U;
  const int nv = g.nv, N = g.N;
  g.dms->integrateSolution(g.robots, g.td, g.primal, g.dual, g.d, g.s);   // unconstr_ocp_solver.cpp:115-116
  for (int i = 0; i <= N; ++i) {
    double* r = sol_out + (size_t)i * 7 * nv;
    for (int k = 0; k < nv; ++k)
      r[k] = g.s[i].q(k), r[nv + k] = g.s[i].v(k), r[2 * nv + k] = g.s[i].a(k), r[3 * nv + k] = g.s[i].u(k), r[4 * nv + k] = g.s[i].lmd(k),
      r[5 * nv + k] = g.s[i].gmm(k), r[6 * nv + k] = g.s[i].beta(k);
    if (g.rows && con_out && i < N) {
      auto c = comps(g.dms->data_[i]);
      for (size_t k = 0; k < c.size(); ++k)
        for (int rr = 0; rr < nv; ++rr)
          con_out[((size_t)i * 2 + 0) * 6 * nv + k * nv + rr] = c[k]->slack(rr), con_out[((size_t)i * 2 + 1) * 6 * nv + k * nv + rr] = c[k]->dual(rr);
    }
  }
  return 0;
}

}  // extern "C"

// ref_ocp_capi.cpp -- OCPSolver::updateSolution (src/solver/ocp_solver.cpp:111-145) of a floating-base robot over a contact
// sequence with lifts and touch-downs, run by the REFERENCE'S OWN sources.  TEST INFRASTRUCTURE ONLY (oracle/_ref).
//
// Reference code that runs here: DirectMultipleShooting (src/ocp/direct_multiple_shooting.cpp: evalKKT, computeInitialStateDirection,
// computeStepSizes, integrateSolution), IntermediateStage / ImpactStage / TerminalStage, ContactSequence, CostFunction +
// ConfigurationSpaceCost, Constraints (six joint-limit components + FrictionCone), the state-equation, contact / impact dynamics
// and switching-constraint sources, RiccatiRecursion with its factorizers, SplitSolution::integrate.  The ten lines of
// updateSolution that string them together are restated below (OCPSolver's constructor wants an OCP built from a URDF).
// NOT reference code: Eigen (mini_eigen.hpp); every Pinocchio quantity, injected by the caller in the order the stages ask for
// them (FIFO, oracle/ref_shim/robotoc/robot/robot.hpp) from this repository's CPU rigid-body restatement.  Because the update of a
// free-flyer configuration needs the Newton direction, the iteration is split: ref_ocp_direction() runs everything up to the
// step sizes and hands back the directions; the caller computes q (+) step dq and ref_ocp_integrate() finishes.
#include <memory>
#include <vector>

#include "../../include/rtoc.h"
#include "robotoc/constraints/constraints.hpp"
#include "robotoc/constraints/friction_cone.hpp"
#include "robotoc/constraints/joint_position_lower_limit.hpp"
#include "robotoc/constraints/joint_position_upper_limit.hpp"
#include "robotoc/constraints/joint_torques_lower_limit.hpp"
#include "robotoc/constraints/joint_torques_upper_limit.hpp"
#include "robotoc/constraints/joint_velocity_lower_limit.hpp"
#include "robotoc/constraints/joint_velocity_upper_limit.hpp"
#include "robotoc/cost/configuration_space_cost.hpp"
#include "robotoc/cost/cost_function.hpp"
#define private public
#include "robotoc/ocp/direct_multiple_shooting.hpp"
#undef private
#include "robotoc/line_search/line_search.hpp"
#include "robotoc/planner/contact_sequence.hpp"
#include "robotoc/riccati/riccati_recursion.hpp"
#include "robotoc/sto/sto_constraints.hpp"
#include "robotoc/sto/sto_cost_function.hpp"
#define private public
#include "robotoc/sto/switching_time_optimization.hpp"
#undef private

using namespace robotoc;

namespace {
struct State {
  int nv, nu, nc, n;
  int cd = 3;                      // rows per contact: 3 (point contacts) or 6 (surface contacts: ref_ocp_begin_surface)
  std::vector<double> rotations;   // [n][nc][9] row-major contact-frame rotations of the schedule (surface contacts)
  aligned_vector<Robot> robots;
  OCP ocp;
  std::unique_ptr<DirectMultipleShooting> dms;
  std::unique_ptr<RiccatiRecursion> riccati;
  TimeDiscretization td;
  Solution s;
  Direction d;
  KKTMatrix km;
  KKTResidual kr;
  RiccatiFactorization fact;
  double primal, dual;
  // switching-time optimisation (ref_ocp_sto_setup): the reference's SwitchingTimeOptimization over its STOConstraints
  bool sto_on = false;
  std::vector<int> event_sto;
  std::vector<double> min_dwell, sto_slack, sto_dual;
  double sto_barrier = 1e-3, sto_tau = 0.995, sto_reg = 0.0;
  std::unique_ptr<SwitchingTimeOptimization> sto;
  std::shared_ptr<ContactSequence> seq;
};
std::unique_ptr<State> G;
Eigen::VectorXd vec(const double* p, int n) {
  Eigen::VectorXd v(n);
  for (int i = 0; i < n; ++i) v(i) = p[i];
  return v;
}
int sol_len(int nv, int nu, int nc, int cd = 3) { return (nv + 1) + 2 * nv + nu + cd * nc + 3 * nv + cd * nc + 6 + cd * nc; }
}  // namespace

extern "C" {

int ref_ocp_sol_len(int nv, int nu, int nc) { return sol_len(nv, nu, nc); }
int ref_ocp_sol_len_cd(int nv, int nu, int nc, int cd) { return sol_len(nv, nu, nc, cd); }

// Injections are pushed with ref_ocp_inject between ref_ocp_begin and ref_ocp_direction.
int ref_ocp_begin(int nv, int nu, int ncontacts) {
  G.reset(new State());
  G->nv = nv, G->nu = nu, G->nc = ncontacts;
  G->robots.push_back(Robot(nv, nu, std::vector<ContactType>(ncontacts, ContactType::PointContact)));
  return 0;
}
// the same for a robot on SURFACE contacts (iCub's soles): six rows per contact in f, mu, xi and in the switching constraints;
// rotations: [n][ncontacts][9] row-major frame rotations of the contact schedule (ContactStatus::setContactPlacement)
int ref_ocp_begin_surface(int nv, int nu, int ncontacts, const double* rotations, int n) {
  G.reset(new State());
  G->nv = nv, G->nu = nu, G->nc = ncontacts, G->cd = 6;
  G->rotations.assign(rotations, rotations + (size_t)n * ncontacts * 9);
  G->robots.push_back(Robot(nv, nu, std::vector<ContactType>(ncontacts, ContactType::SurfaceContact)));
  return 0;
}
int ref_ocp_inject(const char* key, const double* data, int rows, int cols) {
  if (!G) return 1;
  Eigen::MatrixXd m(rows, cols);
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) m(i, j) = data[i + (size_t)j * rows];
  G->robots[0].inject(key, m);
  return 0;
}

// The STO half of the iteration (ocp_solver.cpp:119, 128-132, 143), between ref_ocp_begin and ref_ocp_direction: event_sto[nev] =
// ContactSequence's sto flag of every discrete event in time order, min_dwell[nev + 1] / barrier / tau = the STOConstraints,
// sto_reg = sto_.setRegularization, slack / dual [nev + 1] = the dwell-time rows' ConstraintComponentData (NULL: initConstraints).
// The grid table handed to ref_ocp_direction then carries the sto / sto_next flags and the PhaseBased time steps.
int ref_ocp_sto_setup(const int* event_sto, int nev, const double* min_dwell, double barrier, double tau, double sto_reg,
                      const double* slack, const double* dual) {
  if (!G) return 1;
  G->sto_on = true;
  G->event_sto.assign(event_sto, event_sto + nev);
  G->min_dwell.assign(min_dwell, min_dwell + nev + 1);
  G->sto_barrier = barrier, G->sto_tau = tau, G->sto_reg = sto_reg;
  if (slack && dual) G->sto_slack.assign(slack, slack + nev + 1), G->sto_dual.assign(dual, dual + nev + 1);
  return 0;
}

// grid / masks / positions: the discretisation and contact schedule (rtoc_set_grid / rtoc_set_contact_schedule); sol: [n][sol_len]
// = q (nq), v, a (dv on impact grids), u, f ([nc][3] by contact index), lmd, gmm, beta, mu ([nc][3]), nu_passive (6), xi ([nc][3]
// slots, the first GridInfo::dims used); slack / dual: [n][6 nu + 5 nc] (cone rows by contact index);
// out_dq: [n][nv] the configuration directions; out_steps: primal, dual, KKT error (sum of squares)
int ref_ocp_direction(const rtoc_grid* grid, const unsigned* masks, const double* positions, int n, const double* mu, const double* cost,
                      const double* limits, double barrier, double tau, const double* q0, const double* v0, const double* sol,
                      const double* slack, const double* dual, double* out_dq, double* out_steps) {
  if (!G) return 1;
  State& g = *G;
  Robot& robot = g.robots[0];
  const int nv = g.nv, nu = g.nu, nc = g.nc, nq = nv + 1, M = nv + 1;
  g.n = n;
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
  // ---- contact sequence and grid infos from the grid table: phases advance at lift and impact grids ----
  auto status_of = [&](unsigned mask, int i) {
    ContactStatus cs = robot.createContactStatus();
    for (int c = 0; c < nc; ++c) {
      if ((mask >> c) & 1u) cs.activateContact(c);
      cs.setFrictionCoefficient(c, mu[c]);
      if (g.cd == 6) {
        Eigen::Matrix3d R;
        const double* r = g.rotations.data() + ((size_t)i * nc + c) * 9;
        for (int a_ = 0; a_ < 3; ++a_)
          for (int b_ = 0; b_ < 3; ++b_) R(a_, b_) = r[3 * a_ + b_];
        cs.setContactPlacement(c, Eigen::Vector3d(vec(positions + ((size_t)i * nc + c) * 3, 3)), R);
      } else {
        cs.setContactPlacement(c, Eigen::Vector3d(vec(positions + ((size_t)i * nc + c) * 3, 3)));
      }
    }
    return cs;
  };
  int nevents = 0;
  for (int i = 0; i < n; ++i) nevents += grid[i].type == RTOC_GRID_IMPACT || grid[i].type == RTOC_GRID_LIFT;
  auto seq = std::make_shared<ContactSequence>(robot, nevents + 1);
  std::vector<GridInfo> gi(n);
  int phase = 0, impact_index = -1, lift_index = -1, event = 0;
  double t = 0.0;
  auto event_flag = [&]() { return g.sto_on && event < (int)g.event_sto.size() && g.event_sto[event++] != 0; };
  seq->init(status_of(masks[0], 0));
  for (int i = 0; i < n; ++i) {
    GridInfo& o = gi[i];
    o.type = grid[i].type == RTOC_GRID_IMPACT ? GridType::Impact : grid[i].type == RTOC_GRID_LIFT ? GridType::Lift
             : grid[i].type == RTOC_GRID_TERMINAL ? GridType::Terminal : GridType::Intermediate;
    if (grid[i].type == RTOC_GRID_IMPACT) {
      // the phase after the touch-down: the contacts of the next grid point
      seq->push_back(status_of(masks[i + 1], i + 1), t, event_flag());
      ++impact_index;
      o.phase = phase, o.impact_index = impact_index, o.lift_index = lift_index;
      ++phase;
    } else {
      if (grid[i].type == RTOC_GRID_LIFT) {
        seq->push_back(status_of(masks[i], i), t, event_flag());
        ++lift_index;
        ++phase;
      }
      o.phase = phase, o.impact_index = impact_index, o.lift_index = lift_index;
    }
    o.t = t, o.dt = grid[i].dt, o.dt_next = i + 1 < n ? grid[i + 1].dt : 0.0;
    o.stage = grid[i].time_stage < 0 ? 0 : grid[i].time_stage;
    o.num_grids_in_phase = grid[i].num_grids_in_phase;
    o.switching_constraint = grid[i].switching_constraint != 0;
    o.sto = g.sto_on && grid[i].sto != 0, o.sto_next = g.sto_on && grid[i].sto_next != 0;
    t += grid[i].dt;
  }
  g.td = TimeDiscretization(gi);
  g.ocp.robot = robot, g.ocp.N = n - 1, g.ocp.T = t, g.ocp.reserved_num_discrete_events = nevents + 1;
  g.ocp.cost = cf, g.ocp.constraints = constraints, g.ocp.contact_sequence = seq;
  g.seq = seq;
  if (g.sto_on) {
    g.ocp.sto_cost = std::make_shared<STOCostFunction>();   // empty, like the reference's examples (examples/anymal/python/jump_sto.py:104)
    g.ocp.sto_constraints = std::make_shared<STOConstraints>(g.min_dwell, g.sto_barrier, g.sto_tau);
    g.sto.reset(new SwitchingTimeOptimization(g.ocp));
    g.sto->setRegularization(g.sto_reg);
    g.sto->initConstraints(g.td);
    if (!g.sto_slack.empty())
      for (size_t p = 0; p < g.sto_slack.size(); ++p)
        g.sto->constraint_data_.slack((int)p) = g.sto_slack[p], g.sto->constraint_data_.dual((int)p) = g.sto_dual[p];
  }
  g.dms.reset(new DirectMultipleShooting(g.ocp, 1));
  g.dms->resizeData(g.td);
  g.riccati.reset(new RiccatiRecursion(g.ocp, 0.1));
  g.s = Solution(n, SplitSolution(robot));
  g.d = Direction(n, SplitDirection(robot));
  g.km = KKTMatrix(n, SplitKKTMatrix(robot));
  g.kr = KKTResidual(n, SplitKKTResidual(robot));
  g.fact = RiccatiFactorization(n, SplitRiccatiFactorization(robot));
  const int SL = sol_len(nv, nu, nc, g.cd), cd = g.cd;
  for (int i = 0; i < n; ++i) {
    SplitSolution& s = g.s[i];
    const bool impact = grid[i].type == RTOC_GRID_IMPACT, terminal = grid[i].type == RTOC_GRID_TERMINAL;
    // OCPSolver::resizeData (src/solver/ocp_solver.cpp:461-478): the terminal grid carries no contact status
    if (impact) s.setContactStatus(seq->impactStatus(gi[i].impact_index));
    else if (!terminal) s.setContactStatus(seq->contactStatus(gi[i].phase));
    const int ns = (!impact && !terminal && grid[i].switching_constraint) ? grid[i].dims : 0;
    s.setSwitchingConstraintDimension(ns);
    const double* p = sol + (size_t)i * SL;
    s.q = vec(p, nq), p += nq;
    s.v = vec(p, nv), p += nv;
    (impact ? s.dv : s.a) = vec(p, nv), p += nv;
    s.u = vec(p, nu), p += nu;
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < cd; ++k) s.f[c](k) = terminal ? 0.0 : p[cd * c + k];
    p += cd * nc;
    s.lmd = vec(p, nv), p += nv;
    s.gmm = vec(p, nv), p += nv;
    s.beta = vec(p, nv), p += nv;
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < cd; ++k) s.mu[c](k) = terminal ? 0.0 : p[cd * c + k];
    p += cd * nc;
    s.nu_passive = vec(p, 6), p += 6;
    for (int k = 0; k < ns; ++k) s.xi_stack()(k) = p[k];
    s.set_f_stack(), s.set_mu_stack();
  }
  // initConstraints sets the stage masks of the data (and pops nothing: the slacks are overwritten next); then the caller's slack / dual
  {
    const int nrow = 6 * nu + 5 * nc;
    for (int i = 0; i < n; ++i) {
      OCPData& data = g.dms->ocp_data_[i];
      const int st = grid[i].type == RTOC_GRID_IMPACT ? -1 : (grid[i].type == RTOC_GRID_TERMINAL ? -1 : gi[i].stage);
      data.constraints_data = constraints->createConstraintsData(robot, st);
      std::vector<ConstraintComponentData*> comp;
      for (auto& c : data.constraints_data.position_level_data) comp.push_back(&c);
      for (auto& c : data.constraints_data.velocity_level_data) comp.push_back(&c);
      for (auto& c : data.constraints_data.acceleration_level_data) comp.push_back(&c);
      int o = 0;
      for (size_t k = 0; k < comp.size(); ++k) {
        const int m = k < 6 ? nu : 5 * nc;
        for (int r = 0; r < m; ++r) comp[k]->slack(r) = slack[(size_t)i * nrow + o + r], comp[k]->dual(r) = dual[(size_t)i * nrow + o + r];
        o += m;
      }
    }
  }
  // ---- OCPSolver::updateSolution (ocp_solver.cpp:118-132) ----
  const Eigen::VectorXd q = vec(q0, nq), v = vec(v0, nv);
  g.dms->evalKKT(g.robots, g.td, q, v, g.s, g.km, g.kr);
  if (g.sto_on) g.sto->evalKKT(g.td, g.km, g.kr);                                               // :119
  g.riccati->backwardRiccatiRecursion(g.td, g.km, g.kr, g.fact);
  g.dms->computeInitialStateDirection(robot, q, v, g.s, g.d);
  g.riccati->forwardRiccatiRecursion(g.td, g.km, g.kr, g.fact, g.d);
  g.dms->computeStepSizes(g.td, g.d);
  g.primal = g.dms->maxPrimalStepSize(), g.dual = g.dms->maxDualStepSize();
  if (g.sto_on) {                                                                                // :128-132
    g.sto->computeStepSizes(g.td, g.d);
    g.primal = std::min(g.primal, g.sto->maxPrimalStepSize()), g.dual = std::min(g.dual, g.sto->maxDualStepSize());
  }
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < nv; ++k) out_dq[(size_t)i * nv + k] = g.d[i].dq()(k);
  out_steps[0] = g.primal, out_steps[1] = g.dual, out_steps[2] = g.dms->getEval().kkt_error;
  if (g.sto_on) out_steps[2] += g.sto->getEval().kkt_error;   // OCPSolver::KKTError()^2 (:429-431)
  // dms_.getEval(): what the line search reads of the current iterate (line_search.cpp:58-61)
  out_steps[3] = g.dms->getEval().cost, out_steps[4] = g.dms->getEval().cost_barrier, out_steps[5] = g.dms->getEval().primal_feasibility;
  return (int)robot.pending() == 0 ? 0 : 2;
}

static void pack_solution(const State& g, const Solution& sol, double* sol_out) {
  const int nv = g.nv, nu = g.nu, nc = g.nc, nq = nv + 1, n = g.n, cd = g.cd, SL = sol_len(nv, nu, nc, cd);
  for (int i = 0; i < n; ++i) {
    const SplitSolution& s = sol[i];
    const bool impact = g.td[i].type == GridType::Impact;
    double* p = sol_out + (size_t)i * SL;
    for (int k = 0; k < nq; ++k) *p++ = s.q(k);
    for (int k = 0; k < nv; ++k) *p++ = s.v(k);
    for (int k = 0; k < nv; ++k) *p++ = impact ? s.dv(k) : s.a(k);
    for (int k = 0; k < nu; ++k) *p++ = s.u(k);
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < cd; ++k) *p++ = s.f[c](k);
    for (int k = 0; k < nv; ++k) *p++ = s.lmd(k);
    for (int k = 0; k < nv; ++k) *p++ = s.gmm(k);
    for (int k = 0; k < nv; ++k) *p++ = s.beta(k);
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < cd; ++k) *p++ = s.mu[c](k);
    for (int k = 0; k < 6; ++k) *p++ = s.nu_passive(k);
    for (int k = 0; k < cd * nc; ++k) *p++ = k < s.dims() ? s.xi_stack()(k) : 0.0;
  }
}

// The trial iterate of the line search for step `alpha` (dms_trial_.integratePrimalSolution, line_search.cpp:65-69), so that the
// caller can compute the rigid-body quantities evalOCP will ask for at it.  q_integrated: [n][nq] = s[i].q (+) alpha d[i].dq.
int ref_ocp_trial_solution(double alpha, const double* q_integrated, double* sol_out) {
  if (!G || !G->dms) return 1;
  State& g = *G;
  Robot& robot = g.robots[0];
  const int nq = g.nv + 1;
  for (int i = 0; i < g.n; ++i) robot.inject("integrateConfiguration", vec(q_integrated + (size_t)i * nq, nq));
  DirectMultipleShooting trial = *g.dms;
  Solution s = g.s;
  trial.integratePrimalSolution(g.robots, g.td, alpha, g.d, s);
  pack_solution(g, s, sol_out);
  return (int)robot.pending() == 0 ? 0 : 2;
}

// dms_trial_.integratePrimalSolution(alpha) + evalOCP (line_search.cpp:65-71) for ONE trial step, the injections pushed beforehand
// like for ref_ocp_line_search: out = cost, cost_barrier, primal_feasibility of the trial iterate.
int ref_ocp_trial_eval(double alpha, double* out) {
  if (!G || !G->dms) return 1;
  State& g = *G;
  DirectMultipleShooting trial = *g.dms;
  Solution s = g.s;
  KKTResidual kr = g.kr;
  trial.integratePrimalSolution(g.robots, g.td, alpha, g.d, s);
  trial.evalOCP(g.robots, g.td, g.s[0].q, g.s[0].v, s, kr);
  out[0] = trial.getEval().cost, out[1] = trial.getEval().cost_barrier, out[2] = trial.getEval().primal_feasibility;
  return (int)g.robots[0].pending() == 0 ? 0 : 2;
}

// LineSearch::computeStepSize (src/line_search/line_search.cpp:31-83, filter method) by the reference's own LineSearch over its
// DirectMultipleShooting: the injections of every trial it may evaluate are pushed beforehand (trial k: the n integrated
// configurations, then what evalOCP asks for at the trial iterate); what it does not consume is dropped.
int ref_ocp_line_search(double rate, double min_step, double cost_rate, double viol_rate, double* out_step) {
  if (!G || !G->dms) return 1;
  State& g = *G;
  LineSearchSettings st;
  st.line_search_method = LineSearchMethod::Filter;
  st.step_size_reduction_rate = rate, st.min_step_size = min_step;
  st.filter_cost_reduction_rate = cost_rate, st.filter_constraint_violation_reduction_rate = viol_rate;
  LineSearch ls(g.ocp, st);
  ls.clearHistory();
  const Eigen::VectorXd q = g.s[0].q, v = g.s[0].v;   // (evalOCP does not read the initial state)
  g.primal = ls.computeStepSize(*g.dms, g.robots, g.td, q, v, g.s, g.d, g.primal);   // ocp_solver.cpp:133-139
  out_step[0] = g.primal;
  out_step[1] = (double)g.robots[0].pendingOf("ID");   // what the trials it did not evaluate left behind
  g.robots[0].clearInjections();
  return 0;
}

// LineSearch::computeStepSize with LineSearchMethod::MeritBacktracking (src/line_search/line_search.cpp:87-128): penalty parameter
// from the multipliers, directional derivative of the merit function from a trial at step `eps`, Armijo backtracking.  Injections as
// for ref_ocp_line_search, the eps-trial's first.
int ref_ocp_line_search_merit(double rate, double min_step, double armijo_control_rate, double margin_rate, double eps, double* out_step) {
  if (!G || !G->dms) return 1;
  State& g = *G;
  LineSearchSettings st;
  st.line_search_method = LineSearchMethod::MeritBacktracking;
  st.step_size_reduction_rate = rate, st.min_step_size = min_step;
  st.armijo_control_rate = armijo_control_rate, st.margin_rate = margin_rate, st.eps = eps;
  LineSearch ls(g.ocp, st);
  const Eigen::VectorXd q = g.s[0].q, v = g.s[0].v;
  g.primal = ls.computeStepSize(*g.dms, g.robots, g.td, q, v, g.s, g.d, g.primal);
  out_step[0] = g.primal;
  out_step[1] = (double)g.robots[0].pendingOf("ID");
  g.robots[0].clearInjections();
  return 0;
}

// q_integrated: [n][nq] = s[i].q (+) primal_step d[i].dq (pushed as the integrateConfiguration injections, in grid order)
int ref_ocp_integrate(const double* q_integrated, double* sol_out, double* slack_out, double* dual_out) {
  if (!G || !G->dms) return 1;
  State& g = *G;
  Robot& robot = g.robots[0];
  const int nv = g.nv, nu = g.nu, nc = g.nc, nq = nv + 1, n = g.n;
  for (int i = 0; i < n; ++i) robot.inject("integrateConfiguration", vec(q_integrated + (size_t)i * nq, nq));
  g.dms->integrateSolution(g.robots, g.td, g.primal, g.dual, g.d, g.s);   // ocp_solver.cpp:142
  if (g.sto_on) g.sto->integrateSolution(g.td, g.primal, g.dual, g.d);     // :143
  const int cd = g.cd, SL = sol_len(nv, nu, nc, cd), nrow = 6 * nu + 5 * nc;
  for (int i = 0; i < n; ++i) {
    const SplitSolution& s = g.s[i];
    const bool impact = g.td[i].type == GridType::Impact;
    double* p = sol_out + (size_t)i * SL;
    for (int k = 0; k < nq; ++k) *p++ = s.q(k);
    for (int k = 0; k < nv; ++k) *p++ = s.v(k);
    for (int k = 0; k < nv; ++k) *p++ = impact ? s.dv(k) : s.a(k);
    for (int k = 0; k < nu; ++k) *p++ = s.u(k);
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < cd; ++k) *p++ = s.f[c](k);
    for (int k = 0; k < nv; ++k) *p++ = s.lmd(k);
    for (int k = 0; k < nv; ++k) *p++ = s.gmm(k);
    for (int k = 0; k < nv; ++k) *p++ = s.beta(k);
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < cd; ++k) *p++ = s.mu[c](k);
    for (int k = 0; k < 6; ++k) *p++ = s.nu_passive(k);
    for (int k = 0; k < cd * nc; ++k) *p++ = k < s.dims() ? s.xi_stack()(k) : 0.0;
    OCPData& data = g.dms->ocp_data_[i];
    std::vector<ConstraintComponentData*> comp;
    for (auto& c : data.constraints_data.position_level_data) comp.push_back(&c);
    for (auto& c : data.constraints_data.velocity_level_data) comp.push_back(&c);
    for (auto& c : data.constraints_data.acceleration_level_data) comp.push_back(&c);
    int o = 0;
    for (size_t k = 0; k < comp.size(); ++k) {
      const int m = k < 6 ? nu : 5 * nc;
      for (int r = 0; r < m; ++r) slack_out[(size_t)i * nrow + o + r] = comp[k]->slack(r), dual_out[(size_t)i * nrow + o + r] = comp[k]->dual(r);
      o += m;
    }
  }
  return (int)robot.pending() == 0 ? 0 : 2;
}

// After ref_ocp_direction (stage 0) / ref_ocp_integrate (stage 1) of an STO iteration: event_times[nev] = ContactSequence's event
// times, con[6][nev + 1] = slack, dual, residual, cmpl, dslack, ddual of the dwell-time rows, lt_qtt[2][nev] = the gradient and
// Hessian diagonal SwitchingTimeOptimization::evalKKT scattered, perf[2] = its kkt_error and the rows' own KKTError(),
// dts[n][2] = SplitDirection::dts, dts_next of every grid point.
int ref_ocp_sto_result(double* event_times, double* con, double* lt_qtt, double* perf, double* dts) {
  if (!G || !G->sto_on || !G->sto) return 1;
  State& g = *G;
  const int nev = (int)g.event_sto.size(), np = nev + 1;
  const auto& et = g.seq->eventTimes();
  for (int e = 0; e < nev; ++e) event_times[e] = et[e];
  const ConstraintComponentData& c = g.sto->constraint_data_;
  for (int p = 0; p < np; ++p) {
    con[0 * np + p] = c.slack(p), con[1 * np + p] = c.dual(p), con[2 * np + p] = c.residual(p), con[3 * np + p] = c.cmpl(p);
    con[4 * np + p] = c.dslack(p), con[5 * np + p] = c.ddual(p);
  }
  for (int e = 0; e < nev; ++e) lt_qtt[e] = g.sto->lt_.coeff(e), lt_qtt[nev + e] = g.sto->Qtt_.coeff(e, e);
  perf[0] = g.sto->getEval().kkt_error, perf[1] = c.KKTError();
  for (int i = 0; i < g.n; ++i) dts[2 * i] = g.d[i].dts, dts[2 * i + 1] = g.d[i].dts_next;
  return 0;
}

}  // extern "C"

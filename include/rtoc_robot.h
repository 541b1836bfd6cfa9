/* rtoc_robot.h -- rigid-body model table + batched linearisation of the contact / impact dynamics.
 *
 * What the reference's Robot gets from Pinocchio (include/robotoc/robot/robot.hxx:524-583 RNEA / RNEADerivatives /
 * RNEAImpact / RNEAImpactDerivatives, :291-360 Baumgarte residual and derivatives, src/robot/point_contact.cpp,
 * include/robotoc/robot/point_contact.hxx:14-140) restated for the device: one model table per context, every
 * (instance, grid point) of RTOC_BUF_SOL linearised in one launch into the pre-condensation fields of RTOC_BUF_CDD.
 * Pinocchio (3rd party, not vendored in the reference tree; robotoc's CMake asks for pinocchio >= 2.x) is absent here:
 * the algorithm is Featherstone's recursive Newton-Euler in body coordinates ([linear; angular] spatial vectors,
 * Pinocchio's convention), its partial derivatives by forward-mode differentiation of that recursion along the
 * 3 nv tangent directions (q on the configuration manifold, v, a) -- parity with Pinocchio itself is UNPINNED
 * (DESIGN.md); the derivatives are validated against finite differences of the CPU restatement.
 *
 * Conventions (Pinocchio's): q = [x y z qx qy qz qw, joint angles] for a floating base, v = [linear; angular] of the
 * base in the BASE frame followed by joint rates; joints in depth-first order, children in name order (what
 * pinocchio::urdf::buildModel produces; tools/urdf_to_model.py restates it).  3x3 matrices row-major.
 */
#ifndef RTOC_ROBOT_H_
#define RTOC_ROBOT_H_

#include "rtoc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RTOC_MAX_JOINTS 48
#define RTOC_MAX_CONTACTS 8

enum rtoc_joint_type { RTOC_JOINT_FREE_FLYER = 0, RTOC_JOINT_REVOLUTE = 1 };
enum rtoc_contact_type { RTOC_CONTACT_POINT = 0, RTOC_CONTACT_SURFACE = 1 }; /* ContactType::PointContact / SurfaceContact */

typedef struct rtoc_robot_model {
  int njoints, nq, nv, ncontacts;
  int parent[RTOC_MAX_JOINTS];             /* index of the parent joint, -1: the world; parent[i] < i                 */
  int type[RTOC_MAX_JOINTS];               /* rtoc_joint_type; a free flyer only as joint 0                           */
  int idx_q[RTOC_MAX_JOINTS];              /* first entry of the joint in q / v                                       */
  int idx_v[RTOC_MAX_JOINTS];
  double placement_R[RTOC_MAX_JOINTS][9];  /* joint frame in the parent joint's frame (pinocchio jointPlacements)     */
  double placement_p[RTOC_MAX_JOINTS][3];
  double axis[RTOC_MAX_JOINTS][3];         /* revolute: unit axis in the joint frame                                  */
  double mass[RTOC_MAX_JOINTS];            /* body of the joint (links behind fixed joints welded in)                 */
  double com[RTOC_MAX_JOINTS][3];          /* centre of mass in the joint frame                                       */
  double inertia[RTOC_MAX_JOINTS][9];      /* rotational inertia about the centre of mass, joint-frame axes           */
  int contact_type[RTOC_MAX_CONTACTS];     /* rtoc_contact_type: 3 rows (force) or 6 rows (wrench); points first, as
                                            * the reference stacks them (robot.hxx:291-320)                           */
  int contact_parent[RTOC_MAX_CONTACTS];   /* joint the contact frame is attached to (PointContact / SurfaceContact)  */
  double contact_R[RTOC_MAX_CONTACTS][9];  /* contact frame in that joint's frame (model.frames[id].placement)        */
  double contact_p[RTOC_MAX_CONTACTS][3];
  double contact_kp[RTOC_MAX_CONTACTS];    /* ContactModelInfo::baumgarte_position_gain / velocity_gain               */
  double contact_kd[RTOC_MAX_CONTACTS];
  double gravity[3];                       /* world frame, (0, 0, -9.81)                                              */
} rtoc_robot_model;

/* Copies the table to the device.  RTOC_ERR_BAD_ARG unless nv == dims.nv, nq is nv (fixed base) or nv + 1 (free-flyer
 * root), joints are depth-first ordered, 3 * ncontacts <= dims.nf_max. */
int rtoc_set_robot_model(rtoc_ctx* ctx, const rtoc_robot_model* model);

/* What rtoc_set_robot_model would plan for rtoc_linearize_contact_dynamics' tangent walk, without a context or a device (host
 * arithmetic only): tree levels, LDS slots for forward tangents (= the largest number of branching joints on a root-to-leaf
 * path), tangent directions per pass (RTOC_OPT_LINEARIZE_DOFS_PER_PASS; forced_dofs_per_pass = 0: the library's choice),
 * passes, LDS bytes per wave of the walk behind the values pre-pass, and pass_bodies[p] = bit i set if pass p visits joint i
 * (room for RTOC_MAX_JOINTS + 8 passes; may be NULL).  RTOC_ERR_BAD_ARG for a table rtoc_set_robot_model would refuse on its
 * own grounds (joints not depth first, index maps inconsistent, more LDS than a CU has). */
typedef struct rtoc_linearize_plan {
  int nlevels, nbranch, dofs_per_pass, npass, lds_bytes;
} rtoc_linearize_plan;
int rtoc_robot_model_plan(const rtoc_robot_model* model, int forced_dofs_per_pass, rtoc_linearize_plan* plan,
                          unsigned long long* pass_bodies);

/* Per grid point: bit k of active[i] = contact k is active (ContactStatus::isContactActive; on impact grids:
 * ImpactStatus::isImpactActive), positions[i][k][0..2] = ContactStatus::contactPosition(k) (world frame; NULL: zeros),
 * rotations[i][k][0..8] = ContactStatus::contactRotation(k) (row-major; only read for surface contacts; NULL: identity).
 * The rows of the active contacts (3 per point, 6 per surface contact) must add up to the grid's dimf. */
int rtoc_set_contact_schedule(rtoc_ctx* ctx, const unsigned* active, const double* positions, const double* rotations);

/* linearizeContactDynamics (src/dynamics/contact_dynamics.cpp:12-33) on intermediate / lift grids and
 * linearizeImpactDynamics (src/dynamics/impact_dynamics.cpp:12-27) on impact grids, for every (instance, grid point)
 * but the terminal one: from RTOC_BUF_SOL (q, v, a | dv, f_stack, u) to RTOC_BUF_CDD
 *   IDC      [ID; C]            ID = RNEA(q, v, a, f) - [0; u]        C = Baumgarte residual (impact: contact velocity);
 *                               point contact: 3 rows, classical linear acceleration (point_contact.hxx:14-31); surface
 *                               contact: 6 rows, spatial acceleration + kp Log6(X_ref^-1 X) (surface_contact.hxx:12-29)
 *   DIDDA    dID/da  (= M(q); impact: dID/ddv)          DCDA   dC/da  (impact: unused, dC/dv is in DIDCDQV)
 *   DIDCDQV  [dID/dq dID/dv; dC/dq dC/dv]
 * i.e. everything computeMJtJinv / condenseContactDynamics read.
 * augment_residual != 0: also the multiplier terms of the same functions (contact_dynamics.cpp:35-52,
 * impact_dynamics.cpp:19-27), ADDED to what the cost / constraint stages left in the residuals (so once per iteration):
 *   lq += dIDdq^T beta + dCdq^T mu   lv += dIDdv^T beta + dCdv^T mu   la += dIDda^T beta + dCda^T mu   lf -= dCda beta
 *   lu -= beta (actuated part)       lu_passive = nu_passive - beta (floating base)
 * (impact grids: ldv instead of la, dCdv instead of dCda, lu_passive = 0); beta, mu_stack, nu_passive from RTOC_BUF_SOL,
 * lq / lv / lu in RTOC_BUF_KKT, la / lf / lu_passive in RTOC_BUF_CDD. */
int rtoc_linearize_contact_dynamics(rtoc_ctx* ctx, int augment_residual);

/* ---- evalKKT of the contact path on the device (no inequality rows) ----
 * {Intermediate,Impact,Terminal}Stage::evalKKT up to the condensation (src/ocp/intermediate_stage.cpp:94-132,
 * impact_stage.cpp:82-114, terminal_stage.cpp:72-100) for an OCP whose cost is the ConfigurationSpaceCost of
 * rtoc_set_configuration_cost: setZero, quadratize{Stage,Impact,Terminal}Cost, linearize{,Impact,Terminal}StateEquation
 * (rtoc_linearize_state_equation), linearizeContactDynamics / linearizeImpactDynamics with the multiplier terms
 * (rtoc_linearize_contact_dynamics(ctx, 1)) and, on grids with GridInfo::switching_constraint, linearizeSwitchingConstraint
 * (src/dynamics/switching_constraint.cpp:26-70: P, Phix, Phia, Phit, the xi terms of lx / la and the STO terms of h, Qtt,
 * hv, ha; the impacting contacts and their positions are those of the impact grid two points ahead in the contact
 * schedule; free-flyer transport: RTOC_OPT_SWITCHING_TRANSPORT) -- RTOC_BUF_SOL in, the pre-condensation records
 * RTOC_BUF_KKT / RTOC_BUF_CDD, RTOC_BUF_SE3 and RTOC_BUF_DX0 out; rtoc_newton_iteration takes it from there.  Needs
 * rtoc_set_robot_model, rtoc_set_contact_schedule, rtoc_set_configuration_cost, rtoc_set_initial_state. */
int rtoc_contact_eval_kkt(rtoc_ctx* ctx);

/* ---- inequality rows of the contact path evaluated on the device (the Constraints object of examples/anymal/trot.cpp:
 * six joint-limit components + FrictionCone) ----
 * Joint limits: the rows of rtoc_set_constraint_rows with their bounds from rtoc_set_constraint_bounds.  Friction cones: the
 * rows of rtoc_set_friction_cones with ContactStatus::frictionCoefficient(k) from rtoc_set_friction_coefficients
 * (mu[k] > 0, k < ncontacts <= RTOC_MAX_CONTACTS) and contactRotation(k) from rtoc_set_contact_schedule's rotations; whether
 * impact grids carry them: RTOC_OPT_IMPACT_CONES.  Once bounds / coefficients are on the device, rtoc_contact_eval_kkt also
 * runs Constraints::linearizeConstraints for those rows -- residual = g + slack, cmpl = slack dual - barrier, the cone
 * Jacobians into RTOC_BUF_CONE, l += dg^T dual (joint_position_lower_limit.cpp:58-77 and its siblings, friction_cone.cpp:
 * 119-191) -- and rtoc_newton_iteration condenses, expands and updates them as before.  Rows without bounds / cones without
 * coefficients stay host-evaluated (RTOC_BUF_CON / RTOC_BUF_CONE uploaded by the caller).
 * rtoc_set_barrier_param: Constraints::setBarrierParam / setFractionToBoundaryRule (also set by rtoc_set_constraint_bounds).
 * rtoc_contact_init_constraints: OCPSolver::initConstraints (src/solver/ocp_solver.cpp:92-96) -- slack = -g clipped at
 * sqrt(barrier), dual = barrier / slack (pdipm.hxx:12-23) at the iterate in RTOC_BUF_SOL; zeroes RTOC_BUF_CON first. */
int rtoc_set_barrier_param(rtoc_ctx* ctx, double barrier_param, double fraction_to_boundary_rule);
int rtoc_set_friction_coefficients(rtoc_ctx* ctx, const double* mu, int ncontacts);
int rtoc_contact_init_constraints(rtoc_ctx* ctx);
/* Contact wrench cones of surface contacts on the device (rtoc_set_wrench_cones rows; ContactWrenchCone,
 * src/constraints/contact_wrench_cone.cpp:114-204): xy_mu[k] = {X, Y, mu} of contact k -- the sole rectangle 2X x 2Y and
 * ContactStatus::frictionCoefficient -- from which the 17 x 6 cone matrices are built (rtoc_wrench_cone_matrix).  Once set,
 * rtoc_contact_init_constraints also writes the matrices into RTOC_BUF_CONE and initialises these rows, and
 * rtoc_contact_eval_kkt evaluates them (g = cone f on the local contact wrench; lf += cone^T dual). */
int rtoc_set_wrench_cone_params(rtoc_ctx* ctx, const double* xy_mu, int ncontacts);
/* OCPSolver::updateSolution (src/solver/ocp_solver.cpp:111-145) for that OCP, one launch sequence: rtoc_contact_eval_kkt,
 * then rtoc_newton_iteration(ctx, 0, fraction_to_boundary_rule).  host_kkt_error[count <= batch] (may be NULL / 0): the KKT
 * error of the iterate it linearised at. */
int rtoc_contact_update_solution(rtoc_ctx* ctx, double fraction_to_boundary_rule, double* host_kkt_error, int count);

/* ---- OCPSolver::solve (src/solver/ocp_solver.cpp:169-213): the iteration schedule, ONE owner for every host shell ----
 * The loop of solve() between its `init_solver` prologue and its epilogue: the STO regularisation of the first
 * initial_sto_reg_iter iterations since the last (re-)discretisation (:171-177), updateSolution, the mesh-refinement branch
 * (KKT error below kkt_tol_mesh and TimeDiscretization::maxTimeStep above max_dt_mesh, :181-199; `inner_iter` restarts at 0 and is
 * incremented by the loop header like the reference's), convergence (:200-210), `iter = max_iter` without it (:212-214).  Pure host
 * logic -- what an iteration, a refinement and the largest time step ARE is the shell's business (callbacks; a non-zero return
 * aborts the loop and is returned): robotoc_amd/solver.py (batched: the KKT error it reports is the largest of the batch) and
 * robotoc::OCPSolver::solve of robotoc_amd/host/robotoc_hip_solver.hpp both run THIS function.  No device call, no context. */
#define RTOC_SOLVE_MAX_REFINEMENTS 64
typedef struct rtoc_solve_options {
  int max_iter;              /* SolverOptions::max_iter */
  double kkt_tol;            /* SolverOptions::kkt_tol */
  int sto_enabled;           /* ocp_.sto_cost && ocp_.sto_constraints (and at least one discrete event) */
  int initial_sto_reg_iter;  /* SolverOptions::initial_sto_reg_iter */
  double initial_sto_reg;    /* SolverOptions::initial_sto_reg */
  double kkt_tol_mesh;       /* SolverOptions::kkt_tol_mesh */
  double max_dt_mesh;        /* SolverOptions::max_dt_mesh */
} rtoc_solve_options;
typedef struct rtoc_solve_callbacks {
  void* user;
  int (*set_sto_regularization)(void* user, double sto_reg);  /* sto_.setRegularization + the statistics' event times; STO problems only */
  int (*update_solution)(void* user, double* kkt_error);      /* updateSolution; *kkt_error = KKTError() */
  int (*max_time_step)(void* user, double* max_dt);           /* time_discretization_.maxTimeStep(); STO problems only */
  int (*mesh_refinement)(void* user);                         /* :185-196: store, discretize, interpolate, initConstraints, clearHistory */
} rtoc_solve_callbacks;
typedef struct rtoc_solve_stats {
  int convergence, iter;
  int num_mesh_refinements;
  int mesh_refinement_iter[RTOC_SOLVE_MAX_REFINEMENTS];       /* iter + 1 of every refinement (the first RTOC_SOLVE_MAX_REFINEMENTS) */
} rtoc_solve_stats;
int rtoc_solve_loop(const rtoc_solve_options* options, const rtoc_solve_callbacks* callbacks, rtoc_solve_stats* stats);

/* ---- filter line search on the device (src/line_search/line_search.cpp:31-83, LineSearchSettings) ----
 * rtoc_contact_eval_ocp: DirectMultipleShooting::evalOCP's performance index (direct_multiple_shooting.cpp:100-126) of every
 * instance -- host_cost[count] = cost + cost_barrier, host_violation[count] = primal_feasibility (l1).  trial = 0: of the iterate
 * rtoc_contact_eval_kkt has just linearised (records not yet condensed: dms_.getEval()); trial = 1: of the trial iterate
 * SOL (+) step DIR with slack + step dslack, step = the primal entry of RTOC_BUF_STEP of every instance
 * (dms_trial_.integratePrimalSolution + evalOCP, line_search.cpp:65-71) -- RTOC_BUF_SOL / CON / DIR / STEP keep their contents, the
 * KKT / CDD records are overwritten (the next rtoc_contact_eval_kkt rewrites them).
 * rtoc_set_line_search: SolverOptions::enable_line_search with LineSearchSettings::step_size_reduction_rate (0.75), min_step_size
 * (0.05), filter_cost_reduction_rate / filter_constraint_violation_reduction_rate (0.005): rtoc_newton_iteration (and
 * rtoc_contact_update_solution) then run LineSearch::computeStepSize between the step-size computation and the update
 * (ocp_solver.cpp:133-139) -- rtoc_contact_line_search: every instance backtracks from its maximum primal step until its filter
 * (rtoc_line_search_filter's, cleared by rtoc_line_search_clear) accepts the trial pair; *host_trials = trial evaluations run. */
int rtoc_contact_eval_ocp(rtoc_ctx* ctx, int trial, double* host_cost, double* host_violation, int count);
int rtoc_set_line_search(rtoc_ctx* ctx, int enable, double step_size_reduction_rate, double min_step_size,
                         double filter_cost_reduction_rate, double filter_constraint_violation_reduction_rate);
int rtoc_contact_line_search(rtoc_ctx* ctx, int* host_trials);
/* LineSearchSettings::line_search_method (include/robotoc/line_search/line_search_settings.hpp:13-29): 0 = LineSearchMethod::Filter
 * (default), 1 = MeritBacktracking -- LineSearch::meritBacktrackingLineSearch (src/line_search/line_search.cpp:87-128): penalty
 * parameter (1 + margin_rate) x max over the grid of SplitSolution::lagrangeMultiplierLinfNorm (:120-128), directional derivative of
 * cost + barrier + penalty x violation from one trial at step eps, then backtracking until armijoCondition (:111-117) holds with
 * armijo_control_rate; the reference's defaults are 0.001, 0.05, 1e-8.  rtoc_line_search_merit_terms: the penalty parameters and
 * directional derivatives of the last rtoc_contact_line_search (either pointer may be NULL).
 * The unconstrained solver is filter-only, like UnconstrLineSearch (src/line_search/unconstr_line_search.cpp, which never reads
 * line_search_method): after rtoc_unconstr_eval_kkt the line search takes the filter path whatever `method` says. */
int rtoc_set_line_search_method(rtoc_ctx* ctx, int method, double armijo_control_rate, double margin_rate, double eps);
int rtoc_line_search_merit_terms(rtoc_ctx* ctx, double* host_penalty, double* host_directional_derivative, int count);
/* trial evaluations (dms_trial_.evalOCP calls, line_search.cpp:70, :99, :109) of the last line search, whichever entry point ran it */
int rtoc_line_search_trials(rtoc_ctx* ctx, int* trials);

/* ---- the unconstrained solver iteration closed on the device (BASELINE configuration 1: fixed base, no contacts) ----
 * ConfigurationSpaceCost (src/cost/configuration_space_cost.cpp:274-470): diagonal weights on q - q_ref (on the manifold:
 * q_ref has nq entries, the first 7 a free-flyer placement if dims.np == 6; weights nv entries, the first 6 on the base's
 * log6 difference), v - v_ref, a, u - u_ref, the terminal and the impact weights; all weights non-negative. */
typedef struct rtoc_configuration_cost {
  double q_ref[RTOC_MAX_JOINTS], v_ref[RTOC_MAX_JOINTS], u_ref[RTOC_MAX_JOINTS];
  double q_weight[RTOC_MAX_JOINTS], v_weight[RTOC_MAX_JOINTS], a_weight[RTOC_MAX_JOINTS], u_weight[RTOC_MAX_JOINTS];
  double q_weight_terminal[RTOC_MAX_JOINTS], v_weight_terminal[RTOC_MAX_JOINTS];
  double q_weight_impact[RTOC_MAX_JOINTS], v_weight_impact[RTOC_MAX_JOINTS], dv_weight_impact[RTOC_MAX_JOINTS];
} rtoc_configuration_cost;
int rtoc_set_configuration_cost(rtoc_ctx* ctx, const rtoc_configuration_cost* cost);
/* (q, v) of OCPSolver / UnconstrOCPSolver::updateSolution(t, q, v) for every instance: x0[batch][nq + nv], nq = nv, or
 * nv + 1 with a free-flyer base (dims.np == 6: [x y z qx qy qz qw, joints]). */
int rtoc_set_initial_state(rtoc_ctx* ctx, const double* x0, int count);
/* linearizeStateEquation (src/dynamics/state_equation.cpp:29-66) on intermediate / lift grids, linearizeImpactStateEquation
 * (impact_state_equation.cpp:27-57) on impact grids, from RTOC_BUF_SOL (q, v, a | dv, lmd, gmm of the grid point, its
 * successor and -- for Fqq_prev -- its predecessor; the initial state of rtoc_set_initial_state ahead of grid point 0):
 * Fx and the top half of Fxx are written, the multiplier and STO terms are ADDED to lx, la (CDD.la), h, hv, ha, fx like
 * the reference adds them to what the cost left there.  Floating base: the SE(3) difference and its Jacobians
 * (Pinocchio's difference / dDifference restated: log6 and Jlog6 by forward mode), and RTOC_BUF_SE3 = {Fqq_inv,
 * Fqq_prev_inv} as correctLinearizeStateEquation computes them -- rtoc_condense then applies the corrections.  With an
 * initial state set: RTOC_BUF_DX0 = computeInitialStateDirection (state_equation.cpp:99-109). */
int rtoc_linearize_state_equation(rtoc_ctx* ctx);
/* UnconstrDirectMultipleShooting::evalKKT up to the condensation, on every grid point: kkt_matrix / kkt_residual zeroed,
 * quadratizeStageCost / TerminalCost of the cost above, linearizeUnconstrForwardEuler(+Terminal)
 * (src/dynamics/unconstr_state_equation.cpp:8-24), linearizeUnconstrDynamics (src/dynamics/unconstr_dynamics.cpp:52-64:
 * RNEA, its derivatives, the dt-scaled multiplier terms), and computeInitialStateDirection (x0 - s[0].x into
 * RTOC_BUF_DX0) -- from RTOC_BUF_SOL into RTOC_BUF_KKT / RTOC_BUF_CDD in the record convention of rtoc_unconstr_condense.
 * With joint-limit rows and bounds set (below): constraints_->linearizeConstraints as well (residual, cmpl into
 * RTOC_BUF_CON, the duals into the gradients). */
int rtoc_unconstr_eval_kkt(rtoc_ctx* ctx, double dt);
/* The joint-limit rows of this path live on the device for their whole life.  After rtoc_set_constraint_rows: the bound
 * of every row, g(z) = sign * z - bound <= 0 (lower limit zmin: sign -1, bound -zmin; upper limit zmax: sign +1, bound
 * zmax), the barrier parameter and the fraction-to-boundary rule (ConstraintsBase setters; 1e-3 / 0.995 in the
 * reference).  rtoc_unconstr_init_constraints = initConstraints (setSlackAndDual at the current iterate). */
int rtoc_set_constraint_bounds(rtoc_ctx* ctx, const double* bounds, int nrows, double barrier_param, double fraction_to_boundary_rule);
int rtoc_unconstr_init_constraints(rtoc_ctx* ctx);
/* UnconstrOCPSolver::updateSolution (src/solver/unconstr_ocp_solver.cpp:96-118): rtoc_unconstr_eval_kkt, the KKT error
 * of the iterate it linearised at (host_kkt_error[count <= batch], may be NULL / 0), rtoc_unconstr_condense, backward,
 * forward, rtoc_unconstr_expand, and SplitSolution::integrate -- RTOC_BUF_SOL holds the next iterate.  Step size 1, or,
 * with joint-limit rows, condenseSlackAndDual / expandSlackAndDual, the fraction-to-boundary step sizes and the slack / dual
 * update as well (unconstr_intermediate_stage.cpp:76-118).  With rtoc_set_line_search(enable = 1): the filter line search of
 * UnconstrLineSearch::computeStepSize (src/line_search/unconstr_line_search.cpp:37-67; unconstr_ocp_solver.cpp:107-111) between
 * the step sizes and the update -- every instance backtracks from its maximum primal step over trial iterates evaluated on
 * the device (integratePrimalSolution + UnconstrDirectMultipleShooting::evalOCP: cost value + log barrier, l1 norm of the
 * state-equation, inverse-dynamics and row residuals); RTOC_BUF_STEP then holds the accepted primal steps.
 * rtoc_contact_eval_ocp / rtoc_contact_line_search serve this path as well: they evaluate with whichever of
 * rtoc_unconstr_eval_kkt / rtoc_contact_eval_kkt ran last. */
int rtoc_unconstr_update_solution(rtoc_ctx* ctx, double dt, double* host_kkt_error, int count);

#ifdef __cplusplus
}
#endif
#endif

// robotoc_hip_solver.hpp -- robotoc::DirectMultipleShooting and robotoc::OCPSolver shells over the C ABI.
//
// Mirrors (reference paths): DirectMultipleShooting  include/robotoc/ocp/direct_multiple_shooting.hpp:32-200,
//                                                    src/ocp/direct_multiple_shooting.cpp:47-266
//                            OCPSolver               include/robotoc/solver/ocp_solver.hpp:41-241,
//                                                    src/solver/ocp_solver.cpp:96-225,429-431
//                            SolverOptions / SolverStatistics  include/robotoc/solver/solver_options.hpp:17,
//                                                              solver_statistics.hpp
// Same class and method names, argument order and call sequence as the reference's updateSolution
// (ocp_solver.cpp:111-145); `updateSolution` is public here (the BASELINE north_star times it).
//
// WHERE THE BOUNDARY RUNS.  In the reference DirectMultipleShooting::evalKKT does two things per grid point:
// it LINEARISES (Pinocchio RNEA derivatives, costs, constraint Jacobians: intermediate_stage.cpp:94-132) and it
// CONDENSES (:134-148).  The first half is the CPU / Pinocchio side of the drop-in boundary; this
// shell takes it from a `StageDataSource` -- in an integration that is robotoc's own IntermediateStage code
// writing its outputs into the packed records (INTEGRATION.md section 5 shows the recorder hook); in this
// repository it is `StageDumpSource`, which replays a recorded stage dump (rtoc_save_stage_dump format).  The
// second half and everything after it -- KKT error, condensation, Riccati recursion, expansions, step sizes,
// slack/dual update, SplitSolution::integrate -- run on the GPU; ONE device context is shared by the
// DirectMultipleShooting and RiccatiRecursion members, so the stage data never leave HBM inside an iteration
// and the host containers (getSolution, getLQRPolicy, getRiccatiFactorization) are filled on demand.
// The STO problem: robotoc::SwitchingTimeOptimization below keeps the reference's method names over the device-resident
// switching-time half of the iteration (rtoc_sto_*: dwell-time rows, per-instance event times and time steps); solve() carries
// the regularisation schedule and the mesh-refinement branch (ocp_solver.cpp:164-199) with the planner mirrors of
// robotoc_hip_planner.hpp.  The line search: SolverOptions::enable_line_search (filter method over rtoc_eval_ocp).
// (rtoc_integrate_solution updates q on the manifold, free-flyer base included.)
#ifndef ROBOTOC_HIP_SOLVER_HPP_
#define ROBOTOC_HIP_SOLVER_HPP_

#include <algorithm>
#include <chrono>
#include <exception>
#include <cmath>
#include <cstdio>
#include <memory>
#include <ostream>

#include "../../include/rtoc_robot.h"
#include "robotoc_hip.hpp"
#include "robotoc_hip_planner.hpp"

namespace robotoc {

// include/robotoc/line_search/line_search_settings.hpp
enum class LineSearchMethod { Filter, MeritBacktracking };
struct LineSearchSettings {
  LineSearchMethod line_search_method = LineSearchMethod::Filter;
  double step_size_reduction_rate = 0.75;
  double min_step_size = 0.05;
  double filter_cost_reduction_rate = 0.005;
  double filter_constraint_violation_reduction_rate = 0.005;
  double armijo_control_rate = 0.001;
  double margin_rate = 0.05;
  double eps = 1.0e-08;
};

// include/robotoc/solver/solver_options.hpp:17-130 (the members the hot path reads)
struct SolverOptions {
  int max_iter = 100;
  double kkt_tol = 1.0e-07;
  double mu_init = 1.0e-03;
  double mu_min = 1.0e-03;
  double kkt_tol_mu = 1.0e-07;
  double mu_linear_decrease_factor = 0.2;
  double mu_superlinear_decrease_power = 1.5;
  bool enable_line_search = false;
  LineSearchSettings line_search_settings;
  double fraction_to_boundary_rule = 0.995;  // ConstraintComponentBase default (constraint_component_base.hpp)
  int initial_sto_reg_iter = 0;              // solver_options.hpp:96-113
  double initial_sto_reg = 1.0e30;
  double kkt_tol_mesh = 0.1;
  double max_dt_mesh = 0.0;                  // solver_options.hpp:119
  double max_dts_riccati = 0.1;
  bool enable_solution_interpolation = true;
  bool enable_benchmark = false;
  int horizon_scan = 0;  // not in the reference: RTOC_OPT_BACKWARD_SCAN -- 0 (default) serial kernels, 1 horizon scans, 2 as the
                         // Python shell's "auto": scans for batches of at most 8 OCPs, the latency path of one MPC problem
};

// include/robotoc/solver/solver_statistics.hpp
struct SolverStatistics {
  bool convergence = false;
  int iter = 0;
  std::vector<double> performance_index;  // per iteration: PerformanceIndex::kkt_error, i.e. the SQUARED KKT error, in both solver shells
  std::vector<double> primal_step_size, dual_step_size;
  std::vector<std::vector<double>> ts;    // event times ahead of every iteration, if the OCP has an STO problem
  std::vector<int> mesh_refinement_iter;  // iterations after which the mesh was refined
  double cpu_time = 0.0;  // [ms], if SolverOptions::enable_benchmark
  void clear() { *this = SolverStatistics(); }
};

// include/robotoc/core/performance_index.hpp: the entry the convergence test reads
struct PerformanceIndex {
  double kkt_error = 0.0;  // squared, like the reference accumulates it (sqrt in OCPSolver::KKTError())
  double cost_plus_barrier = 0.0, primal_feasibility = 0.0;   // cost + cost_barrier, l1 violation (with the line search on)
};

// include/robotoc/core/split_solution.hpp: the members SplitSolution::integrate updates
class SplitSolution {
 public:
  SplitSolution() {}
  explicit SplitSolution(const RobotDims& r)
      : q(r.dimv + (r.dim_passive > 0 ? 1 : 0)), v(r.dimv), a(r.dimv), u(r.dimu), f_full(r.max_dimf), lmd(r.dimv),
        gmm(r.dimv), beta(r.dimv), mu_full(r.max_dimf), nu_passive(r.dim_passive), xi_full(r.max_dimf) {}
  Vec q, v, a, u, f_full, lmd, gmm, beta, mu_full, nu_passive, xi_full;
};
typedef std::vector<SplitSolution> Solution;

// The linearisation half of evalKKT (see the header comment).  linearize() must leave the PRE-condensation stage
// data of the current iterate in the device buffers RTOC_BUF_KKT, RTOC_BUF_CDD and, if the problem has them,
// RTOC_BUF_CON / RTOC_BUF_CONE / RTOC_BUF_SE3; dx0() the initial state direction q0 (-) q, v0 - v.
class StageDataSource {
 public:
  virtual ~StageDataSource() {}
  virtual RobotDims robot() const = 0;
  virtual int ncMax() const = 0;
  virtual const TimeDiscretization& timeDiscretization() const = 0;
  virtual void configure(rtoc_ctx* ctx) = 0;  // constraint rows, cones: once after rtoc_create
  // the (q, v) of updateSolution(t, q, v), ahead of linearize(): sources that linearise on the device need it for
  // Fqq_prev of grid point 0 and the initial state direction; host-side sources ignore it
  virtual void setInitialState(rtoc_ctx*, const Vec&, const Vec&) {}
  virtual void linearize(rtoc_ctx* ctx, const Solution& s) = 0;
  // slack / dual start values (OCPSolver::initConstraints).  Host-side sources deliver them with the linearised stage data
  // (RTOC_BUF_CON), so the default only pulls the first linearisation in; device-side sources initialise them there.
  virtual void initConstraints(rtoc_ctx* ctx, const Solution& s) { linearize(ctx, s); }
  virtual void initialStateDirection(const Vec& q, const Vec& v, const Solution& s, Vec& dx0) const = 0;
  // true: linearize() already left computeInitialStateDirection's result in RTOC_BUF_DX0 (nothing to compute or upload here)
  virtual bool initialStateDirectionOnDevice() const { return false; }
  // false: linearize() / initialStateDirection() never read their `s` argument (the source linearises at the iterate in
  // RTOC_BUF_SOL on the device), so OCPSolver::updateSolution need not download the iterate before calling them
  virtual bool needsHostSolution() const { return true; }
  // ---- sources that own the contact sequence can re-discretise (OCPSolver::discretize, mesh refinement) ----
  // the contact sequence with its current event times, or nullptr (the grid comes with the stage data and is fixed)
  virtual ContactSequence* contactSequence() { return nullptr; }
  virtual const STOConstraints* stoConstraints() const { return nullptr; }   // non-null: the OCP has an STO problem
  virtual double horizonLength() const { return 0.0; }                        // OCP::T
  // TimeDiscretization::discretize(contact_sequence, t) (+ correctTimeSteps when phase based) from the sequence's current
  // event times; the next linearize() / initConstraints() must hand the new contact schedule to the device
  virtual bool discretize(const double) { return false; }
  // the most grid points a re-discretisation can produce (mesh refinement moves grid points between the phases of a PhaseBased
  // discretisation): N + 1 + lifts + 2 impacts for a source that owns its contact sequence; the solver reserves that many
  virtual int maxGridPoints() const { return timeDiscretization().size(); }
  virtual const std::vector<unsigned>* contactMasks() const { return nullptr; }
  virtual void initialSolution(Solution& s) const = 0;
};

// Replays a recorded stage dump (rtoc_save_stage_dump; robotoc_amd/replay.py writes the same format): the
// linearisation of ONE iterate of ONE problem instance.  Every linearize() call re-uploads it -- the GPU side of
// the iteration is exercised in full, the iterate-dependence of the linearisation is not (that needs robotoc +
// Pinocchio on the host).
class StageDumpSource : public StageDataSource {
 public:
  explicit StageDumpSource(const std::string& path, const int instance = 0) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("[StageDumpSource] cannot open " + path);
    bool ok = std::fread(&h_, sizeof(h_), 1, f) == 1 && std::memcmp(h_.magic, "RTOCDMP1", 8) == 0 && h_.version == 1 &&
              h_.header_bytes == sizeof(h_) && instance >= 0 && instance < h_.batch;
    std::vector<rtoc_grid> grid(ok ? h_.nstages : 0);
    rows_.resize(ok && h_.nrows > 0 ? h_.nrows : 0);
    ok = ok && std::fread(grid.data(), sizeof(rtoc_grid), grid.size(), f) == grid.size() &&
         (rows_.empty() || std::fread(rows_.data(), sizeof(rtoc_box_row), rows_.size(), f) == rows_.size());
    for (int b = 0; ok && b < 16; ++b) {
      if (!h_.count[b]) continue;
      // keep the records of `instance` only: buffers are [batch][...]
      const size_t per = h_.count[b] / h_.batch;
      std::vector<double> all(h_.count[b]);
      ok = std::fread(all.data(), sizeof(double), all.size(), f) == all.size();
      if (ok && b < RTOC_NUM_BUFFERS) buf_[b].assign(all.begin() + instance * per, all.begin() + (instance + 1) * per);
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("[StageDumpSource] malformed stage dump " + path);
    std::vector<GridInfo> gi(h_.nstages);
    for (int i = 0; i < h_.nstages; ++i) {
      gi[i].type = static_cast<GridType>(grid[i].type);
      gi[i].dt = grid[i].dt;
      gi[i].sto = grid[i].sto != 0;
      gi[i].sto_next = grid[i].sto_next != 0;
      gi[i].switching_constraint = grid[i].switching_constraint != 0;
      gi[i].dimf = grid[i].dimf;
      gi[i].dims = grid[i].dims;
      gi[i].num_grids_in_phase = grid[i].num_grids_in_phase;
      gi[i].stage = grid[i].time_stage < 0 ? 0 : grid[i].time_stage;
    }
    td_ = TimeDiscretization(gi);
  }
  RobotDims robot() const override { return RobotDims{h_.dims.nv, h_.dims.nu, h_.dims.np, h_.dims.nf_max}; }
  int ncMax() const override { return h_.dims.nc_max; }
  const TimeDiscretization& timeDiscretization() const override { return td_; }
  void configure(rtoc_ctx* ctx) override {
    if (!rows_.empty()) chk(rtoc_set_constraint_rows(ctx, rows_.data(), static_cast<int>(rows_.size())), "rtoc_set_constraint_rows");
    if (h_.cone_contacts > 0)
      chk(h_.cone_rows == RTOC_WRENCH_ROWS ? rtoc_set_wrench_cones(ctx, h_.cone_contacts)
                                           : rtoc_set_friction_cones(ctx, h_.cone_contacts, h_.cone_dim),
          "rtoc_set_*_cones");
  }
  void linearize(rtoc_ctx* ctx, const Solution&) override {
    const int bufs[] = {RTOC_BUF_KKT, RTOC_BUF_CDD, RTOC_BUF_CON, RTOC_BUF_CONE, RTOC_BUF_SE3};
    for (int b : bufs)
      if (!buf_[b].empty()) chk(rtoc_upload(ctx, b, 0, buf_[b].data(), buf_[b].size()), "rtoc_upload");
  }
  // q0 (-) q, v0 - v: the recorded direction if the dump has one (floating bases: the SE3 difference is Pinocchio's),
  // else the Euclidean difference
  void initialStateDirection(const Vec& q, const Vec& v, const Solution& s, Vec& dx0) const override {
    const int nv = h_.dims.nv;
    if (!buf_[RTOC_BUF_DX0].empty()) {
      for (int i = 0; i < 2 * nv; ++i) dx0(i) = buf_[RTOC_BUF_DX0][i];
      return;
    }
    if (h_.dims.np > 0) throw std::logic_error("[StageDumpSource] floating base: the dump must carry RTOC_BUF_DX0");
    for (int i = 0; i < nv; ++i) {
      dx0(i) = q(i) - s[0].q(i);
      dx0(nv + i) = v(i) - s[0].v(i);
    }
  }
  void initialSolution(Solution& s) const override {
    if (buf_[RTOC_BUF_SOL].empty()) return;
    rtoc_layout L;
    rtoc_compute_layout(&h_.dims, &L);
    for (int i = 0; i < h_.nstages && i < static_cast<int>(s.size()); ++i) unpackSolution(L, &buf_[RTOC_BUF_SOL][static_cast<size_t>(i) * L.sol.stride], s[i]);
  }
  const std::vector<double>& buffer(int b) const { return buf_[b]; }

  static void unpackSolution(const rtoc_layout& L, const double* r, SplitSolution& s) {
    const int* o = L.sol.off;
    auto get = [&](Vec& v, int f) { std::memcpy(v.data(), r + o[f], sizeof(double) * v.size()); };
    get(s.q, RTOC_SOL_Q);
    get(s.v, RTOC_SOL_V);
    get(s.a, RTOC_SOL_A);
    get(s.u, RTOC_SOL_U);
    get(s.f_full, RTOC_SOL_F);
    get(s.lmd, RTOC_SOL_LMD);
    get(s.gmm, RTOC_SOL_GMM);
    get(s.beta, RTOC_SOL_BETA);
    get(s.mu_full, RTOC_SOL_MU);
    get(s.nu_passive, RTOC_SOL_NUP);
    get(s.xi_full, RTOC_SOL_XI);
  }
  static void packSolution(const rtoc_layout& L, const SplitSolution& s, double* r) {
    const int* o = L.sol.off;
    auto put = [&](const Vec& v, int f) { std::memcpy(r + o[f], v.data(), sizeof(double) * v.size()); };
    put(s.q, RTOC_SOL_Q);
    put(s.v, RTOC_SOL_V);
    put(s.a, RTOC_SOL_A);
    put(s.u, RTOC_SOL_U);
    put(s.f_full, RTOC_SOL_F);
    put(s.lmd, RTOC_SOL_LMD);
    put(s.gmm, RTOC_SOL_GMM);
    put(s.beta, RTOC_SOL_BETA);
    put(s.mu_full, RTOC_SOL_MU);
    put(s.nu_passive, RTOC_SOL_NUP);
    put(s.xi_full, RTOC_SOL_XI);
  }

 private:
  static void chk(int rc, const char* what) {
    if (rc != RTOC_OK) throw std::runtime_error(std::string("[StageDumpSource] ") + what + ": " + rtoc_error_string(rc));
  }
  rtoc_dump_header h_;
  std::vector<rtoc_box_row> rows_;
  std::vector<double> buf_[16];
  TimeDiscretization td_;
};

// robotoc::OCP for the shells: the reference's OCP carries robot, cost, constraints, contact sequence, T, N
// (ocp.hpp:22-147); everything upstream of the evalKKT boundary is represented by the stage-data source.
struct SolverOCP : public OCP {
  std::shared_ptr<StageDataSource> source;
  int device = 0;
  explicit SolverOCP(const std::shared_ptr<StageDataSource>& src, const int dev = 0) : source(src), device(dev) {
    robot = src->robot();
    N = src->timeDiscretization().size() - 1;
    reserved_num_discrete_events = std::max(0, src->maxGridPoints() - (N + 1));   // (ocp.hpp: reserved_num_discrete_events)
  }
  SolverOCP() {}
};

// One device context shared by the solver's members; destroyed with the last owner, cloned on copy.
class DeviceContext {
 public:
  DeviceContext(const RobotDims& robot, const int nc_max, const int max_stages, const int device) : ctx_(nullptr) {
    rtoc_dims d = robot.c();
    d.nc_max = nc_max;
    chk(rtoc_create(&d, max_stages, 1, device, &ctx_), "rtoc_create");
  }
  DeviceContext(const DeviceContext& o) : ctx_(nullptr) { chk(rtoc_clone(o.ctx_, &ctx_), "rtoc_clone"); }
  DeviceContext& operator=(const DeviceContext&) = delete;
  ~DeviceContext() {
    if (ctx_) rtoc_destroy(ctx_);
  }
  rtoc_ctx* get() const { return ctx_; }

 private:
  static void chk(int rc, const char* what) {
    if (rc != RTOC_OK) throw std::runtime_error(std::string("[DeviceContext] ") + what + ": " + rtoc_error_string(rc));
  }
  rtoc_ctx* ctx_;
};

class DirectMultipleShooting {
 public:
  // reference: DirectMultipleShooting(const OCP& ocp, const int nthreads).  nthreads sized the OpenMP team of the
  // per-stage loops (direct_multiple_shooting.cpp:52,77,106,135,180,219,250); here every loop is one kernel launch
  // over all grid points and the argument is kept for signature compatibility only.
  DirectMultipleShooting(const SolverOCP& ocp, const int nthreads, const std::shared_ptr<DeviceContext>& dev)
      : nthreads_(nthreads), source_(ocp.source), dev_(dev), tau_(0.995), max_primal_(1.0), max_dual_(1.0) {
    if (nthreads <= 0) throw std::out_of_range("[DirectMultipleShooting] invalid argument: nthreads must be positive!");
    if (!dev || !ocp.source) throw std::invalid_argument("[DirectMultipleShooting] null device context / stage-data source");
    chk(rtoc_get_layout(dev_->get(), &L_), "rtoc_get_layout");
  }
  DirectMultipleShooting() : nthreads_(0), tau_(0.995), max_primal_(1.0), max_dual_(1.0) {}

  void setNumThreads(const int nthreads) {
    if (nthreads <= 0) throw std::out_of_range("[DirectMultipleShooting] invalid argument: nthreads must be positive!");
    nthreads_ = nthreads;
  }
  void setFractionToBoundaryRule(const double tau) {
    if (!(tau > 0.0 && tau <= 1.0)) throw std::out_of_range("[DirectMultipleShooting] fraction-to-boundary rule must be in (0, 1]");
    tau_ = tau;
  }
  void rebind(const std::shared_ptr<DeviceContext>& dev) { dev_ = dev; }

  // initConstraints (direct_multiple_shooting.cpp:47-69): slack / dual start values.  They are part of the
  // linearised stage data the source delivers (RTOC_BUF_CON), so this only pulls the first linearisation in.
  void initConstraints(const TimeDiscretization& td, const Solution& s) {
    RiccatiRecursion::setGridOf(ctx(), td);
    source_->initConstraints(ctx(), s);
  }
  // isFeasible (direct_multiple_shooting.cpp:72-97): all slacks positive
  bool isFeasible(const TimeDiscretization& td, const Solution&) {
    const size_t n = rtoc_buffer_count(ctx(), RTOC_BUF_CON);
    if (n == 0 || L_.dims.nc_max == 0) return true;
    std::vector<double> con(static_cast<size_t>(td.size()) * L_.con.stride);
    if (!rtoc_device_ptr(ctx(), RTOC_BUF_CON)) return true;  // no inequality rows on this context
    chk(rtoc_download(ctx(), RTOC_BUF_CON, 0, con.data(), con.size()), "rtoc_download(RTOC_BUF_CON)");  // an error is an error, not "feasible"
    // rows that are never written stay zero; the active ones carry positive slacks
    for (int i = 0; i + 1 < td.size(); ++i)
      for (int r = 0; r < L_.dims.nc_max; ++r) {
        const double sl = con[static_cast<size_t>(i) * L_.con.stride + L_.con.off[RTOC_CON_SLACK] + r];
        if (sl < 0.0) return false;
      }
    return true;
  }
  // evalOCP (direct_multiple_shooting.cpp:100-126): cost / feasibility of the iterate without the Hessians;
  // here: linearise and evaluate the KKT error, no condensation
  void evalOCP(const TimeDiscretization& td, const Vec&, const Vec&, const Solution& s, KKTResidual&) {
    RiccatiRecursion::setGridOf(ctx(), td);
    source_->linearize(ctx(), s);
    double e = 0.0;
    chk(rtoc_kkt_error(ctx(), &e, 1), "rtoc_kkt_error");
    performance_index_.kkt_error = e * e;
  }
  // evalKKT (direct_multiple_shooting.cpp:129-159): linearise (source), KKT error of the linearised records
  // (:155-158), then the condensation tail of every stage (intermediate_stage.cpp:134-148) on the GPU.
  // kkt_matrix / kkt_residual stay device resident (the reference's containers are not filled).
  void evalKKT(const TimeDiscretization& td, const Vec& q, const Vec& v, const Solution& s, KKTMatrix&, KKTResidual&) {
    RiccatiRecursion::setGridOf(ctx(), td);
    source_->setInitialState(ctx(), q, v);
    source_->linearize(ctx(), s);
    chk(rtoc_clear_status(ctx()), "rtoc_clear_status");
    double e = 0.0;
    chk(rtoc_kkt_error(ctx(), &e, 1), "rtoc_kkt_error");
    performance_index_.kkt_error = e * e;
    if (line_search_) {   // cost + cost_barrier, primal_feasibility of the iterate: what LineSearch reads of dms.getEval()
      double c = 0.0, v = 0.0;
      chk(rtoc_contact_eval_ocp(ctx(), 0, &c, &v, 1), "rtoc_contact_eval_ocp");
      performance_index_.cost_plus_barrier = c, performance_index_.primal_feasibility = v;
    }
    chk(rtoc_condense(ctx()), "rtoc_condense");
  }
  void setLineSearch(const bool on) { line_search_ = on; }
  // computeInitialStateDirection (direct_multiple_shooting.cpp:162-171 -> state_equation.cpp:98-109)
  void computeInitialStateDirection(const Vec& q, const Vec& v, const Solution& s, Direction& d) const {
    if (source_->initialStateDirectionOnDevice()) return;
    source_->initialStateDirection(q, v, s, d[0].dx);
    chk(rtoc_upload(ctx(), RTOC_BUF_DX0, 0, d[0].dx.data(), d[0].dx.size()), "rtoc_upload");
  }
  const PerformanceIndex& getEval() const { return performance_index_; }
  // computeStepSizes (direct_multiple_shooting.cpp:174-199): expandPrimal + fraction-to-boundary of every stage,
  // min-reduced over the horizon on the device (the reference reduces max_*_step_sizes_ on the host, :202-209)
  void computeStepSizes(const TimeDiscretization&, Direction&) {
    chk(rtoc_expand(ctx(), tau_), "rtoc_expand");
    double st[2] = {1.0, 1.0};
    chk(rtoc_download(ctx(), RTOC_BUF_STEP, 0, st, 2), "rtoc_download");
    max_primal_ = st[0];
    max_dual_ = st[1];
  }
  double maxPrimalStepSize() const { return max_primal_; }
  double maxDualStepSize() const { return max_dual_; }
  // integrateSolution (direct_multiple_shooting.cpp:212-241): expandDual was done with the step sizes; slack /
  // dual update and SplitSolution::integrate with the given steps
  void integrateSolution(const TimeDiscretization&, const double primal_step_size, const double dual_step_size, Direction&,
                         Solution&) {
    const double st[2] = {primal_step_size, dual_step_size};
    chk(rtoc_upload(ctx(), RTOC_BUF_STEP, 0, st, 2), "rtoc_upload");
    chk(rtoc_update(ctx()), "rtoc_update");
    chk(rtoc_integrate_solution(ctx()), "rtoc_integrate_solution");
  }
  // integratePrimalSolution (direct_multiple_shooting.cpp:244-266): primal part only (line search trial points)
  void integratePrimalSolution(const TimeDiscretization&, const double primal_step_size, const Direction&, Solution&) {
    const double st[2] = {primal_step_size, 0.0};
    chk(rtoc_upload(ctx(), RTOC_BUF_STEP, 0, st, 2), "rtoc_upload");
    chk(rtoc_integrate_solution(ctx()), "rtoc_integrate_solution");
  }
  void resizeData(const TimeDiscretization& td) {
    if (static_cast<size_t>(td.size()) * L_.kkt.stride > rtoc_buffer_count(ctx(), RTOC_BUF_KKT))
      throw std::out_of_range("[DirectMultipleShooting] discretisation larger than the device context");
  }

 private:
  static void chk(int rc, const char* what) {
    if (rc != RTOC_OK) throw std::runtime_error(std::string("[DirectMultipleShooting] ") + what + ": " + rtoc_error_string(rc));
  }
  rtoc_ctx* ctx() const {
    if (!dev_) throw std::logic_error("[DirectMultipleShooting] default-constructed object");
    return dev_->get();
  }
  int nthreads_;
  std::shared_ptr<StageDataSource> source_;
  std::shared_ptr<DeviceContext> dev_;
  rtoc_layout L_;
  double tau_;
  PerformanceIndex performance_index_;
  double max_primal_, max_dual_;
  bool line_search_ = false;
};

// robotoc::SwitchingTimeOptimization (include/robotoc/sto/switching_time_optimization.hpp, src/sto/switching_time_optimization.cpp)
// over the device: the dwell-time rows, event times and time steps of the instance live in the context (rtoc_sto_*).
class SwitchingTimeOptimization {
 public:
  SwitchingTimeOptimization() : enabled_(false) {}
  SwitchingTimeOptimization(const std::shared_ptr<StageDataSource>& source, const std::shared_ptr<DeviceContext>& dev)
      : source_(source), dev_(dev), enabled_(source && source->stoConstraints() && source->contactSequence()), sto_reg_(0.0) {}
  void rebind(const std::shared_ptr<DeviceContext>& dev) { dev_ = dev; }
  bool enabled() const { return enabled_ && source_->contactSequence()->numDiscreteEvents() > 0; }
  // hands the contact sequence's event times and the STOConstraints to the device (after every rtoc_set_grid)
  void setProblem(const double t) {
    if (!enabled()) return;
    const ContactSequence& cs = *source_->contactSequence();
    const STOConstraints& sc = *source_->stoConstraints();
    if (static_cast<int>(sc.minimum_dwell_times.size()) != cs.numDiscreteEvents() + 1)
      throw std::runtime_error("[STOConstraints] : invalid size of minimum_dwell_times_ is detected! It should be " + std::to_string(cs.numDiscreteEvents() + 1));
    chk(rtoc_sto_set_problem(ctx(), t, source_->horizonLength(), cs.eventTimes().data(), cs.numDiscreteEvents(), 0,
                             sc.minimum_dwell_times.data(), sc.barrier_param, sc.fraction_to_boundary_rule), "rtoc_sto_set_problem");
    chk(rtoc_sto_set_regularization(ctx(), sto_reg_), "rtoc_sto_set_regularization");
  }
  void setRegularization(const double sto_reg) {
    sto_reg_ = sto_reg;
    if (enabled()) chk(rtoc_sto_set_regularization(ctx(), sto_reg), "rtoc_sto_set_regularization");
  }
  void initConstraints(const TimeDiscretization&) {
    if (enabled()) chk(rtoc_sto_init_constraints(ctx()), "rtoc_sto_init_constraints");
  }
  void evalKKT(const TimeDiscretization&, KKTMatrix&, KKTResidual&) {
    performance_index_.kkt_error = 0.0;
    if (!enabled()) return;
    chk(rtoc_sto_eval_kkt_device(ctx()), "rtoc_sto_eval_kkt_device");
    chk(rtoc_sto_get_kkt_terms(ctx(), nullptr, nullptr, &performance_index_.kkt_error, 1), "rtoc_sto_get_kkt_terms");
  }
  void computeStepSizes(const TimeDiscretization&, const Direction&) {
    max_primal_ = max_dual_ = 1.0;
    if (!enabled()) return;
    // the device folds the rows' fraction-to-boundary steps into RTOC_BUF_STEP (the stages' steps are there already)
    chk(rtoc_sto_compute_step_sizes(ctx()), "rtoc_sto_compute_step_sizes");
    double st[2] = {1.0, 1.0};
    chk(rtoc_download(ctx(), RTOC_BUF_STEP, 0, st, 2), "rtoc_download");
    max_primal_ = st[0], max_dual_ = st[1];
  }
  double maxPrimalStepSize() const { return max_primal_; }
  double maxDualStepSize() const { return max_dual_; }
  // event times of the contact sequence += primal step x dts, slack / dual of the rows (:181-206); the steps are those
  // DirectMultipleShooting::integrateSolution left in RTOC_BUF_STEP
  void integrateSolution(const TimeDiscretization&, const double, const double, const Direction&) {
    if (!enabled()) return;
    chk(rtoc_sto_integrate_solution(ctx()), "rtoc_sto_integrate_solution");
    syncEventTimes();
  }
  void syncEventTimes() {
    if (!enabled()) return;
    std::vector<double> ts(source_->contactSequence()->numDiscreteEvents());
    chk(rtoc_sto_get_event_times(ctx(), ts.data(), 1), "rtoc_sto_get_event_times");
    source_->contactSequence()->setEventTimes(ts);
  }
  const PerformanceIndex& getEval() const { return performance_index_; }

 private:
  static void chk(int rc, const char* what) {
    if (rc != RTOC_OK) throw std::runtime_error(std::string("[SwitchingTimeOptimization] ") + what + ": " + rtoc_error_string(rc));
  }
  rtoc_ctx* ctx() const { return dev_->get(); }
  std::shared_ptr<StageDataSource> source_;
  std::shared_ptr<DeviceContext> dev_;
  bool enabled_;
  double sto_reg_, max_primal_ = 1.0, max_dual_ = 1.0;
  PerformanceIndex performance_index_;
};

class OCPSolver {
 public:
  // reference: OCPSolver(const OCP& ocp, const SolverOptions& solver_options = SolverOptions()) (ocp_solver.cpp:14-49)
  explicit OCPSolver(const SolverOCP& ocp, const SolverOptions& solver_options = SolverOptions())
      : ocp_(ocp), solver_options_(solver_options) {
    if (!ocp.source) throw std::invalid_argument("[OCPSolver] the OCP carries no stage-data source");
    if (ocp.N <= 0) throw std::out_of_range("[OCPSolver] invalid argument: ocp.N must be positive!");
    const int n = ocp.N + 1 + ocp.reserved_num_discrete_events;  // ocp_solver.cpp:20-24
    dev_ = std::make_shared<DeviceContext>(ocp.robot, ocp.source->ncMax(), n, ocp.device);
    ocp.source->configure(dev_->get());
    dms_ = DirectMultipleShooting(ocp, 1, dev_);
    sto_ = SwitchingTimeOptimization(ocp.source, dev_);
    riccati_recursion_ = RiccatiRecursion(ocp, dev_->get());
    kkt_matrix_.assign(n, SplitKKTMatrix(ocp.robot));
    kkt_residual_.assign(n, SplitKKTResidual(ocp.robot));
    s_.assign(n, SplitSolution(ocp.robot));
    d_.assign(n, SplitDirection(ocp.robot));
    riccati_factorization_.assign(n, SplitRiccatiFactorization(ocp.robot));
    rtoc_get_layout(dev_->get(), &L_);
    setSolverOptions(solver_options);
    ocp.source->initialSolution(s_);
    uploadSolution();
  }
  OCPSolver() {}
  // value semantics like the reference (ocp_solver.hpp:62-77): a copy owns a deep copy of the device context
  OCPSolver(const OCPSolver& o)
      : ocp_(o.ocp_), time_discretization_(o.time_discretization_), kkt_matrix_(o.kkt_matrix_), kkt_residual_(o.kkt_residual_),
        s_(o.s_), d_(o.d_), riccati_factorization_(o.riccati_factorization_), solver_options_(o.solver_options_),
        solver_statistics_(o.solver_statistics_), L_(o.L_), host_solution_valid_(o.host_solution_valid_) {
    solution_interpolator_ = o.solution_interpolator_;
    t_ = o.t_;
    if (o.dev_) {
      dev_ = std::make_shared<DeviceContext>(*o.dev_);
      dms_ = o.dms_;
      dms_.rebind(dev_);
      sto_ = o.sto_;
      sto_.rebind(dev_);
      riccati_recursion_ = RiccatiRecursion(ocp_, dev_->get());
    }
  }
  OCPSolver& operator=(const OCPSolver& o) {
    if (this != &o) {
      OCPSolver tmp(o);
      *this = std::move(tmp);
    }
    return *this;
  }
  OCPSolver(OCPSolver&&) = default;
  OCPSolver& operator=(OCPSolver&&) = default;

  void setSolverOptions(const SolverOptions& solver_options) {
    if (solver_options.enable_line_search && ocp_.source->needsHostSolution())
      throw std::logic_error("[OCPSolver] the line search evaluates trial iterates on the device: it needs a source that linearises there");
    solver_options_ = solver_options;
    const LineSearchSettings& ls = solver_options.line_search_settings;   // line_search_.set(...) (ocp_solver.cpp:86)
    chk(rtoc_set_line_search(dev_->get(), solver_options.enable_line_search ? 1 : 0, ls.step_size_reduction_rate, ls.min_step_size,
                             ls.filter_cost_reduction_rate, ls.filter_constraint_violation_reduction_rate), "rtoc_set_line_search");
    chk(rtoc_set_line_search_method(dev_->get(), ls.line_search_method == LineSearchMethod::MeritBacktracking ? 1 : 0, ls.armijo_control_rate,
                                    ls.margin_rate, ls.eps), "rtoc_set_line_search_method");
    dms_.setLineSearch(solver_options.enable_line_search);
    dms_.setFractionToBoundaryRule(solver_options.fraction_to_boundary_rule);
    riccati_recursion_.setRegularization(solver_options.max_dts_riccati);   // ocp_solver.cpp:84
    riccati_recursion_.setHorizonScanMode(solver_options.horizon_scan);
  }

  // discretize (ocp_solver.cpp:96-102): the contact-sequence planner is upstream of the boundary; the grid comes
  // with the stage data
  // with a source that owns its contact sequence: TimeDiscretization::discretize (+ correctTimeSteps, PhaseBased, when the OCP
  // has an STO problem: ocp_solver.cpp:46-48, 96-102) at the sequence's current event times; the event times and the
  // STOConstraints then go to the device
  void discretize(const double t) {
    ocp_.source->discretize(t);
    time_discretization_ = ocp_.source->timeDiscretization();
    if (time_discretization_.size() > static_cast<int>(s_.size()))
      throw std::out_of_range("[OCPSolver] the discretisation has more grid points than the solver reserved (ocp.N + 1 + reserved_num_discrete_events)");
    dms_.resizeData(time_discretization_);
    riccati_recursion_.resizeData(time_discretization_);
    RiccatiRecursion::setGridOf(dev_->get(), time_discretization_);
    sto_.setProblem(t);
    t_ = t;
  }
  void initConstraints() {
    if (time_discretization_.size() < 2) discretize(0.0);
    dms_.initConstraints(time_discretization_, s_);
    sto_.initConstraints(time_discretization_);                                                 // :107
  }

  // One Newton / SQP iteration: the reference's updateSolution (ocp_solver.cpp:111-145), same order of calls.
  void updateSolution(const double t, const Vec& q, const Vec& v) {
    (void)t;
    if (time_discretization_.size() < 2) discretize(t);
    // the iterate lives in RTOC_BUF_SOL: a source that linearises on the host must see the current one, not the initial guess
    if (ocp_.source->needsHostSolution()) syncSolution();
    // (:115-117, correctTimeSteps of a PhaseBased discretisation: rtoc_contact_eval_kkt opens with it; host-side sources
    // linearise at the grid they were given)
    dms_.evalKKT(time_discretization_, q, v, s_, kkt_matrix_, kkt_residual_);                  // :118
    sto_.evalKKT(time_discretization_, kkt_matrix_, kkt_residual_);                             // :119
    riccati_recursion_.backwardRiccatiRecursionResident(time_discretization_);                  // :120
    dms_.computeInitialStateDirection(q, v, s_, d_);                                            // :123
    if (ocp_.robot.dim_passive > 0 && hasSE3()) rtoc_compute_initial_state_direction(dev_->get());
    riccati_recursion_.forwardRiccatiRecursionResident();                                       // :124
    dms_.computeStepSizes(time_discretization_, d_);                                            // :127
    sto_.computeStepSizes(time_discretization_, d_);                                            // :128
    double primal_step_size = std::min(dms_.maxPrimalStepSize(), sto_.maxPrimalStepSize());         // :129-132
    const double dual_step_size = std::min(dms_.maxDualStepSize(), sto_.maxDualStepSize());
    if (solver_options_.enable_line_search) {                                                   // :133-139
      // line_search_.computeStepSize: the filter's backtracking loop over trial iterates evaluated on the device
      // (rtoc_contact_line_search; the maximum steps are where computeStepSizes left them, in RTOC_BUF_STEP)
      const double st[2] = {primal_step_size, dual_step_size};
      chk(rtoc_upload(dev_->get(), RTOC_BUF_STEP, 0, st, 2), "rtoc_upload");
      chk(rtoc_contact_line_search(dev_->get(), nullptr), "rtoc_contact_line_search");
      chk(rtoc_download(dev_->get(), RTOC_BUF_STEP, 0, &primal_step_size, 1), "rtoc_download");
    }
    solver_statistics_.primal_step_size.push_back(primal_step_size);                            // :140-141
    solver_statistics_.dual_step_size.push_back(dual_step_size);
    dms_.integrateSolution(time_discretization_, primal_step_size, dual_step_size, d_, s_);     // :142
    sto_.integrateSolution(time_discretization_, primal_step_size, dual_step_size, d_);         // :143
    host_solution_valid_ = false;
  }

  // solve (ocp_solver.cpp:148-225): Newton iterations until KKTError() < kkt_tol or max_iter
  void solve(const double t, const Vec& q, const Vec& v, const bool init_solver = true) {
    if (q.size() != s_.at(0).q.size()) throw std::out_of_range("[OCPSolver] invalid argument: q.size() must be " + std::to_string(s_[0].q.size()) + "!");
    if (v.size() != ocp_.robot.dimv) throw std::out_of_range("[OCPSolver] invalid argument: v.size() must be " + std::to_string(ocp_.robot.dimv) + "!");
    const auto t0 = std::chrono::high_resolution_clock::now();
    if (init_solver) {
      discretize(t);
      initConstraints();
      chk(rtoc_line_search_clear(dev_->get()), "rtoc_line_search_clear");                       // :166
    }
    solver_statistics_.clear();
    // the iteration schedule (ocp_solver.cpp:169-213) is rtoc_solve_loop's (include/rtoc_robot.h) -- the function the Python shell
    // (robotoc_amd/solver.py) runs too; what an iteration and a refinement ARE is this class's (the callbacks below)
    struct Frame {
      OCPSolver* self;
      double t;
      const Vec* q;
      const Vec* v;
      std::exception_ptr error;
    } frame{this, t, &q, &v, nullptr};
    rtoc_solve_callbacks cb;
    cb.user = &frame;
    cb.set_sto_regularization = [](void* u, double sto_reg) -> int {                            // :169-177
      Frame& f = *static_cast<Frame*>(u);
      try {
        f.self->sto_.setRegularization(sto_reg);
        f.self->solver_statistics_.ts.push_back(f.self->ocp_.source->contactSequence()->eventTimes());
      } catch (...) { f.error = std::current_exception(); return -100; }
      return 0;
    };
    cb.update_solution = [](void* u, double* kkt_error) -> int {                                // :178-180
      Frame& f = *static_cast<Frame*>(u);
      try {
        f.self->updateSolution(f.t, *f.q, *f.v);
        *kkt_error = f.self->KKTError();
        f.self->solver_statistics_.performance_index.push_back(*kkt_error * *kkt_error);  // PerformanceIndex::kkt_error is the squared residual
      } catch (...) { f.error = std::current_exception(); return -100; }
      return 0;
    };
    cb.max_time_step = [](void* u, double* max_dt) -> int {
      Frame& f = *static_cast<Frame*>(u);
      try { *max_dt = f.self->maxTimeStep(); } catch (...) { f.error = std::current_exception(); return -100; }
      return 0;
    };
    cb.mesh_refinement = [](void* u) -> int {                                                   // :184-196
      Frame& f = *static_cast<Frame*>(u);
      try { f.self->meshRefinement(f.t); } catch (...) { f.error = std::current_exception(); return -100; }
      return 0;
    };
    rtoc_solve_options so;
    so.max_iter = solver_options_.max_iter, so.kkt_tol = solver_options_.kkt_tol, so.sto_enabled = sto_.enabled() ? 1 : 0;
    so.initial_sto_reg_iter = solver_options_.initial_sto_reg_iter, so.initial_sto_reg = solver_options_.initial_sto_reg;
    so.kkt_tol_mesh = solver_options_.kkt_tol_mesh, so.max_dt_mesh = solver_options_.max_dt_mesh;
    rtoc_solve_stats ss;
    const int rc = rtoc_solve_loop(&so, &cb, &ss);
    if (frame.error) std::rethrow_exception(frame.error);
    chk(rc, "rtoc_solve_loop");
    solver_statistics_.convergence = ss.convergence != 0;
    solver_statistics_.iter = ss.iter;
    for (int k = 0; k < ss.num_mesh_refinements && k < RTOC_SOLVE_MAX_REFINEMENTS; ++k) solver_statistics_.mesh_refinement_iter.push_back(ss.mesh_refinement_iter[k]);
    if (solver_options_.enable_benchmark)
      solver_statistics_.cpu_time = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
  }

  const SolverStatistics& getSolverStatistics() const { return solver_statistics_; }
  const Solution& getSolution() {
    syncSolution();
    return s_;
  }
  const SplitSolution& getSolution(const int stage) {
    syncSolution();
    return s_.at(stage);
  }
  const std::vector<LQRPolicy>& getLQRPolicy() {
    riccati_recursion_.downloadFactorization(time_discretization_, riccati_factorization_);
    return riccati_recursion_.getLQRPolicy();
  }
  const RiccatiFactorization& getRiccatiFactorization() {
    riccati_recursion_.downloadFactorization(time_discretization_, riccati_factorization_);
    return riccati_factorization_;
  }
  const Direction& getDirection() {  // not in the reference's public surface; the MPC gather exchanges it
    riccati_recursion_.downloadDirection(time_discretization_, d_);
    return d_;
  }
  void setSolution(const Solution& s) {
    if (s.size() != s_.size()) throw std::out_of_range("[OCPSolver] invalid argument: s.size() must be " + std::to_string(s_.size()) + "!");
    s_ = s;
    uploadSolution();
  }
  // KKTError(t, q, v) (ocp_solver.cpp:414-426): linearise at the current iterate and evaluate
  double KKTError(const double t, const Vec& q, const Vec& v) {
    if (time_discretization_.size() < 2) discretize(t);
    dms_.evalOCP(time_discretization_, q, v, s_, kkt_residual_);
    if (sto_.enabled()) {   // KKTError(t, q, v) evaluates both halves (:422-424); the STO scatter reads the condensed h
      chk(rtoc_condense(dev_->get()), "rtoc_condense");
      sto_.evalKKT(time_discretization_, kkt_matrix_, kkt_residual_);
    }
    return KKTError();
  }
  // KKTError() (ocp_solver.cpp:429-431): sqrt of the accumulated squared residual, STO term included
  double KKTError() const { return std::sqrt(dms_.getEval().kkt_error + sto_.getEval().kkt_error); }
  const std::vector<double>& eventTimes() const {
    static const std::vector<double> none;
    ContactSequence* cs = ocp_.source->contactSequence();
    return cs ? cs->eventTimes() : none;
  }
  const TimeDiscretization& getTimeDiscretization() const { return time_discretization_; }
  unsigned status() const { return riccati_recursion_.status(); }
  rtoc_ctx* context() const { return dev_ ? dev_->get() : nullptr; }

  void disp(std::ostream& os) const {
    os << "OCPSolver (MI355X): dimv " << ocp_.robot.dimv << ", dimu " << ocp_.robot.dimu << ", grid points "
       << time_discretization_.size() << ", iterations " << solver_statistics_.iter << "\n";
  }
  friend std::ostream& operator<<(std::ostream& os, const OCPSolver& s) {
    s.disp(os);
    return os;
  }

 private:
  static void chk(int rc, const char* what) {
    if (rc != RTOC_OK) throw std::runtime_error(std::string("[OCPSolver] ") + what + ": " + rtoc_error_string(rc));
  }
  // TimeDiscretization::maxTimeStep of the time steps the device holds for this instance (the event times moved since discretize)
  double maxTimeStep() {
    std::vector<double> dt(time_discretization_.size());
    chk(rtoc_sto_get_time_steps(dev_->get(), dt.data(), 1), "rtoc_sto_get_time_steps");
    double m = 0.0;
    for (size_t i = 0; i + 1 < dt.size(); ++i) m = std::max(m, dt[i]);
    return m;
  }
  // the mesh-refinement branch of solve (ocp_solver.cpp:184-199): store the solution over the current grid, re-discretise at the
  // current switching times, interpolate, re-initialise the constraints, clear the line-search filter
  void meshRefinement(const double t) {
    const std::vector<unsigned>* masks = ocp_.source->contactMasks();
    const bool interp = solver_options_.enable_solution_interpolation && masks != nullptr;
    if (interp) {
      // correctTimeSteps + store (:186-189): the grid times that belong to the current event times
      // (the device's time steps date from the last evalKKT, BEFORE integrateSolution moved the event times: correct them first,
      // as the reference does, or the stored impact / lift times never match the new grid's and the event solutions are rebuilt
      // by the initEventSolution blend instead of being copied)
      chk(rtoc_sto_correct_time_steps(dev_->get()), "rtoc_sto_correct_time_steps");
      std::vector<double> dt(time_discretization_.size());
      chk(rtoc_sto_get_time_steps(dev_->get(), dt.data(), 1), "rtoc_sto_get_time_steps");
      double tt = t_;
      for (int i = 0; i < time_discretization_.size(); ++i) {
        time_discretization_.grids()[i].t = tt, time_discretization_.grids()[i].dt = dt[i];
        tt += dt[i];
      }
      std::vector<double> sol(static_cast<size_t>(time_discretization_.size()) * L_.sol.stride);
      chk(rtoc_download(dev_->get(), RTOC_BUF_SOL, 0, sol.data(), sol.size()), "rtoc_download(RTOC_BUF_SOL)");
      solution_interpolator_.store(time_discretization_, *masks, sol);
    }
    discretize(t);                                                                              // :190
    if (interp) {
      std::vector<double> sol(static_cast<size_t>(time_discretization_.size()) * L_.sol.stride, 0.0);
      const ContactSequence* cs = ocp_.source->contactSequence();
      std::vector<int> contact_rows(cs->numContacts());
      for (int c = 0; c < cs->numContacts(); ++c) contact_rows[c] = cs->contactRows(c);
      solution_interpolator_.interpolate(L_, cs->numContacts(), ocp_.robot.dim_passive > 0, time_discretization_,
                                         *ocp_.source->contactMasks(), sol, contact_rows);
      chk(rtoc_upload(dev_->get(), RTOC_BUF_SOL, 0, sol.data(), sol.size()), "rtoc_upload(RTOC_BUF_SOL)");
      host_solution_valid_ = false;
    }
    initConstraints();                                                                          // :194-195
    chk(rtoc_line_search_clear(dev_->get()), "rtoc_line_search_clear");                         // :196
  }
  bool hasSE3() const {
    const StageDumpSource* d = dynamic_cast<const StageDumpSource*>(ocp_.source.get());
    return d && !d->buffer(RTOC_BUF_SE3).empty();
  }
  void uploadSolution() {
    std::vector<double> b(static_cast<size_t>(s_.size()) * L_.sol.stride, 0.0);
    for (size_t i = 0; i < s_.size(); ++i) StageDumpSource::packSolution(L_, s_[i], &b[i * L_.sol.stride]);
    if (rtoc_upload(dev_->get(), RTOC_BUF_SOL, 0, b.data(), b.size()) != RTOC_OK)
      throw std::runtime_error("[OCPSolver] rtoc_upload(RTOC_BUF_SOL)");
    host_solution_valid_ = true;
  }
  void syncSolution() {
    if (host_solution_valid_) return;
    std::vector<double> b(static_cast<size_t>(s_.size()) * L_.sol.stride);
    if (rtoc_download(dev_->get(), RTOC_BUF_SOL, 0, b.data(), b.size()) != RTOC_OK)
      throw std::runtime_error("[OCPSolver] rtoc_download(RTOC_BUF_SOL)");
    for (size_t i = 0; i < s_.size(); ++i) StageDumpSource::unpackSolution(L_, &b[i * L_.sol.stride], s_[i]);
    host_solution_valid_ = true;
  }

  SolverOCP ocp_;
  std::shared_ptr<DeviceContext> dev_;
  TimeDiscretization time_discretization_;
  DirectMultipleShooting dms_;
  SwitchingTimeOptimization sto_;
  RiccatiRecursion riccati_recursion_;
  SolutionInterpolator solution_interpolator_;
  double t_ = 0.0;
  KKTMatrix kkt_matrix_;
  KKTResidual kkt_residual_;
  Solution s_;
  Direction d_;
  RiccatiFactorization riccati_factorization_;
  SolverOptions solver_options_;
  SolverStatistics solver_statistics_;
  rtoc_layout L_;
  bool host_solution_valid_ = true;
};

}  // namespace robotoc

#endif  // ROBOTOC_HIP_SOLVER_HPP_

// robotoc_hip.hpp -- C++ host mirror of the robotoc classes that sit directly on the hot path,
// backed by the C ABI (include/rtoc.h).  Header-only, C++11, no Eigen (it is absent from this
// image): the containers expose the reference's member names over a minimal column-major
// `Mat`/`Vec` value type with Eigen-like `(i,j)` access, so that a robotoc maintainer can see
// one-to-one what is packed where.  With Eigen available the same pack/unpack code works on
// `Eigen::MatrixXd::data()` (column-major, identical element order).
//
// Mirrors (reference paths):
//   robotoc::GridInfo / GridType            include/robotoc/ocp/grid_info.hpp:13-93
//   robotoc::TimeDiscretization (size/[])   include/robotoc/ocp/time_discretization.hpp
//   robotoc::SplitKKTMatrix / Residual      include/robotoc/core/split_kkt_matrix.hpp:18, split_kkt_residual.hpp
//   robotoc::SplitRiccatiFactorization      include/robotoc/riccati/split_riccati_factorization.hpp:15
//   robotoc::LQRPolicy                      include/robotoc/riccati/lqr_policy.hpp:16
//   robotoc::SplitDirection                 include/robotoc/core/split_direction.hpp
//   robotoc::RiccatiRecursion               include/robotoc/riccati/riccati_recursion.hpp:26-117
//       RiccatiRecursion(ocp, max_dts0) / setRegularization / backwardRiccatiRecursion /
//       forwardRiccatiRecursion / getLQRPolicy / resizeData -- same names, argument order and
//       in-place semantics (backward mutates kkt_matrix.{Qxx,Qxu,Quu}, kkt_residual.lu).
// Error behaviour: argument misuse throws std::invalid_argument / std::out_of_range like the
// reference's solver layer (src/solver/ocp_solver.cpp:29-49,150-155); numerical failure, which the
// reference only asserts in Debug builds, is reported by RiccatiRecursion::status().
#ifndef ROBOTOC_HIP_HPP_
#define ROBOTOC_HIP_HPP_

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtoc.h"

namespace robotoc {

enum class GridType { Intermediate = 0, Impact = 1, Lift = 2, Terminal = 3 };

struct GridInfo {
  GridType type = GridType::Intermediate;
  double t0 = 0, t = 0, dt = 0, dt_next = 0;
  int phase = 0, stage = 0, impact_index = -1, lift_index = -1, stage_in_phase = 0;
  int num_grids_in_phase = 0;
  bool sto = false, sto_next = false, switching_constraint = false;
  // the two per-stage dimensions the reference reads from ContactSequence
  int dimf = 0;  // contactStatus(phase).dimf()  (impactStatus(impact_index).dimf() on Impact grids)
  int dims = 0;  // impactStatus(impact_index+1).dimf() when switching_constraint
};

class TimeDiscretization {
 public:
  TimeDiscretization() {}
  explicit TimeDiscretization(const std::vector<GridInfo>& grid) : grid_(grid) {}
  int size() const { return static_cast<int>(grid_.size()); }
  const GridInfo& operator[](int i) const { return grid_.at(i); }
  const GridInfo& grid(int i) const { return grid_.at(i); }
  std::vector<GridInfo>& grids() { return grid_; }

 private:
  std::vector<GridInfo> grid_;
};

class Vec {
 public:
  Vec() {}
  explicit Vec(int n) : d_(n, 0.0) {}
  int size() const { return static_cast<int>(d_.size()); }
  double& operator()(int i) { return d_[i]; }
  double operator()(int i) const { return d_[i]; }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
  void setZero() { std::fill(d_.begin(), d_.end(), 0.0); }

 private:
  std::vector<double> d_;
};

class Mat {  // column-major like Eigen::MatrixXd
 public:
  Mat() : r_(0), c_(0) {}
  Mat(int r, int c) : r_(r), c_(c), d_(static_cast<size_t>(r) * c, 0.0) {}
  int rows() const { return r_; }
  int cols() const { return c_; }
  double& operator()(int i, int j) { return d_[i + static_cast<size_t>(j) * r_]; }
  double operator()(int i, int j) const { return d_[i + static_cast<size_t>(j) * r_]; }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
  void setZero() { std::fill(d_.begin(), d_.end(), 0.0); }

 private:
  int r_, c_;
  std::vector<double> d_;
};

struct RobotDims {  // what the hot path reads from robotoc::Robot
  int dimv, dimu, dim_passive, max_dimf;
  rtoc_dims c() const { return rtoc_dims{dimv, dimu, dim_passive, max_dimf, max_dimf, 0}; }
};

struct OCP {  // robotoc::OCP (include/robotoc/ocp/ocp.hpp:22-147): the members the Riccati classes use
  RobotDims robot;
  int N = 0;
  int reserved_num_discrete_events = 0;
  double T = 0.0;  // horizon length (UnconstrRiccatiRecursion: dt = T / N)
};

class SplitKKTMatrix {
 public:
  SplitKKTMatrix() {}
  explicit SplitKKTMatrix(const RobotDims& r)
      : Fxx(2 * r.dimv, 2 * r.dimv), Fvu(r.dimv, r.dimu), Qxx(2 * r.dimv, 2 * r.dimv),
        Qxu(2 * r.dimv, r.dimu), Quu(r.dimu, r.dimu), fx(2 * r.dimv), hx(2 * r.dimv), hu(r.dimu),
        Phix_full(r.max_dimf, 2 * r.dimv), Phiu_full(r.max_dimf, r.dimu), Phit_full(r.max_dimf),
        Qaa(r.dimv, r.dimv), Qdvdv(r.dimv, r.dimv), Qff_full(r.max_dimf, r.max_dimf),
        Qqf_full(r.dimv, r.max_dimf), ha(r.dimv), hf_full(r.max_dimf) {}
  Mat Fxx, Fvu, Qxx, Qxu, Quu;
  Vec fx, hx, hu;
  double Qtt = 0, Qtt_prev = 0;
  Mat Phix_full, Phiu_full;  // max-size backing; active rows = dims()
  Vec Phit_full;
  // un-condensed acceleration / contact-force blocks (split_kkt_matrix.hpp Qaa, Qff(), Qqf(), ha, hf()):
  // inputs of condenseContactDynamics / condenseImpactDynamics (robotoc_hip_dynamics.hpp); only
  // Qaa.diagonal() (Qdvdv.diagonal() on impact stages) is read
  Mat Qaa, Qdvdv, Qff_full, Qqf_full;
  Vec ha, hf_full;
  void setSwitchingConstraintDimension(int dims) { dims_ = dims; }
  int dims() const { return dims_; }

 private:
  int dims_ = 0;
};

class SplitKKTResidual {
 public:
  SplitKKTResidual() {}
  explicit SplitKKTResidual(const RobotDims& r)
      : Fx(2 * r.dimv), lx(2 * r.dimv), lu(r.dimu), P_full(r.max_dimf), la(r.dimv), ldv(r.dimv),
        lf_full(r.max_dimf) {}
  Vec Fx, lx, lu, P_full;
  Vec la, ldv, lf_full;  // split_kkt_residual.hpp la, ldv, lf(): inputs of condenseContact/ImpactDynamics
  double h = 0;
};

class SplitRiccatiFactorization {
 public:
  SplitRiccatiFactorization() {}
  explicit SplitRiccatiFactorization(const RobotDims& r)
      : P(2 * r.dimv, 2 * r.dimv), s(2 * r.dimv), psi_x(2 * r.dimv), psi_u(r.dimu), Psi(2 * r.dimv),
        phi_x(2 * r.dimv), phi_u(r.dimu), Phi(2 * r.dimv), M_full(r.max_dimf, 2 * r.dimv),
        m_full(r.max_dimf), mt_full(r.max_dimf), mt_next_full(r.max_dimf) {}
  Mat P;
  Vec s, psi_x, psi_u, Psi, phi_x, phi_u, Phi;
  double xi = 0, chi = 0, rho = 0, eta = 0, iota = 0;
  Mat M_full;
  Vec m_full, mt_full, mt_next_full;
};

class LQRPolicy {
 public:
  LQRPolicy() {}
  explicit LQRPolicy(const RobotDims& r) : Kt(2 * r.dimv, r.dimu), k(r.dimu), T(r.dimu), W(r.dimu) {}
  // K is row-major dimu x dimx in the reference (lqr_policy.hpp:18-19); stored here as its
  // column-major transpose Kt (identical memory), K(i,j) == Kt(j,i).
  Mat Kt;
  double K(int i, int j) const { return Kt(j, i); }
  Vec k, T, W;
};

class SplitDirection {
 public:
  SplitDirection() {}
  explicit SplitDirection(const RobotDims& r)
      : dx(2 * r.dimv), du(r.dimu), dlmdgmm(2 * r.dimv), dxi_full(r.max_dimf), daf_full(r.dimv + r.max_dimf),
        dbetamu_full(r.dimv + r.max_dimf), dnu_passive(r.dim_passive) {}
  Vec dx, du, dlmdgmm, dxi_full;
  Vec daf_full, dbetamu_full, dnu_passive;  // split_direction.hpp daf(), dbetamu(), dnu_passive (contact-dynamics expansion)
  double dts = 0, dts_next = 0;
};

typedef std::vector<SplitKKTMatrix> KKTMatrix;
typedef std::vector<SplitKKTResidual> KKTResidual;
typedef std::vector<SplitRiccatiFactorization> RiccatiFactorization;
typedef std::vector<SplitDirection> Direction;

class RiccatiRecursion {
 public:
  RiccatiRecursion(const OCP& ocp, const double max_dts0 = 0.1, const int device = 0)
      : robot_(ocp.robot), max_stages_(ocp.N + 1 + 3 * ocp.reserved_num_discrete_events + 1),
        lqr_policy_(ocp.N + 1 + ocp.reserved_num_discrete_events, LQRPolicy(ocp.robot)), ctx_(nullptr), owns_(true),
        resident_hash_(0) {
    if (max_dts0 <= 0) throw std::out_of_range("[RiccatiRecursion] invalid argument: max_dts0 must be positive!");
    const rtoc_dims d = robot_.c();
    check(rtoc_create(&d, max_stages_, 1, device, &ctx_), "rtoc_create");
    try {
      check(rtoc_get_layout(ctx_, &L_), "rtoc_get_layout");
      check(rtoc_set_option(ctx_, RTOC_OPT_WRITEBACK_KKT, 1), "rtoc_set_option");  // reference in-place semantics
      setRegularization(max_dts0);
    } catch (...) {
      rtoc_destroy(ctx_);
      ctx_ = nullptr;
      throw;
    }
  }
  // Over a context someone else owns (robotoc::OCPSolver shares ONE device context between its
  // DirectMultipleShooting and RiccatiRecursion members so that the stage data stay resident in HBM).
  RiccatiRecursion(const OCP& ocp, rtoc_ctx* shared)
      : robot_(ocp.robot), max_stages_(0), lqr_policy_(ocp.N + 1 + ocp.reserved_num_discrete_events, LQRPolicy(ocp.robot)),
        ctx_(shared), owns_(false), resident_hash_(0) {
    if (!shared) throw std::invalid_argument("[RiccatiRecursion] null device context");
    check(rtoc_get_layout(ctx_, &L_), "rtoc_get_layout");
    max_stages_ = static_cast<int>(rtoc_buffer_count(ctx_, RTOC_BUF_KKT) / L_.kkt.stride);
  }
  // Default constructor like the reference's (riccati_recursion.hpp:40): an empty object, usable after assignment.
  RiccatiRecursion() : robot_{0, 0, 0, 0}, max_stages_(0), ctx_(nullptr), owns_(true), resident_hash_(0) {}
  ~RiccatiRecursion() { release(); }
  // Value semantics like the reference (riccati_recursion.hpp:50-60): a copy owns a deep copy of the device context.
  RiccatiRecursion(const RiccatiRecursion& o)
      : robot_(o.robot_), max_stages_(o.max_stages_), lqr_policy_(o.lqr_policy_), ctx_(nullptr), owns_(true), L_(o.L_),
        resident_hash_(o.resident_hash_) {
    if (o.ctx_) check(rtoc_clone(o.ctx_, &ctx_), "rtoc_clone");
  }
  RiccatiRecursion& operator=(const RiccatiRecursion& o) {
    if (this != &o) {
      RiccatiRecursion tmp(o);
      swap(tmp);
    }
    return *this;
  }
  RiccatiRecursion(RiccatiRecursion&& o) noexcept
      : robot_(o.robot_), max_stages_(o.max_stages_), lqr_policy_(std::move(o.lqr_policy_)), ctx_(o.ctx_), owns_(o.owns_),
        L_(o.L_), resident_hash_(o.resident_hash_) {
    o.ctx_ = nullptr;
  }
  RiccatiRecursion& operator=(RiccatiRecursion&& o) noexcept {
    if (this != &o) {
      release();
      robot_ = o.robot_;
      max_stages_ = o.max_stages_;
      lqr_policy_ = std::move(o.lqr_policy_);
      ctx_ = o.ctx_;
      owns_ = o.owns_;
      L_ = o.L_;
      resident_hash_ = o.resident_hash_;
      o.ctx_ = nullptr;
    }
    return *this;
  }

  void setRegularization(const double max_dts0) {
    if (max_dts0 <= 0) throw std::out_of_range("[RiccatiRecursion] invalid argument: max_dts0 must be positive!");
    int64_t bits;
    std::memcpy(&bits, &max_dts0, sizeof(bits));
    check(rtoc_set_option(ctx_, RTOC_OPT_MAX_DTS0, bits), "rtoc_set_option");
  }

  // Not in the reference: run backwardRiccatiRecursion as a scan over the horizon (RTOC_OPT_BACKWARD_SCAN) --
  // the low-latency path for ONE OCP, which is what this class holds.  Same results to <= 1e-8 relative;
  // with switching-time optimisation the matrix half is the scan, the vector half a serial pass behind it (riccati_scan_sto.hpp).
  void setHorizonScan(const bool on) {
    check(rtoc_set_option(ctx_, RTOC_OPT_BACKWARD_SCAN, on ? 1 : 0), "rtoc_set_option");
  }
  // 0: serial kernels, 1: scans, 2: automatic (scans for batches of at most 8 instances) -- RTOC_OPT_BACKWARD_SCAN as it is
  void setHorizonScanMode(const int mode) { check(rtoc_set_option(ctx_, RTOC_OPT_BACKWARD_SCAN, mode), "rtoc_set_option"); }

  void resizeData(const TimeDiscretization& td) {
    const int N = td.size() - 1;
    while (static_cast<int>(lqr_policy_.size()) < N + 1) lqr_policy_.push_back(lqr_policy_.back());
  }

  void backwardRiccatiRecursion(const TimeDiscretization& td, KKTMatrix& kkt_matrix,
                                KKTResidual& kkt_residual, RiccatiFactorization& factorization) {
    need_ctx();
    resizeData(td);
    const int n = td.size();
    if (n > max_stages_ || static_cast<int>(kkt_matrix.size()) < n || static_cast<int>(kkt_residual.size()) < n ||
        static_cast<int>(factorization.size()) < n)
      throw std::invalid_argument("[RiccatiRecursion] horizon containers smaller than the discretisation");
    setGrid(td);
    // staging buffers live as long as the object: no allocation per call
    kkt_stage_.assign(static_cast<size_t>(n) * L_.kkt.stride, 0.0);
    for (int i = 0; i < n; ++i) packKKT(kkt_matrix[i], kkt_residual[i], &kkt_stage_[static_cast<size_t>(i) * L_.kkt.stride]);
    check(rtoc_upload(ctx_, RTOC_BUF_KKT, 0, kkt_stage_.data(), kkt_stage_.size()), "rtoc_upload");
    check(rtoc_clear_status(ctx_), "rtoc_clear_status");
    check(rtoc_riccati_backward(ctx_), "rtoc_riccati_backward");
    check(rtoc_download(ctx_, RTOC_BUF_KKT, 0, kkt_stage_.data(), kkt_stage_.size()), "rtoc_download");
    for (int i = 0; i < n - 1; ++i)  // mutated blocks, like the reference (brrf.cpp:37-44,82-83)
      unpackMutatedKKT(&kkt_stage_[static_cast<size_t>(i) * L_.kkt.stride], td[i], kkt_matrix[i], kkt_residual[i]);
    ric_stage_.resize(static_cast<size_t>(n) * L_.ric.stride);
    check(rtoc_download(ctx_, RTOC_BUF_RIC, 0, ric_stage_.data(), ric_stage_.size()), "rtoc_download");
    for (int i = 0; i < n; ++i) unpackRiccati(&ric_stage_[static_cast<size_t>(i) * L_.ric.stride], factorization[i], lqr_policy_[i]);
    // fingerprint of what is resident in HBM now (the mutated KKT blocks and the factorisation): the forward
    // recursion compares it with the containers it is handed
    resident_hash_ = hashHorizon(n, kkt_matrix, kkt_residual, factorization);
  }

  // The reference's forward recursion reads the kkt_matrix / kkt_residual / factorization it is handed
  // (riccati_recursion.cpp:83-131).  Normally these are the containers of the preceding backward pass, whose
  // contents are still resident in HBM; if the caller edited them in between (fingerprint mismatch) they are
  // packed and uploaded again, so the arguments are always honoured.
  void forwardRiccatiRecursion(const TimeDiscretization& td, const KKTMatrix& kkt_matrix, const KKTResidual& kkt_residual,
                               const RiccatiFactorization& factorization, Direction& d) {
    need_ctx();
    const int n = td.size();
    if (static_cast<int>(d.size()) < n || static_cast<int>(kkt_matrix.size()) < n || static_cast<int>(kkt_residual.size()) < n ||
        static_cast<int>(factorization.size()) < n)
      throw std::invalid_argument("[RiccatiRecursion] horizon containers smaller than the discretisation");
    if (hashHorizon(n, kkt_matrix, kkt_residual, factorization) != resident_hash_) {
      setGrid(td);
      kkt_stage_.assign(static_cast<size_t>(n) * L_.kkt.stride, 0.0);
      ric_stage_.assign(static_cast<size_t>(n) * L_.ric.stride, 0.0);
      for (int i = 0; i < n; ++i) {
        packKKT(kkt_matrix[i], kkt_residual[i], &kkt_stage_[static_cast<size_t>(i) * L_.kkt.stride]);
        packRiccati(factorization[i], lqr_policy_[i], &ric_stage_[static_cast<size_t>(i) * L_.ric.stride]);
      }
      check(rtoc_upload(ctx_, RTOC_BUF_KKT, 0, kkt_stage_.data(), kkt_stage_.size()), "rtoc_upload");
      // the STO policy entries of the records (dtsdx, dtsdts, dts0) have no host container in the reference
      // (RiccatiRecursion::sto_policy_ is private): keep the resident ones
      std::vector<double> cur(ric_stage_.size());
      check(rtoc_download(ctx_, RTOC_BUF_RIC, 0, cur.data(), cur.size()), "rtoc_download");
      for (int i = 0; i < n; ++i) {
        double* r = &ric_stage_[static_cast<size_t>(i) * L_.ric.stride];
        const double* c0 = &cur[static_cast<size_t>(i) * L_.ric.stride];
        cp(r + L_.ric.off[RTOC_RIC_DTSDX], c0 + L_.ric.off[RTOC_RIC_DTSDX], 2 * robot_.dimv);
        r[L_.ric.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTSDTS] = c0[L_.ric.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTSDTS];
        r[L_.ric.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0] = c0[L_.ric.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0];
      }
      check(rtoc_upload(ctx_, RTOC_BUF_RIC, 0, ric_stage_.data(), ric_stage_.size()), "rtoc_upload");
      resident_hash_ = hashHorizon(n, kkt_matrix, kkt_residual, factorization);
    }
    check(rtoc_upload(ctx_, RTOC_BUF_DX0, 0, d[0].dx.data(), 2 * robot_.dimv), "rtoc_upload");
    check(rtoc_riccati_forward(ctx_), "rtoc_riccati_forward");
    dir_stage_.resize(static_cast<size_t>(n) * L_.dir.stride);
    check(rtoc_download(ctx_, RTOC_BUF_DIR, 0, dir_stage_.data(), dir_stage_.size()), "rtoc_download");
    for (int i = 0; i < n; ++i) unpackDirection(&dir_stage_[static_cast<size_t>(i) * L_.dir.stride], d[i]);
  }

  // Device-resident forms used by robotoc::OCPSolver (robotoc_hip_solver.hpp): the stage data are already in HBM
  // (RTOC_BUF_KKT after rtoc_condense, RTOC_BUF_DX0), nothing crosses PCIe.
  void backwardRiccatiRecursionResident(const TimeDiscretization& td) {
    need_ctx();
    resizeData(td);
    check(rtoc_riccati_backward(ctx_), "rtoc_riccati_backward");
  }
  void forwardRiccatiRecursionResident() {
    need_ctx();
    check(rtoc_riccati_forward(ctx_), "rtoc_riccati_forward");
  }
  // factorisation + LQR policies / directions of the resident horizon into host containers (on demand)
  void downloadFactorization(const TimeDiscretization& td, RiccatiFactorization& factorization) {
    need_ctx();
    const int n = td.size();
    resizeData(td);
    if (static_cast<int>(factorization.size()) < n) throw std::invalid_argument("[RiccatiRecursion] factorization too short");
    ric_stage_.resize(static_cast<size_t>(n) * L_.ric.stride);
    check(rtoc_download(ctx_, RTOC_BUF_RIC, 0, ric_stage_.data(), ric_stage_.size()), "rtoc_download");
    for (int i = 0; i < n; ++i) unpackRiccati(&ric_stage_[static_cast<size_t>(i) * L_.ric.stride], factorization[i], lqr_policy_[i]);
  }
  void downloadDirection(const TimeDiscretization& td, Direction& d) {
    need_ctx();
    const int n = td.size();
    if (static_cast<int>(d.size()) < n) throw std::invalid_argument("[RiccatiRecursion] direction too short");
    dir_stage_.resize(static_cast<size_t>(n) * L_.dir.stride);
    check(rtoc_download(ctx_, RTOC_BUF_DIR, 0, dir_stage_.data(), dir_stage_.size()), "rtoc_download");
    for (int i = 0; i < n; ++i) unpackDirection(&dir_stage_[static_cast<size_t>(i) * L_.dir.stride], d[i]);
  }

  const std::vector<LQRPolicy>& getLQRPolicy() const { return lqr_policy_; }

  // RTOC_STAT_* bits of the last backward pass (the reference asserts in Debug builds only)
  unsigned status() const {
    need_ctx();
    uint32_t s = 0;
    check(rtoc_status(ctx_, &s, 1), "rtoc_status");
    return s;
  }
  rtoc_ctx* context() const { return ctx_; }

  static void setGridOf(rtoc_ctx* ctx, const TimeDiscretization& td) {
    std::vector<rtoc_grid> g(td.size());
    for (int i = 0; i < td.size(); ++i) {
      const GridInfo& gi = td[i];
      g[i].type = static_cast<int>(gi.type);
      g[i].sto = gi.sto;
      g[i].sto_next = gi.sto_next;
      g[i].switching_constraint = gi.switching_constraint;
      g[i].dimf = gi.dimf;
      g[i].dims = gi.switching_constraint ? gi.dims : 0;
      g[i].num_grids_in_phase = gi.num_grids_in_phase;
      g[i].time_stage = gi.type == GridType::Impact ? -1 : gi.stage;
      g[i].dt = gi.dt;
    }
    check(rtoc_set_grid(ctx, g.data(), td.size()), "rtoc_set_grid");
  }

 private:
  static void check(int rc, const char* what) {
    if (rc != RTOC_OK) throw std::runtime_error(std::string("[RiccatiRecursion] ") + what + ": " + rtoc_error_string(rc));
  }
  void need_ctx() const {
    if (!ctx_) throw std::logic_error("[RiccatiRecursion] default-constructed object: assign a constructed one first");
  }
  void release() {
    if (ctx_ && owns_) rtoc_destroy(ctx_);
    ctx_ = nullptr;
  }
  void swap(RiccatiRecursion& o) {
    std::swap(robot_, o.robot_);
    std::swap(max_stages_, o.max_stages_);
    lqr_policy_.swap(o.lqr_policy_);
    std::swap(ctx_, o.ctx_);
    std::swap(owns_, o.owns_);
    std::swap(L_, o.L_);
    std::swap(resident_hash_, o.resident_hash_);
  }
  // FNV-1a over the bytes the forward recursion reads
  static void fnv(unsigned long long& h, const double* p, size_t n) {
    const unsigned char* b = reinterpret_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n * sizeof(double); ++i) {
      h ^= b[i];
      h *= 1099511628211ull;
    }
  }
  unsigned long long hashHorizon(int n, const KKTMatrix& m, const KKTResidual& r, const RiccatiFactorization& f) const {
    unsigned long long h = 1469598103934665603ull;
    const size_t nx = 2 * static_cast<size_t>(robot_.dimv);
    for (int i = 0; i < n; ++i) {
      fnv(h, m[i].Fxx.data(), nx * nx);
      fnv(h, m[i].Fvu.data(), static_cast<size_t>(robot_.dimv) * robot_.dimu);
      fnv(h, m[i].fx.data(), nx);
      fnv(h, r[i].Fx.data(), nx);
      fnv(h, f[i].P.data(), nx * nx);
      fnv(h, f[i].s.data(), nx);
      fnv(h, f[i].Psi.data(), nx);
      fnv(h, f[i].Phi.data(), nx);
      if (robot_.max_dimf > 0) {
        fnv(h, f[i].M_full.data(), static_cast<size_t>(robot_.max_dimf) * nx);
        fnv(h, f[i].m_full.data(), robot_.max_dimf);
      }
      if (i < static_cast<int>(lqr_policy_.size())) {
        fnv(h, lqr_policy_[i].Kt.data(), nx * robot_.dimu);
        fnv(h, lqr_policy_[i].k.data(), robot_.dimu);
      }
    }
    return h;
  }
  void setGrid(const TimeDiscretization& td) { setGridOf(ctx_, td); }
  static void cp(double* dst, const double* src, size_t n) { std::memcpy(dst, src, n * sizeof(double)); }
  void packKKT(const SplitKKTMatrix& m, const SplitKKTResidual& r, double* rec) const {
    const int nv = robot_.dimv, nu = robot_.dimu, nx = 2 * nv, ns = robot_.max_dimf;
    const int* o = L_.kkt.off;
    cp(rec + o[RTOC_KKT_FXX], m.Fxx.data(), static_cast<size_t>(nx) * nx);
    cp(rec + o[RTOC_KKT_FVU], m.Fvu.data(), static_cast<size_t>(nv) * nu);
    cp(rec + o[RTOC_KKT_QXX], m.Qxx.data(), static_cast<size_t>(nx) * nx);
    cp(rec + o[RTOC_KKT_QXU], m.Qxu.data(), static_cast<size_t>(nx) * nu);
    cp(rec + o[RTOC_KKT_QUU], m.Quu.data(), static_cast<size_t>(nu) * nu);
    cp(rec + o[RTOC_KKT_FX], r.Fx.data(), nx);
    cp(rec + o[RTOC_KKT_LX], r.lx.data(), nx);
    cp(rec + o[RTOC_KKT_LU], r.lu.data(), nu);
    cp(rec + o[RTOC_KKT_FFX], m.fx.data(), nx);
    cp(rec + o[RTOC_KKT_HX], m.hx.data(), nx);
    cp(rec + o[RTOC_KKT_HU], m.hu.data(), nu);
    rec[o[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_QTT] = m.Qtt;
    rec[o[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_QTT_PREV] = m.Qtt_prev;
    rec[o[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_H] = r.h;
    if (ns > 0) {
      cp(rec + o[RTOC_KKT_PHIX], m.Phix_full.data(), static_cast<size_t>(ns) * nx);
      cp(rec + o[RTOC_KKT_PHIU], m.Phiu_full.data(), static_cast<size_t>(ns) * nu);
      cp(rec + o[RTOC_KKT_PHIT], m.Phit_full.data(), ns);
      cp(rec + o[RTOC_KKT_PRES], r.P_full.data(), ns);
    }
  }
  void unpackMutatedKKT(const double* rec, const GridInfo& g, SplitKKTMatrix& m, SplitKKTResidual& r) const {
    const int nv = robot_.dimv, nu = robot_.dimu, nx = 2 * nv;
    const int* o = L_.kkt.off;
    cp(m.Qxx.data(), rec + o[RTOC_KKT_QXX], static_cast<size_t>(nx) * nx);
    if (g.type != GridType::Impact) {
      cp(m.Qxu.data(), rec + o[RTOC_KKT_QXU], static_cast<size_t>(nx) * nu);
      cp(m.Quu.data(), rec + o[RTOC_KKT_QUU], static_cast<size_t>(nu) * nu);
      cp(r.lu.data(), rec + o[RTOC_KKT_LU], nu);
    }
  }
  void unpackRiccati(const double* rec, SplitRiccatiFactorization& f, LQRPolicy& p) const {
    const int nv = robot_.dimv, nu = robot_.dimu, nx = 2 * nv, ns = robot_.max_dimf;
    const int* o = L_.ric.off;
    cp(f.P.data(), rec + o[RTOC_RIC_P], static_cast<size_t>(nx) * nx);
    cp(f.s.data(), rec + o[RTOC_RIC_S], nx);
    cp(f.Psi.data(), rec + o[RTOC_RIC_PSI], nx);
    cp(f.Phi.data(), rec + o[RTOC_RIC_PHI], nx);
    cp(f.psi_x.data(), rec + o[RTOC_RIC_PSIX], nx);
    cp(f.phi_x.data(), rec + o[RTOC_RIC_PHIX], nx);
    cp(f.psi_u.data(), rec + o[RTOC_RIC_PSIU], nu);
    cp(f.phi_u.data(), rec + o[RTOC_RIC_PHIU], nu);
    const double* sc = rec + o[RTOC_RIC_SCAL];
    f.xi = sc[RTOC_RIC_SCAL_XI];
    f.chi = sc[RTOC_RIC_SCAL_CHI];
    f.rho = sc[RTOC_RIC_SCAL_RHO];
    f.eta = sc[RTOC_RIC_SCAL_ETA];
    f.iota = sc[RTOC_RIC_SCAL_IOTA];
    if (ns > 0) {
      cp(f.M_full.data(), rec + o[RTOC_RIC_M], static_cast<size_t>(ns) * nx);
      cp(f.m_full.data(), rec + o[RTOC_RIC_MV], ns);
      cp(f.mt_full.data(), rec + o[RTOC_RIC_MT], ns);
      cp(f.mt_next_full.data(), rec + o[RTOC_RIC_MTN], ns);
    }
    cp(p.Kt.data(), rec + o[RTOC_RIC_K], static_cast<size_t>(nx) * nu);
    cp(p.k.data(), rec + o[RTOC_RIC_KV], nu);
    cp(p.T.data(), rec + o[RTOC_RIC_T], nu);
    cp(p.W.data(), rec + o[RTOC_RIC_W], nu);
  }

  void packRiccati(const SplitRiccatiFactorization& f, const LQRPolicy& p, double* rec) const {
    const int nv = robot_.dimv, nu = robot_.dimu, nx = 2 * nv, ns = robot_.max_dimf;
    const int* o = L_.ric.off;
    cp(rec + o[RTOC_RIC_P], f.P.data(), static_cast<size_t>(nx) * nx);
    cp(rec + o[RTOC_RIC_S], f.s.data(), nx);
    cp(rec + o[RTOC_RIC_PSI], f.Psi.data(), nx);
    cp(rec + o[RTOC_RIC_PHI], f.Phi.data(), nx);
    cp(rec + o[RTOC_RIC_PSIX], f.psi_x.data(), nx);
    cp(rec + o[RTOC_RIC_PHIX], f.phi_x.data(), nx);
    cp(rec + o[RTOC_RIC_PSIU], f.psi_u.data(), nu);
    cp(rec + o[RTOC_RIC_PHIU], f.phi_u.data(), nu);
    double* sc = rec + o[RTOC_RIC_SCAL];
    sc[RTOC_RIC_SCAL_XI] = f.xi;
    sc[RTOC_RIC_SCAL_CHI] = f.chi;
    sc[RTOC_RIC_SCAL_RHO] = f.rho;
    sc[RTOC_RIC_SCAL_ETA] = f.eta;
    sc[RTOC_RIC_SCAL_IOTA] = f.iota;
    if (ns > 0) {
      cp(rec + o[RTOC_RIC_M], f.M_full.data(), static_cast<size_t>(ns) * nx);
      cp(rec + o[RTOC_RIC_MV], f.m_full.data(), ns);
      cp(rec + o[RTOC_RIC_MT], f.mt_full.data(), ns);
      cp(rec + o[RTOC_RIC_MTN], f.mt_next_full.data(), ns);
    }
    cp(rec + o[RTOC_RIC_K], p.Kt.data(), static_cast<size_t>(nx) * nu);
    cp(rec + o[RTOC_RIC_KV], p.k.data(), nu);
    cp(rec + o[RTOC_RIC_T], p.T.data(), nu);
    cp(rec + o[RTOC_RIC_W], p.W.data(), nu);
  }
  void unpackDirection(const double* r, SplitDirection& d) const {
    cp(d.dx.data(), r + L_.dir.off[RTOC_DIR_DX], 2 * robot_.dimv);
    cp(d.du.data(), r + L_.dir.off[RTOC_DIR_DU], robot_.dimu);
    cp(d.dlmdgmm.data(), r + L_.dir.off[RTOC_DIR_DLMDGMM], 2 * robot_.dimv);
    cp(d.dxi_full.data(), r + L_.dir.off[RTOC_DIR_DXI], robot_.max_dimf);
    cp(d.daf_full.data(), r + L_.dir.off[RTOC_DIR_DAF], robot_.dimv + robot_.max_dimf);
    cp(d.dbetamu_full.data(), r + L_.dir.off[RTOC_DIR_DBETAMU], robot_.dimv + robot_.max_dimf);
    cp(d.dnu_passive.data(), r + L_.dir.off[RTOC_DIR_DNUP], robot_.dim_passive);
    d.dts = r[L_.dir.off[RTOC_DIR_DTS] + 0];
    d.dts_next = r[L_.dir.off[RTOC_DIR_DTS] + 1];
  }

  RobotDims robot_;
  int max_stages_;
  std::vector<LQRPolicy> lqr_policy_;
  rtoc_ctx* ctx_;
  bool owns_;
  rtoc_layout L_;
  unsigned long long resident_hash_;
  std::vector<double> kkt_stage_, ric_stage_, dir_stage_;  // persistent host staging
};

typedef RiccatiFactorization UnconstrRiccatiFactorization;

// robotoc::UnconstrRiccatiRecursion (include/robotoc/riccati/unconstr_riccati_recursion.hpp:38-86,
// src/riccati/unconstr_riccati_recursion.cpp:10-48): fixed-base, contact-free OCP with the acceleration
// as the control.  As in the reference's KKT objects for this solver, kkt_matrix[i].Quu holds Qaa,
// kkt_matrix[i].Qxu holds [Qqa; Qva] and kkt_residual[i].lu holds la
// (unconstr_backward_riccati_recursion_factorizer.cpp:27-70); Fxx / Fvu are not read -- the structured
// A = [[I, dt I],[0, I]], B = [0; dt I] are materialised on the device.
class UnconstrRiccatiRecursion {
 public:
  explicit UnconstrRiccatiRecursion(const OCP& ocp, const int device = 0)
      : robot_(ocp.robot), N_(ocp.N), dt_(ocp.T / ocp.N), lqr_policy_(ocp.N, LQRPolicy(ocp.robot)), ctx_(nullptr) {
    if (ocp.N <= 0 || !(ocp.T > 0)) throw std::out_of_range("[UnconstrRiccatiRecursion] invalid argument: N and T must be positive!");
    if (robot_.dimu != robot_.dimv || robot_.max_dimf != 0)
      throw std::invalid_argument("[UnconstrRiccatiRecursion] robot must be fixed-base without contacts");
    const rtoc_dims d = robot_.c();
    check(rtoc_create(&d, N_ + 1, 1, device, &ctx_), "rtoc_create");
    check(rtoc_get_layout(ctx_, &L_), "rtoc_get_layout");
    std::vector<rtoc_grid> g(N_ + 1);
    for (int i = 0; i <= N_; ++i) {
      g[i] = rtoc_grid{i == N_ ? RTOC_GRID_TERMINAL : RTOC_GRID_INTERMEDIATE, 0, 0, 0, 0, 0, i == N_ ? 0 : N_, i,
                       i == N_ ? 0.0 : dt_};
    }
    check(rtoc_set_grid(ctx_, g.data(), N_ + 1), "rtoc_set_grid");
  }
  ~UnconstrRiccatiRecursion() {
    if (ctx_) rtoc_destroy(ctx_);
  }
  UnconstrRiccatiRecursion(const UnconstrRiccatiRecursion&) = delete;
  UnconstrRiccatiRecursion& operator=(const UnconstrRiccatiRecursion&) = delete;

  // Not in the reference: both recursions as scans over the horizon (RTOC_OPT_BACKWARD_SCAN), see RiccatiRecursion.
  void setHorizonScan(const bool on) {
    check(rtoc_set_option(ctx_, RTOC_OPT_BACKWARD_SCAN, on ? 1 : 0), "rtoc_set_option");
  }

  void backwardRiccatiRecursion(KKTMatrix& kkt_matrix, KKTResidual& kkt_residual,
                                UnconstrRiccatiFactorization& factorization) {
    const int n = N_ + 1, nv = robot_.dimv, nx = 2 * nv;
    if (static_cast<int>(kkt_matrix.size()) < n || static_cast<int>(kkt_residual.size()) < n ||
        static_cast<int>(factorization.size()) < n)
      throw std::invalid_argument("[UnconstrRiccatiRecursion] horizon containers smaller than N + 1");
    const int* o = L_.kkt.off;
    std::vector<double> buf(static_cast<size_t>(n) * L_.kkt.stride, 0.0);
    for (int i = 0; i < n; ++i) {
      double* rec = &buf[static_cast<size_t>(i) * L_.kkt.stride];
      std::memcpy(rec + o[RTOC_KKT_QXX], kkt_matrix[i].Qxx.data(), sizeof(double) * nx * nx);
      std::memcpy(rec + o[RTOC_KKT_LX], kkt_residual[i].lx.data(), sizeof(double) * nx);
      if (i == N_) continue;
      std::memcpy(rec + o[RTOC_KKT_QXU], kkt_matrix[i].Qxu.data(), sizeof(double) * nx * nv);
      std::memcpy(rec + o[RTOC_KKT_QUU], kkt_matrix[i].Quu.data(), sizeof(double) * nv * nv);
      std::memcpy(rec + o[RTOC_KKT_FX], kkt_residual[i].Fx.data(), sizeof(double) * nx);
      std::memcpy(rec + o[RTOC_KKT_LU], kkt_residual[i].lu.data(), sizeof(double) * nv);
    }
    check(rtoc_upload(ctx_, RTOC_BUF_KKT, 0, buf.data(), buf.size()), "rtoc_upload");
    check(rtoc_clear_status(ctx_), "rtoc_clear_status");
    check(rtoc_unconstr_backward(ctx_, dt_), "rtoc_unconstr_backward");
    std::vector<double> rb(static_cast<size_t>(n) * L_.ric.stride);
    check(rtoc_download(ctx_, RTOC_BUF_RIC, 0, rb.data(), rb.size()), "rtoc_download");
    const int* ro = L_.ric.off;
    for (int i = 0; i < n; ++i) {
      const double* rec = &rb[static_cast<size_t>(i) * L_.ric.stride];
      std::memcpy(factorization[i].P.data(), rec + ro[RTOC_RIC_P], sizeof(double) * nx * nx);
      std::memcpy(factorization[i].s.data(), rec + ro[RTOC_RIC_S], sizeof(double) * nx);
      if (i < N_) {
        std::memcpy(lqr_policy_[i].Kt.data(), rec + ro[RTOC_RIC_K], sizeof(double) * nx * nv);
        std::memcpy(lqr_policy_[i].k.data(), rec + ro[RTOC_RIC_KV], sizeof(double) * nv);
      }
    }
  }

  // d[0].dx must hold the initial state direction; on return d[i].du holds the acceleration
  // direction da of stage i (the Riccati control of this solver), d[i].dlmdgmm the costate direction
  void forwardRiccatiRecursion(const KKTResidual&, const UnconstrRiccatiFactorization&, Direction& d) const {
    const int n = N_ + 1, nv = robot_.dimv, nx = 2 * nv;
    if (static_cast<int>(d.size()) < n) throw std::invalid_argument("[UnconstrRiccatiRecursion] direction too short");
    check(rtoc_upload(ctx_, RTOC_BUF_DX0, 0, d[0].dx.data(), nx), "rtoc_upload");
    check(rtoc_unconstr_forward(ctx_, dt_), "rtoc_unconstr_forward");
    std::vector<double> db(static_cast<size_t>(n) * L_.dir.stride);
    check(rtoc_download(ctx_, RTOC_BUF_DIR, 0, db.data(), db.size()), "rtoc_download");
    for (int i = 0; i < n; ++i) {
      const double* r = &db[static_cast<size_t>(i) * L_.dir.stride];
      std::memcpy(d[i].dx.data(), r + L_.dir.off[RTOC_DIR_DX], sizeof(double) * nx);
      std::memcpy(d[i].du.data(), r + L_.dir.off[RTOC_DIR_DU], sizeof(double) * nv);
      std::memcpy(d[i].dlmdgmm.data(), r + L_.dir.off[RTOC_DIR_DLMDGMM], sizeof(double) * nx);
    }
  }

  const std::vector<LQRPolicy>& getLQRPolicy() const { return lqr_policy_; }

  unsigned status() const {
    uint32_t s = 0;
    check(rtoc_status(ctx_, &s, 1), "rtoc_status");
    return s;
  }

 private:
  static void check(int rc, const char* what) {
    if (rc != RTOC_OK)
      throw std::runtime_error(std::string("[UnconstrRiccatiRecursion] ") + what + ": " + rtoc_error_string(rc));
  }
  RobotDims robot_;
  int N_;
  double dt_;
  std::vector<LQRPolicy> lqr_policy_;
  rtoc_ctx* ctx_;
  rtoc_layout L_;
};

}  // namespace robotoc

#endif  // ROBOTOC_HIP_HPP_

// robotoc_hip_dynamics.hpp -- C++ host mirror of the contact-dynamics free functions of the reference
// (include/robotoc/dynamics/contact_dynamics.hpp:23-44), backed by the C ABI (include/rtoc.h):
//
//   condenseContactDynamics(robot, contact_status, dt, data, kkt_matrix, kkt_residual)   src/dynamics/contact_dynamics.cpp:55-164
//   expandContactDynamicsPrimal(data, d)                                                 :167-174
//   expandContactDynamicsDual(dt, dts, data, d_next, d)                                  :177-202
// and of include/robotoc/dynamics/impact_dynamics.hpp:25-36:
//   condenseImpactDynamics(robot, impact_status, data, kkt_matrix, kkt_residual)         src/dynamics/impact_dynamics.cpp:38-80
//   expandImpactDynamicsPrimal(data, d)                                                  :83-88
//   expandImpactDynamicsDual(data, d_next, d)                                            :91-96
//
// Same names, argument order and in-place semantics as the reference: the condensation mutates
// kkt_matrix.{Qxx,Qxu,Quu,Fxx(bottom rows),Fvu,hx,hu}, kkt_residual.{lx,lu,Fx,h} and fills the condensed
// members of `data`; the dual expansion updates data.laf() in place.  One call = one (instance, grid point)
// work item of rtoc_condense / rtoc_expand -- a solver that wants throughput packs whole horizons of many
// instances; this mirror exists so that the reference's own unit tests
// (test/dynamics/contact_dynamics_test.cpp) can be run against the HIP path line by line.
//
// robotoc::Robot here carries only what the path reads from the reference's Robot (dimensions,
// contact_inv_damping) plus the device context the calls run on; everything Pinocchio computes
// (RNEA derivatives, Baumgarte derivatives) is INPUT, handed over in ContactDynamicsData exactly as
// linearizeContactDynamics leaves it.
#ifndef ROBOTOC_HIP_DYNAMICS_HPP_
#define ROBOTOC_HIP_DYNAMICS_HPP_

#include "robotoc_hip.hpp"

namespace robotoc {

class Robot {
 public:
  Robot(int dimv, int dimu, int dim_passive, int max_dimf, int device = 0)
      : dims_{dimv, dimu, dim_passive, max_dimf}, ctx_(nullptr) {
    if (dimv <= 0 || dimu <= 0 || dimu + dim_passive != dimv || max_dimf < 0)
      throw std::invalid_argument("[Robot] inconsistent dimensions");
    const rtoc_dims d = dims_.c();
    check(rtoc_create(&d, 4, 1, device, &ctx_), "rtoc_create");  // [filler, THE grid point, filler, terminal]
    check(rtoc_set_option(ctx_, RTOC_OPT_CONDENSE_KEEP_QAF, 1), "rtoc_set_option");  // ContactDynamicsData exposes Qafqv / Qafu_full
    check(rtoc_get_layout(ctx_, &L_), "rtoc_get_layout");
  }
  ~Robot() {
    if (ctx_) rtoc_destroy(ctx_);
  }
  Robot(const Robot&) = delete;
  Robot& operator=(const Robot&) = delete;
  int dimv() const { return dims_.dimv; }
  int dimu() const { return dims_.dimu; }
  int dim_passive() const { return dims_.dim_passive; }
  int max_dimf() const { return dims_.max_dimf; }
  bool hasFloatingBase() const { return dims_.dim_passive > 0; }
  const RobotDims& dims() const { return dims_; }
  // RobotModelInfo::contact_inv_damping (robot_model_info.hpp:95, robot.hxx:662-664)
  void setContactInvDamping(double damping) {
    if (damping < 0) throw std::out_of_range("[Robot] invalid argument: contact_inv_damping must be non-negative!");
    int64_t bits;
    std::memcpy(&bits, &damping, sizeof(bits));
    check(rtoc_set_option(ctx_, RTOC_OPT_CONTACT_INV_DAMPING, bits), "rtoc_set_option");
  }
  rtoc_ctx* context() { return ctx_; }
  const rtoc_layout& layout() const { return L_; }
  static void check(int rc, const char* what) {
    if (rc != RTOC_OK) throw std::runtime_error(std::string("[Robot] ") + what + ": " + rtoc_error_string(rc));
  }

 private:
  RobotDims dims_;
  rtoc_ctx* ctx_;
  rtoc_layout L_;
};

class ContactStatus {  // include/robotoc/robot/contact_status.hpp: the hot path reads dimf() only
 public:
  explicit ContactStatus(int dimf = 0) : dimf_(dimf) {}
  int dimf() const { return dimf_; }

 private:
  int dimf_;
};
typedef ContactStatus ImpactStatus;  // include/robotoc/robot/impact_status.hpp

struct detail_access;

class ContactDynamicsData {  // include/robotoc/dynamics/contact_dynamics_data.hpp
 public:
  ContactDynamicsData() : dimf_(0), dims_(0), ctx_(nullptr) {}
  explicit ContactDynamicsData(const Robot& robot)
      : Qxu_passive(2 * robot.dimv(), robot.dim_passive()), Quu_passive_topRight(robot.dim_passive(), robot.dimu()),
        lu_passive(robot.dim_passive()), dIDda(robot.dimv(), robot.dimv()), dIDddv(robot.dimv(), robot.dimv()),
        dCda_full(robot.max_dimf(), robot.dimv()),
        dIDCdqv_full(robot.dimv() + robot.max_dimf(), 2 * robot.dimv()),
        MJtJinv_full(robot.dimv() + robot.max_dimf(), robot.dimv() + robot.max_dimf()),
        MJtJinv_dIDCdqv_full(robot.dimv() + robot.max_dimf(), 2 * robot.dimv()),
        Qafqv_full(robot.dimv() + robot.max_dimf(), 2 * robot.dimv()),
        Qafu_full_full(robot.dimv() + robot.max_dimf(), robot.dimv()), Phia_full(robot.max_dimf(), robot.dimv()),
        IDC_full(robot.dimv() + robot.max_dimf()), MJtJinv_IDC_full(robot.dimv() + robot.max_dimf()),
        laf_full(robot.dimv() + robot.max_dimf()), haf_full(robot.dimv() + robot.max_dimf()), r_(robot.dims()),
        dimf_(0), dims_(0), ctx_(nullptr) {}
  void setContactDimension(const int dimf) {
    if (dimf < 0 || dimf > r_.max_dimf) throw std::out_of_range("[ContactDynamicsData] invalid contact dimension");
    dimf_ = dimf;
  }
  void setSwitchingConstraintDimension(const int dims) {
    if (dims < 0 || dims > r_.max_dimf) throw std::out_of_range("[ContactDynamicsData] invalid switching constraint dimension");
    dims_ = dims;
  }
  int dimv() const { return r_.dimv; }
  int dimu() const { return r_.dimu; }
  int dimf() const { return dimf_; }
  int dimvf() const { return r_.dimv + dimf_; }
  int dims() const { return dims_; }
  int dim_passive() const { return r_.dim_passive; }
  bool hasFloatingBase() const { return r_.dim_passive > 0; }
  // reference member names; *_full = the max-size backing whose leading dimvf() / dimf() rows are active
  Mat Qxu_passive, Quu_passive_topRight;
  Vec lu_passive;
  Mat dIDda, dIDddv;         // dIDddv: impact stages (RNEAImpactDerivatives)
  Mat dCda_full;             // dimf x dimv
  Mat dIDCdqv_full;          // [dIDdq dIDdv; dCdq dCdv], dimvf x 2 dimv (impact: dIDdv = 0, dCdv is the J of MJtJinv)
  Mat MJtJinv_full;          // dimvf x dimvf
  Mat MJtJinv_dIDCdqv_full;  // dimvf x 2 dimv
  Mat Qafqv_full;            // dimvf x 2 dimv
  Mat Qafu_full_full;        // dimvf x dimv
  Mat Phia_full;             // dims x dimv
  Vec IDC_full, MJtJinv_IDC_full, laf_full, haf_full;
  double KKTError() const {  // contact_dynamics_data.hpp:204-206
    double e = 0;
    for (int i = 0; i < dimvf(); ++i) e += IDC_full(i) * IDC_full(i);
    for (int i = 0; i < r_.dim_passive; ++i) e += lu_passive(i) * lu_passive(i);
    return e;
  }

 private:
  friend struct detail_access;
  RobotDims r_;
  int dimf_, dims_;
  rtoc_ctx* ctx_;    // the condensed record of this stage is resident on the device of the Robot
  rtoc_layout L_;
  bool impact_ = false;
  double dt_ = 0.0;
};

struct detail_access {  // the free functions below share these two routines
  static void cp(double* dst, const double* src, size_t n) { std::memcpy(dst, src, n * sizeof(double)); }

  // grid [filler, stage, filler, terminal]: rtoc_set_grid wants an impact neither first nor within the last two
  static void set_stage_grid(rtoc_ctx* ctx, bool impact, int dimf, int dims, double dt) {
    rtoc_grid g[4];
    std::memset(g, 0, sizeof(g));
    for (int i = 0; i < 3; ++i) {
      g[i].type = RTOC_GRID_INTERMEDIATE;
      g[i].num_grids_in_phase = 1;  // evalKKT's 1/num_grids_in_phase scalings are the identity
      g[i].time_stage = 2;
      g[i].dt = dt > 0 ? dt : 1.0;
    }
    g[1].type = impact ? RTOC_GRID_IMPACT : RTOC_GRID_INTERMEDIATE;
    g[1].time_stage = impact ? -1 : 2;
    g[1].switching_constraint = !impact && dims > 0;
    g[1].dimf = dimf;
    g[1].dims = impact ? 0 : dims;
    g[1].dt = impact ? 0.0 : dt;
    g[3].type = RTOC_GRID_TERMINAL;
    Robot::check(rtoc_set_grid(ctx, g, 4), "rtoc_set_grid");
  }

  static void condense(Robot& robot, const bool impact, const int dimf, const double dt, ContactDynamicsData& data,
                       SplitKKTMatrix& kkt_matrix, SplitKKTResidual& kkt_residual) {
    if (dimf != data.dimf())
      throw std::invalid_argument("[condenseContactDynamics] data.setContactDimension() does not match the contact status");
    rtoc_ctx* ctx = robot.context();
    const rtoc_layout& L = robot.layout();
    const int nv = robot.dimv(), nu = robot.dimu(), np = robot.dim_passive(), nx = 2 * nv, nfm = robot.max_dimf(),
              nvfm = nv + nfm;
    set_stage_grid(ctx, impact, data.dimf(), data.dims(), dt);
    std::vector<double> kb(4 * static_cast<size_t>(L.kkt.stride), 0.0), cbuf(4 * static_cast<size_t>(L.cdd.stride), 0.0);
    for (int filler = 0; filler < 3; filler += 2)  // the filler grid points condense a unit inertia matrix
      for (int i = 0; i < nv; ++i) cbuf[filler * static_cast<size_t>(L.cdd.stride) + L.cdd.off[RTOC_CDD_DIDDA] + i + i * nv] = 1.0;
    double* kr = kb.data() + L.kkt.stride;
    double* cr = cbuf.data() + L.cdd.stride;
    const int* ko = L.kkt.off;
    const int* co = L.cdd.off;
    cp(kr + ko[RTOC_KKT_FXX], kkt_matrix.Fxx.data(), static_cast<size_t>(nx) * nx);
    cp(kr + ko[RTOC_KKT_QXX], kkt_matrix.Qxx.data(), static_cast<size_t>(nx) * nx);
    cp(kr + ko[RTOC_KKT_FX], kkt_residual.Fx.data(), nx);
    cp(kr + ko[RTOC_KKT_LX], kkt_residual.lx.data(), nx);
    if (!impact) {
      cp(kr + ko[RTOC_KKT_FVU], kkt_matrix.Fvu.data(), static_cast<size_t>(nv) * nu);
      cp(kr + ko[RTOC_KKT_QXU], kkt_matrix.Qxu.data(), static_cast<size_t>(nx) * nu);
      cp(kr + ko[RTOC_KKT_QUU], kkt_matrix.Quu.data(), static_cast<size_t>(nu) * nu);
      cp(kr + ko[RTOC_KKT_LU], kkt_residual.lu.data(), nu);
      cp(kr + ko[RTOC_KKT_FFX], kkt_matrix.fx.data(), nx);
      cp(kr + ko[RTOC_KKT_HX], kkt_matrix.hx.data(), nx);
      cp(kr + ko[RTOC_KKT_HU], kkt_matrix.hu.data(), nu);
      kr[ko[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_QTT] = kkt_matrix.Qtt;
      kr[ko[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_QTT_PREV] = kkt_matrix.Qtt_prev;
      kr[ko[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_H] = kkt_residual.h;
    }
    if (nfm > 0) {
      if (!impact) {
        cp(kr + ko[RTOC_KKT_PHIX], kkt_matrix.Phix_full.data(), static_cast<size_t>(nfm) * nx);
        cp(kr + ko[RTOC_KKT_PHIU], kkt_matrix.Phiu_full.data(), static_cast<size_t>(nfm) * nu);
        cp(kr + ko[RTOC_KKT_PHIT], kkt_matrix.Phit_full.data(), nfm);
        cp(kr + ko[RTOC_KKT_PRES], kkt_residual.P_full.data(), nfm);
        cp(cr + co[RTOC_CDD_DCDA], data.dCda_full.data(), static_cast<size_t>(nfm) * nv);
        cp(cr + co[RTOC_CDD_HF], kkt_matrix.hf_full.data(), nfm);
        cp(cr + co[RTOC_CDD_PHIA], data.Phia_full.data(), static_cast<size_t>(nfm) * nv);
      }
      cp(cr + co[RTOC_CDD_QFF], kkt_matrix.Qff_full.data(), static_cast<size_t>(nfm) * nfm);
      cp(cr + co[RTOC_CDD_QQF], kkt_matrix.Qqf_full.data(), static_cast<size_t>(nv) * nfm);
      cp(cr + co[RTOC_CDD_LF], kkt_residual.lf_full.data(), nfm);
    }
    cp(cr + co[RTOC_CDD_DIDDA], (impact ? data.dIDddv : data.dIDda).data(), static_cast<size_t>(nv) * nv);
    cp(cr + co[RTOC_CDD_DIDCDQV], data.dIDCdqv_full.data(), static_cast<size_t>(nvfm) * nx);
    cp(cr + co[RTOC_CDD_IDC], data.IDC_full.data(), nvfm);
    for (int i = 0; i < nv; ++i) cr[co[RTOC_CDD_QAA] + i] = impact ? kkt_matrix.Qdvdv(i, i) : kkt_matrix.Qaa(i, i);
    cp(cr + co[RTOC_CDD_LA], (impact ? kkt_residual.ldv : kkt_residual.la).data(), nv);
    if (!impact) {
      cp(cr + co[RTOC_CDD_HA], kkt_matrix.ha.data(), nv);
      for (int i = 0; i < np; ++i) cr[co[RTOC_CDD_LUP] + i] = data.lu_passive(i);
    }
    Robot::check(rtoc_upload(ctx, RTOC_BUF_KKT, 0, kb.data(), kb.size()), "rtoc_upload");
    Robot::check(rtoc_upload(ctx, RTOC_BUF_CDD, 0, cbuf.data(), cbuf.size()), "rtoc_upload");
    Robot::check(rtoc_clear_status(ctx), "rtoc_clear_status");
    Robot::check(rtoc_condense(ctx), "rtoc_condense");
    Robot::check(rtoc_download(ctx, RTOC_BUF_KKT, 0, kb.data(), kb.size()), "rtoc_download");
    Robot::check(rtoc_download(ctx, RTOC_BUF_CDD, 0, cbuf.data(), cbuf.size()), "rtoc_download");
    uint32_t st = 0;
    Robot::check(rtoc_status(ctx, &st, 1), "rtoc_status");
    if (st) throw std::runtime_error("[condenseContactDynamics] the inertia matrix or J M^-1 J^T is not positive definite");
    // mutated KKT members
    cp(kkt_matrix.Fxx.data(), kr + ko[RTOC_KKT_FXX], static_cast<size_t>(nx) * nx);
    cp(kkt_matrix.Qxx.data(), kr + ko[RTOC_KKT_QXX], static_cast<size_t>(nx) * nx);
    cp(kkt_residual.Fx.data(), kr + ko[RTOC_KKT_FX], nx);
    cp(kkt_residual.lx.data(), kr + ko[RTOC_KKT_LX], nx);
    if (!impact) {
      cp(kkt_matrix.Fvu.data(), kr + ko[RTOC_KKT_FVU], static_cast<size_t>(nv) * nu);
      cp(kkt_matrix.Qxu.data(), kr + ko[RTOC_KKT_QXU], static_cast<size_t>(nx) * nu);
      cp(kkt_matrix.Quu.data(), kr + ko[RTOC_KKT_QUU], static_cast<size_t>(nu) * nu);
      cp(kkt_residual.lu.data(), kr + ko[RTOC_KKT_LU], nu);
      cp(kkt_matrix.hx.data(), kr + ko[RTOC_KKT_HX], nx);
      cp(kkt_matrix.hu.data(), kr + ko[RTOC_KKT_HU], nu);
      kkt_residual.h = kr[ko[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_H];
      if (nfm > 0 && data.dims() > 0) {
        cp(kkt_matrix.Phix_full.data(), kr + ko[RTOC_KKT_PHIX], static_cast<size_t>(nfm) * nx);
        cp(kkt_matrix.Phiu_full.data(), kr + ko[RTOC_KKT_PHIU], static_cast<size_t>(nfm) * nu);
        cp(kkt_matrix.Phit_full.data(), kr + ko[RTOC_KKT_PHIT], nfm);
        cp(kkt_residual.P_full.data(), kr + ko[RTOC_KKT_PRES], nfm);
      }
    }
    // condensed ContactDynamicsData
    cp(data.MJtJinv_full.data(), cr + co[RTOC_CDD_MJTJINV], static_cast<size_t>(nvfm) * nvfm);
    cp(data.MJtJinv_dIDCdqv_full.data(), cr + co[RTOC_CDD_MJD], static_cast<size_t>(nvfm) * nx);
    cp(data.MJtJinv_IDC_full.data(), cr + co[RTOC_CDD_MJIDC], nvfm);
    cp(data.Qafqv_full.data(), cr + co[RTOC_CDD_QAFQV], static_cast<size_t>(nvfm) * nx);
    cp(data.laf_full.data(), cr + co[RTOC_CDD_LAF], nvfm);
    if (!impact) {
      cp(data.Qafu_full_full.data(), cr + co[RTOC_CDD_QAFU], static_cast<size_t>(nvfm) * nv);
      cp(data.haf_full.data(), cr + co[RTOC_CDD_HAF], nvfm);
      if (np > 0) {
        cp(data.Qxu_passive.data(), cr + co[RTOC_CDD_QXUP], static_cast<size_t>(nx) * np);
        cp(data.Quu_passive_topRight.data(), cr + co[RTOC_CDD_QUUPTR], static_cast<size_t>(np) * nu);
        for (int i = 0; i < np; ++i) data.lu_passive(i) = cr[co[RTOC_CDD_LUP] + i];
      }
    }
    data.ctx_ = ctx;
    data.L_ = L;
    data.impact_ = impact;
    data.dt_ = dt;
  }

  // primal and dual expansion are one device routine; `dual` selects what is read back and whether the
  // in-place update of laf (contact_dynamics.cpp:190-198) is kept
  static void expand(const bool dual, const double dt, const double dts, const ContactDynamicsData& data,
                     double* laf_out, const SplitDirection* d_next, SplitDirection& d) {
    if (!data.ctx_) throw std::logic_error("[expandContactDynamics] the condensation has not run on this data");
    rtoc_ctx* ctx = data.ctx_;
    if (dual && !data.impact_ && dt != data.dt_) set_stage_grid(ctx, false, data.dimf(), data.dims(), dt);
    const rtoc_layout& L = data.L_;
    const int nv = data.dimv(), nu = data.dimu(), nx = 2 * nv, np = data.dim_passive(), nfm = data.r_.max_dimf,
              nvfm = nv + nfm;
    const size_t laf_at = static_cast<size_t>(L.cdd.stride) + L.cdd.off[RTOC_CDD_LAF];
    std::vector<double> db(4 * static_cast<size_t>(L.dir.stride), 0.0);
    double* r1 = db.data() + L.dir.stride;
    double* r2 = db.data() + 2 * static_cast<size_t>(L.dir.stride);
    cp(r1 + L.dir.off[RTOC_DIR_DX], d.dx.data(), nx);
    if (!data.impact_) cp(r1 + L.dir.off[RTOC_DIR_DU], d.du.data(), nu);
    if (nfm > 0) cp(r1 + L.dir.off[RTOC_DIR_DXI], d.dxi_full.data(), nfm);
    r1[L.dir.off[RTOC_DIR_DTS] + 0] = 0.0;  // the device forms (dts_next - dts) / num_grids_in_phase, = dts here
    r1[L.dir.off[RTOC_DIR_DTS] + 1] = dts;
    if (d_next) cp(r2 + L.dir.off[RTOC_DIR_DLMDGMM], d_next->dlmdgmm.data(), nx);
    std::vector<double> laf(nvfm);
    Robot::check(rtoc_download(ctx, RTOC_BUF_CDD, laf_at, laf.data(), laf.size()), "rtoc_download");
    Robot::check(rtoc_upload(ctx, RTOC_BUF_DIR, 0, db.data(), db.size()), "rtoc_upload");
    Robot::check(rtoc_expand(ctx, 0.995), "rtoc_expand");
    Robot::check(rtoc_download(ctx, RTOC_BUF_DIR, 0, db.data(), db.size()), "rtoc_download");
    if (!dual) {
      Robot::check(rtoc_upload(ctx, RTOC_BUF_CDD, laf_at, laf.data(), laf.size()), "rtoc_upload");  // laf untouched
      cp(d.daf_full.data(), r1 + L.dir.off[RTOC_DIR_DAF], nvfm);
    } else {
      cp(d.dbetamu_full.data(), r1 + L.dir.off[RTOC_DIR_DBETAMU], nvfm);
      if (!data.impact_)
        for (int i = 0; i < np; ++i) d.dnu_passive(i) = r1[L.dir.off[RTOC_DIR_DNUP] + i];
      Robot::check(rtoc_download(ctx, RTOC_BUF_CDD, laf_at, laf_out, nvfm), "rtoc_download");
    }
  }
};

// condenseContactDynamics (contact_dynamics.cpp:55-164)
inline void condenseContactDynamics(Robot& robot, const ContactStatus& contact_status, const double dt,
                                    ContactDynamicsData& data, SplitKKTMatrix& kkt_matrix,
                                    SplitKKTResidual& kkt_residual) {
  if (dt <= 0) throw std::out_of_range("[condenseContactDynamics] invalid argument: dt must be positive!");
  detail_access::condense(robot, false, contact_status.dimf(), dt, data, kkt_matrix, kkt_residual);
}

// expandContactDynamicsPrimal (contact_dynamics.cpp:167-174): d.daf() = [da; df] from d.dx, d.du
inline void expandContactDynamicsPrimal(const ContactDynamicsData& data, SplitDirection& d) {
  detail_access::expand(false, 0.0, 0.0, data, nullptr, nullptr, d);
}

// expandContactDynamicsDual (contact_dynamics.cpp:177-202): d.dbetamu() = [dbeta; dmu], d.dnu_passive;
// updates data.laf() in place like the reference (:190-198).
inline void expandContactDynamicsDual(const double dt, const double dts, ContactDynamicsData& data,
                                      const SplitDirection& d_next, SplitDirection& d) {
  if (dt <= 0) throw std::out_of_range("[expandContactDynamicsDual] invalid argument: dt must be positive!");
  detail_access::expand(true, dt, dts, data, data.laf_full.data(), &d_next, d);
}

// condenseImpactDynamics (impact_dynamics.cpp:38-80): data.dIDddv, dCdv (inside dIDCdqv), kkt_matrix.Qdvdv,
// kkt_residual.ldv take the places of dIDda, dCda, Qaa, la; there is no control input.
inline void condenseImpactDynamics(Robot& robot, const ImpactStatus& impact_status, ContactDynamicsData& data,
                                   SplitKKTMatrix& kkt_matrix, SplitKKTResidual& kkt_residual) {
  detail_access::condense(robot, true, impact_status.dimf(), 0.0, data, kkt_matrix, kkt_residual);
}

// expandImpactDynamicsPrimal (impact_dynamics.cpp:83-88): d.ddvf() = [ddv; df] (stored in d.daf_full)
inline void expandImpactDynamicsPrimal(const ContactDynamicsData& data, SplitDirection& d) {
  detail_access::expand(false, 0.0, 0.0, data, nullptr, nullptr, d);
}

// expandImpactDynamicsDual (impact_dynamics.cpp:91-96); updates data.ldvf() (= laf) in place
inline void expandImpactDynamicsDual(ContactDynamicsData& data, const SplitDirection& d_next, SplitDirection& d) {
  detail_access::expand(true, 0.0, 0.0, data, data.laf_full.data(), &d_next, d);
}

}  // namespace robotoc

#endif  // ROBOTOC_HIP_DYNAMICS_HPP_

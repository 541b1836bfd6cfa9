// ref_capi.cpp -- C entry points that run the REFERENCE'S OWN sources (compiled where they lie under
// /root/reference by oracle/Makefile.ref, against oracle/ref_shim) on this repo's packed records.
// TEST INFRASTRUCTURE ONLY (oracle/_ref/librtoc_ref.so): it pins the C restatement in oracle/*.c -- and, through
// the committed golden vectors in tests/golden/ref_*.npz, the HIP path -- to what robotoc's code computes.
//
// What runs here is reference code:  RiccatiRecursion::{backward,forward}RiccatiRecursion
// (src/riccati/riccati_recursion.cpp) with RiccatiFactorizer / BackwardRiccatiRecursionFactorizer, the
// Unconstr* variants, condenseContactDynamics / condenseImpactDynamics and the expansions
// (src/dynamics/contact_dynamics.cpp, impact_dynamics.cpp), the floating-base corrections
// (src/dynamics/state_equation.cpp:68-109, impact_state_equation.cpp:57-72), UnconstrDynamics, the containers of
// src/core.  What does NOT: Eigen (absent; oracle/ref_shim/mini_eigen.hpp evaluates the same expressions eagerly),
// Pinocchio (absent; Robot is a dimension-only stand-in whose computeMJtJinv uses dense LLT), OCP /
// TimeDiscretization (stand-ins carrying the members the Riccati sources read).  This file only copies between
// records and the reference's containers.
#include <cstring>
#include <vector>

#include "../../include/rtoc.h"

// the STO policies of RiccatiRecursion have no accessor in the reference; the wrapper reads the member
#define private public
#include "robotoc/riccati/riccati_recursion.hpp"
#undef private
#include "robotoc/riccati/unconstr_riccati_recursion.hpp"
#include "robotoc/dynamics/contact_dynamics.hpp"
#include "robotoc/dynamics/impact_dynamics.hpp"
#include "robotoc/dynamics/state_equation.hpp"
#include "robotoc/dynamics/unconstr_dynamics.hpp"

using namespace robotoc;

namespace {

Robot make_robot(const rtoc_layout* L, int contact_dim, double damping = 0.0) {
  std::vector<ContactType> types;
  const int nc = contact_dim > 0 ? L->dims.nf_max / contact_dim : 0;
  for (int i = 0; i < nc; ++i) types.push_back(contact_dim == 3 ? ContactType::PointContact : ContactType::SurfaceContact);
  return Robot(L->dims.nv, L->dims.nu, types, damping);
}

// column-major block of a record (leading dimension ld) <-> Eigen-like object
template <class M>
void get_mat(M&& m, const double* src, int ld) {
  for (int j = 0; j < (int)m.cols(); ++j)
    for (int i = 0; i < (int)m.rows(); ++i) m(i, j) = src[i + (size_t)j * ld];
}
template <class M>
void put_mat(const M& m, double* dst, int ld) {
  for (int j = 0; j < (int)m.cols(); ++j)
    for (int i = 0; i < (int)m.rows(); ++i) dst[i + (size_t)j * ld] = m(i, j);
}
template <class V>
void get_vec(V&& v, const double* src) {
  for (int i = 0; i < (int)v.size(); ++i) v(i) = src[i];
}
template <class V>
void put_vec(const V& v, double* dst) {
  for (int i = 0; i < (int)v.size(); ++i) dst[i] = v(i);
}

GridInfo to_grid(const rtoc_grid& g) {
  GridInfo o;
  o.type = g.type == RTOC_GRID_IMPACT ? GridType::Impact
                                      : (g.type == RTOC_GRID_LIFT ? GridType::Lift
                                                                  : (g.type == RTOC_GRID_TERMINAL ? GridType::Terminal : GridType::Intermediate));
  o.dt = g.dt;
  o.sto = g.sto != 0;
  o.sto_next = g.sto_next != 0;
  o.switching_constraint = g.switching_constraint != 0;
  o.num_grids_in_phase = g.num_grids_in_phase;
  return o;
}

void load_kkt(const rtoc_layout* L, const rtoc_grid& g, const double* rec, SplitKKTMatrix& km, SplitKKTResidual& kr) {
  const int* o = L->kkt.off;
  const int nx = L->nx, nv = L->dims.nv, nu = L->dims.nu, lds = L->dims.ns_max > 0 ? L->dims.ns_max : 1;
  const bool terminal = g.type == RTOC_GRID_TERMINAL, impact = g.type == RTOC_GRID_IMPACT;
  const int ns = (terminal || impact) ? 0 : g.dims;
  km.setContactDimension(g.dimf);
  kr.setContactDimension(g.dimf);
  km.setSwitchingConstraintDimension(ns);
  kr.setSwitchingConstraintDimension(ns);
  get_mat(km.Qxx, rec + o[RTOC_KKT_QXX], nx);
  get_vec(kr.lx, rec + o[RTOC_KKT_LX]);
  if (terminal) return;
  get_mat(km.Fxx, rec + o[RTOC_KKT_FXX], nx);
  get_vec(kr.Fx, rec + o[RTOC_KKT_FX]);
  if (impact) return;
  get_mat(km.Fvu, rec + o[RTOC_KKT_FVU], nv);
  get_mat(km.Qxu, rec + o[RTOC_KKT_QXU], nx);
  get_mat(km.Quu, rec + o[RTOC_KKT_QUU], nu);
  get_vec(kr.lu, rec + o[RTOC_KKT_LU]);
  get_vec(km.fx, rec + o[RTOC_KKT_FFX]);
  get_vec(km.hx, rec + o[RTOC_KKT_HX]);
  get_vec(km.hu, rec + o[RTOC_KKT_HU]);
  km.Qtt = rec[o[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_QTT];
  km.Qtt_prev = rec[o[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_QTT_PREV];
  kr.h = rec[o[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_H];
  if (ns > 0) {
    get_mat(km.Phix(), rec + o[RTOC_KKT_PHIX], lds);
    get_mat(km.Phiu(), rec + o[RTOC_KKT_PHIU], lds);
    get_vec(km.Phit(), rec + o[RTOC_KKT_PHIT]);
    get_vec(kr.P(), rec + o[RTOC_KKT_PRES]);
  }
}

// the blocks the backward recursion mutates in place (riccati_factorizer_test.cpp:65-66)
void store_kkt_mutated(const rtoc_layout* L, const rtoc_grid& g, double* rec, const SplitKKTMatrix& km,
                       const SplitKKTResidual& kr) {
  const int* o = L->kkt.off;
  const int nx = L->nx, nu = L->dims.nu;
  if (g.type == RTOC_GRID_TERMINAL) return;
  put_mat(km.Qxx, rec + o[RTOC_KKT_QXX], nx);
  if (g.type == RTOC_GRID_IMPACT) return;
  put_mat(km.Qxu, rec + o[RTOC_KKT_QXU], nx);
  put_mat(km.Quu, rec + o[RTOC_KKT_QUU], nu);
  put_vec(kr.lu, rec + o[RTOC_KKT_LU]);
}

void store_ric(const rtoc_layout* L, const rtoc_grid& g, double* rec, const SplitRiccatiFactorization& r,
               const LQRPolicy* lqr, const STOPolicy* sto) {
  const int* o = L->ric.off;
  const int nx = L->nx, nu = L->dims.nu, lds = L->dims.ns_max > 0 ? L->dims.ns_max : 1;
  put_mat(r.P, rec + o[RTOC_RIC_P], nx);
  put_vec(r.s, rec + o[RTOC_RIC_S]);
  put_vec(r.Psi, rec + o[RTOC_RIC_PSI]);
  put_vec(r.Phi, rec + o[RTOC_RIC_PHI]);
  put_vec(r.psi_x, rec + o[RTOC_RIC_PSIX]);
  put_vec(r.phi_x, rec + o[RTOC_RIC_PHIX]);
  put_vec(r.psi_u, rec + o[RTOC_RIC_PSIU]);
  put_vec(r.phi_u, rec + o[RTOC_RIC_PHIU]);
  double* sc = rec + o[RTOC_RIC_SCAL];
  sc[RTOC_RIC_SCAL_XI] = r.xi;
  sc[RTOC_RIC_SCAL_CHI] = r.chi;
  sc[RTOC_RIC_SCAL_RHO] = r.rho;
  sc[RTOC_RIC_SCAL_ETA] = r.eta;
  sc[RTOC_RIC_SCAL_IOTA] = r.iota;
  if (lqr) {
    for (int i = 0; i < nu; ++i)
      for (int j = 0; j < nx; ++j) rec[o[RTOC_RIC_K] + (size_t)i * nx + j] = lqr->K(i, j);  // row-major like the reference
    put_vec(lqr->k, rec + o[RTOC_RIC_KV]);
    put_vec(lqr->T, rec + o[RTOC_RIC_T]);
    put_vec(lqr->W, rec + o[RTOC_RIC_W]);
  }
  if (r.dims() > 0) {
    put_mat(r.M(), rec + o[RTOC_RIC_M], lds);
    put_vec(r.m(), rec + o[RTOC_RIC_MV]);
    put_vec(r.mt(), rec + o[RTOC_RIC_MT]);
    put_vec(r.mt_next(), rec + o[RTOC_RIC_MTN]);
  }
  if (sto) {
    put_vec(sto->dtsdx, rec + o[RTOC_RIC_DTSDX]);
    sc[RTOC_RIC_SCAL_DTSDTS] = sto->dtsdts;
    sc[RTOC_RIC_SCAL_DTS0] = sto->dts0;
  }
  (void)g;
}

void store_dir(const rtoc_layout* L, double* rec, const SplitDirection& d, bool has_u) {
  const int* o = L->dir.off;
  put_vec(d.dx, rec + o[RTOC_DIR_DX]);
  if (has_u) put_vec(d.du, rec + o[RTOC_DIR_DU]);
  put_vec(d.dlmdgmm, rec + o[RTOC_DIR_DLMDGMM]);
  if (d.dims() > 0) put_vec(d.dxi(), rec + o[RTOC_DIR_DXI]);
  rec[o[RTOC_DIR_DTS]] = d.dts;
  rec[o[RTOC_DIR_DTS] + 1] = d.dts_next;
}

}  // namespace

extern "C" {

// RiccatiRecursion::backwardRiccatiRecursion (+ forwardRiccatiRecursion if do_forward) of ONE instance.
// kkt [nstages][stride] is mutated like the reference mutates its KKT containers; ric / dir receive the outputs
// (dir[0].dx is the input d[0].dx).  contact_dim: 3 point / 6 surface contacts (only sizes max_dimf).
int ref_riccati_sweep(const rtoc_layout* L, const rtoc_grid* grid, int nstages, double* kkt, double* ric, double* dir,
                      double max_dts0, int contact_dim, int do_forward) {
  Robot robot = make_robot(L, contact_dim);
  OCP ocp;
  ocp.robot = robot;
  ocp.N = nstages - 1;
  ocp.reserved_num_discrete_events = 0;
  std::vector<GridInfo> gi;
  for (int i = 0; i < nstages; ++i) gi.push_back(to_grid(grid[i]));
  TimeDiscretization td(gi);
  KKTMatrix km(nstages, SplitKKTMatrix(robot));
  KKTResidual kr(nstages, SplitKKTResidual(robot));
  RiccatiFactorization fac(nstages, SplitRiccatiFactorization(robot));
  Direction d(nstages, SplitDirection(robot));
  for (int i = 0; i < nstages; ++i) load_kkt(L, grid[i], kkt + (size_t)i * L->kkt.stride, km[i], kr[i]);
  RiccatiRecursion rec(ocp, max_dts0);
  rec.backwardRiccatiRecursion(td, km, kr, fac);
  const auto& lqr = rec.getLQRPolicy();
  for (int i = 0; i < nstages; ++i) {
    const bool has_u = i < nstages - 1 && grid[i].type != RTOC_GRID_IMPACT;
    store_ric(L, grid[i], ric + (size_t)i * L->ric.stride, fac[i], has_u ? &lqr[i] : nullptr, &rec.sto_policy_[i]);
    store_kkt_mutated(L, grid[i], kkt + (size_t)i * L->kkt.stride, km[i], kr[i]);
  }
  if (do_forward) {
    for (int i = 0; i < nstages; ++i) {
      const bool plain = i < nstages - 1 && grid[i].type != RTOC_GRID_IMPACT;
      d[i].setSwitchingConstraintDimension(plain ? grid[i].dims : 0);
    }
    get_vec(d[0].dx, dir + L->dir.off[RTOC_DIR_DX]);
    rec.forwardRiccatiRecursion(td, km, kr, fac, d);
    for (int i = 0; i < nstages; ++i)
      store_dir(L, dir + (size_t)i * L->dir.stride, d[i], i < nstages - 1 && grid[i].type != RTOC_GRID_IMPACT);
  }
  return 0;
}

// The same sweep, timed (bench.py: cpu_baseline.reference_sources): `reps` backward + forward recursions of ONE instance by the
// reference's RiccatiRecursion.  The recursion mutates its KKT containers, so they are reloaded from the (read-only) record before
// every repetition; only the two recursion calls are inside the clock -- the way OCPBenchmarker times updateSolution
// (include/robotoc/utils/ocp_benchmarker.hxx:14-32: a steady clock around the solver call).  out[0] = seconds inside the clock,
// out[1] = seconds of the reloads.  NOTE what this measures: the reference's sources at -O2 with assertions on, their Eigen
// expressions evaluated eagerly by oracle/ref_shim/mini_eigen.hpp (temporaries, no vectorised kernels) -- NOT an Eigen build.
}  // extern "C"
#include <chrono>
extern "C" {
int ref_riccati_sweep_bench(const rtoc_layout* L, const rtoc_grid* grid, int nstages, const double* kkt, const double* dx0,
                            double max_dts0, int contact_dim, int reps, double* out) {
  Robot robot = make_robot(L, contact_dim);
  OCP ocp;
  ocp.robot = robot;
  ocp.N = nstages - 1;
  ocp.reserved_num_discrete_events = 0;
  std::vector<GridInfo> gi;
  for (int i = 0; i < nstages; ++i) gi.push_back(to_grid(grid[i]));
  TimeDiscretization td(gi);
  KKTMatrix km(nstages, SplitKKTMatrix(robot));
  KKTResidual kr(nstages, SplitKKTResidual(robot));
  RiccatiFactorization fac(nstages, SplitRiccatiFactorization(robot));
  Direction d(nstages, SplitDirection(robot));
  RiccatiRecursion rec(ocp, max_dts0);
  for (int i = 0; i < nstages; ++i) {
    const bool plain = i < nstages - 1 && grid[i].type != RTOC_GRID_IMPACT;
    d[i].setSwitchingConstraintDimension(plain ? grid[i].dims : 0);
  }
  double t_rec = 0.0, t_load = 0.0;
  typedef std::chrono::steady_clock clk;
  for (int r = 0; r < reps; ++r) {
    const clk::time_point t0 = clk::now();
    for (int i = 0; i < nstages; ++i) load_kkt(L, grid[i], kkt + (size_t)i * L->kkt.stride, km[i], kr[i]);
    get_vec(d[0].dx, dx0);
    const clk::time_point t1 = clk::now();
    rec.backwardRiccatiRecursion(td, km, kr, fac);
    rec.forwardRiccatiRecursion(td, km, kr, fac, d);
    const clk::time_point t2 = clk::now();
    t_load += std::chrono::duration<double>(t1 - t0).count();
    t_rec += std::chrono::duration<double>(t2 - t1).count();
  }
  out[0] = t_rec;
  out[1] = t_load;
  return 0;
}

// UnconstrRiccatiRecursion::{backward,forward}RiccatiRecursion of one instance (uniform dt = T / N).
// Records: Quu / lu / Qxu slots hold Qaa / la / [Qqa; Qva] (the acceleration is the control).
int ref_unconstr_sweep(const rtoc_layout* L, int nstages, double dt, double* kkt, double* ric, double* dir, int do_forward) {
  Robot robot(L->dims.nv, L->dims.nu, std::vector<ContactType>());
  OCP ocp;
  ocp.robot = robot;
  ocp.N = nstages - 1;
  ocp.T = dt * (nstages - 1);
  const int* o = L->kkt.off;
  const int nx = L->nx, nv = L->dims.nv;
  KKTMatrix km(nstages, SplitKKTMatrix(robot));
  KKTResidual kr(nstages, SplitKKTResidual(robot));
  UnconstrRiccatiFactorization fac(nstages, SplitRiccatiFactorization(robot));
  Direction d(nstages, SplitDirection(robot));
  for (int i = 0; i < nstages; ++i) {
    const double* rec = kkt + (size_t)i * L->kkt.stride;
    get_mat(km[i].Qxx, rec + o[RTOC_KKT_QXX], nx);
    get_vec(kr[i].lx, rec + o[RTOC_KKT_LX]);
    if (i == nstages - 1) break;
    get_mat(km[i].Qxu, rec + o[RTOC_KKT_QXU], nx);
    get_mat(km[i].Qaa, rec + o[RTOC_KKT_QUU], nv);
    get_vec(kr[i].Fx, rec + o[RTOC_KKT_FX]);
    get_vec(kr[i].la, rec + o[RTOC_KKT_LU]);
  }
  UnconstrRiccatiRecursion rr(ocp);
  rr.backwardRiccatiRecursion(km, kr, fac);
  const auto& lqr = rr.getLQRPolicy();
  rtoc_grid g;
  std::memset(&g, 0, sizeof(g));
  for (int i = 0; i < nstages; ++i)
    store_ric(L, g, ric + (size_t)i * L->ric.stride, fac[i], i < nstages - 1 ? &lqr[i] : nullptr, nullptr);
  if (do_forward) {
    get_vec(d[0].dx, dir + L->dir.off[RTOC_DIR_DX]);
    rr.forwardRiccatiRecursion(kr, fac, d);
    for (int i = 0; i < nstages; ++i) {
      double* rec = dir + (size_t)i * L->dir.stride;
      put_vec(d[i].dx, rec + L->dir.off[RTOC_DIR_DX]);
      // the unconstrained recursion's control is the acceleration (SplitDirection::da() = daf head)
      if (i < nstages - 1) put_vec(d[i].da(), rec + L->dir.off[RTOC_DIR_DU]);
      put_vec(d[i].dlmdgmm, rec + L->dir.off[RTOC_DIR_DLMDGMM]);
    }
  }
  return 0;
}

// condenseContactDynamics (contact_dynamics.cpp:55-164) or, on impact grids, condenseImpactDynamics
// (impact_dynamics.cpp:38-80) of one grid point, in place on the KKT and ContactDynamicsData records.  The scalings
// of IntermediateStage::evalKKT's tail (intermediate_stage.cpp:140-148; that file needs the cost / constraint
// libraries) are NOT applied: with apply_tail = 0 the oracle's orc_condense_stage_core is compared.
int ref_condense_stage(const rtoc_layout* L, const rtoc_grid* g, double* kkt_rec, double* cdd_rec, double damping,
                       int contact_dim) {
  Robot robot = make_robot(L, contact_dim, damping);
  const int nv = L->dims.nv, nu = L->dims.nu, nx = L->nx, np = L->dims.np;
  const int nf = g->dimf, nvf = nv + nf;
  const int ldv = L->nvf_max, ldf = L->dims.nf_max > 0 ? L->dims.nf_max : 1, lds = L->dims.ns_max > 0 ? L->dims.ns_max : 1;
  const bool impact = g->type == RTOC_GRID_IMPACT;
  const int ns = impact ? 0 : g->dims;
  const int* ko = L->kkt.off;
  const int* co = L->cdd.off;
  SplitKKTMatrix km(robot);
  SplitKKTResidual kr(robot);
  ContactDynamicsData data(robot);
  km.setContactDimension(nf);
  kr.setContactDimension(nf);
  data.setContactDimension(nf);
  km.setSwitchingConstraintDimension(ns);
  kr.setSwitchingConstraintDimension(ns);
  // state equation / cost blocks as linearizeStateEquation and the cost leave them
  get_mat(km.Fxx, kkt_rec + ko[RTOC_KKT_FXX], nx);
  get_mat(km.Qxx, kkt_rec + ko[RTOC_KKT_QXX], nx);
  get_vec(kr.Fx, kkt_rec + ko[RTOC_KKT_FX]);
  get_vec(kr.lx, kkt_rec + ko[RTOC_KKT_LX]);
  // ContactDynamicsData inputs
  get_mat(data.dIDCdqv(), cdd_rec + co[RTOC_CDD_DIDCDQV], ldv);
  get_vec(data.IDC(), cdd_rec + co[RTOC_CDD_IDC]);
  if (nf > 0) {
    get_mat(km.Qff(), cdd_rec + co[RTOC_CDD_QFF], ldf);
    get_mat(km.Qqf(), cdd_rec + co[RTOC_CDD_QQF], nv);
    get_vec(kr.lf(), cdd_rec + co[RTOC_CDD_LF]);
  }
  if (impact) {
    ImpactStatus status = robot.createImpactStatus();
    // activate contacts until dimf is reached (which contacts are active does not enter the arithmetic)
    for (int c = 0, f = 0; c < robot.maxNumContacts() && f < nf; ++c) {
      status.activateImpact(c);
      f += contact_dim;
    }
    get_mat(data.dIDddv, cdd_rec + co[RTOC_CDD_DIDDA], nv);
    for (int i = 0; i < nv; ++i) km.Qdvdv(i, i) = cdd_rec[co[RTOC_CDD_QAA] + i];
    get_vec(kr.ldv, cdd_rec + co[RTOC_CDD_LA]);
    condenseImpactDynamics(robot, status, data, km, kr);
  } else {
    ContactStatus status = robot.createContactStatus();
    for (int c = 0, f = 0; c < robot.maxNumContacts() && f < nf; ++c) {
      status.activateContact(c);
      f += contact_dim;
    }
    get_mat(km.Fvu, kkt_rec + ko[RTOC_KKT_FVU], nv);
    get_mat(km.Qxu, kkt_rec + ko[RTOC_KKT_QXU], nx);
    get_mat(km.Quu, kkt_rec + ko[RTOC_KKT_QUU], nu);
    get_vec(kr.lu, kkt_rec + ko[RTOC_KKT_LU]);
    get_vec(km.hx, kkt_rec + ko[RTOC_KKT_HX]);
    get_vec(km.hu, kkt_rec + ko[RTOC_KKT_HU]);
    kr.h = kkt_rec[ko[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_H];
    get_mat(data.dIDda, cdd_rec + co[RTOC_CDD_DIDDA], nv);
    if (nf > 0) {
      get_mat(data.dCda(), cdd_rec + co[RTOC_CDD_DCDA], ldf);
      get_vec(km.hf(), cdd_rec + co[RTOC_CDD_HF]);
    }
    for (int i = 0; i < nv; ++i) km.Qaa(i, i) = cdd_rec[co[RTOC_CDD_QAA] + i];
    get_vec(kr.la, cdd_rec + co[RTOC_CDD_LA]);
    get_vec(km.ha, cdd_rec + co[RTOC_CDD_HA]);
    if (np > 0) get_vec(data.lu_passive, cdd_rec + co[RTOC_CDD_LUP]);
    if (ns > 0) {
      get_mat(km.Phix(), kkt_rec + ko[RTOC_KKT_PHIX], lds);
      get_mat(km.Phia(), cdd_rec + co[RTOC_CDD_PHIA], lds);
      get_vec(km.Phit(), kkt_rec + ko[RTOC_KKT_PHIT]);
      get_vec(kr.P(), kkt_rec + ko[RTOC_KKT_PRES]);
    }
    condenseContactDynamics(robot, status, g->dt, data, km, kr);
  }
  // ---- outputs ----
  put_mat(km.Fxx, kkt_rec + ko[RTOC_KKT_FXX], nx);
  put_mat(km.Qxx, kkt_rec + ko[RTOC_KKT_QXX], nx);
  put_vec(kr.Fx, kkt_rec + ko[RTOC_KKT_FX]);
  put_vec(kr.lx, kkt_rec + ko[RTOC_KKT_LX]);
  put_mat(data.MJtJinv(), cdd_rec + co[RTOC_CDD_MJTJINV], ldv);
  put_mat(data.MJtJinv_dIDCdqv(), cdd_rec + co[RTOC_CDD_MJD], ldv);
  put_vec(data.MJtJinv_IDC(), cdd_rec + co[RTOC_CDD_MJIDC]);
  put_mat(data.Qafqv(), cdd_rec + co[RTOC_CDD_QAFQV], ldv);
  put_vec(data.laf(), cdd_rec + co[RTOC_CDD_LAF]);
  if (!impact) {
    put_mat(km.Fvu, kkt_rec + ko[RTOC_KKT_FVU], nv);
    put_mat(km.Qxu, kkt_rec + ko[RTOC_KKT_QXU], nx);
    put_mat(km.Quu, kkt_rec + ko[RTOC_KKT_QUU], nu);
    put_vec(kr.lu, kkt_rec + ko[RTOC_KKT_LU]);
    put_vec(km.hx, kkt_rec + ko[RTOC_KKT_HX]);
    put_vec(km.hu, kkt_rec + ko[RTOC_KKT_HU]);
    kkt_rec[ko[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_H] = kr.h;
    put_mat(data.Qafu_full(), cdd_rec + co[RTOC_CDD_QAFU], ldv);
    put_vec(data.haf(), cdd_rec + co[RTOC_CDD_HAF]);
    if (np > 0) {
      put_mat(data.Qxu_passive, cdd_rec + co[RTOC_CDD_QXUP], nx);
      put_mat(data.Quu_passive_topRight, cdd_rec + co[RTOC_CDD_QUUPTR], np);
      put_vec(data.lu_passive, cdd_rec + co[RTOC_CDD_LUP]);
    }
    if (ns > 0) {
      put_mat(km.Phix(), kkt_rec + ko[RTOC_KKT_PHIX], lds);
      put_mat(km.Phiu(), kkt_rec + ko[RTOC_KKT_PHIU], lds);
      put_vec(km.Phit(), kkt_rec + ko[RTOC_KKT_PHIT]);
      put_vec(kr.P(), kkt_rec + ko[RTOC_KKT_PRES]);
    }
  }
  (void)nvf;
  return 0;
}

// expandContactDynamicsPrimal + expandContactDynamicsDual (contact_dynamics.cpp:167-202), or the impact forms
// (impact_dynamics.cpp:83-96), of one grid point after ref_condense_stage on the same ContactDynamicsData record.
// dts = (dts_next - dts) / num_grids_in_phase as IntermediateStage::expandDual computes it.
int ref_expand_stage(const rtoc_layout* L, const rtoc_grid* g, double* cdd_rec, double* dir_rec, const double* dir_next_rec,
                     int contact_dim) {
  Robot robot = make_robot(L, contact_dim);
  const int nv = L->dims.nv, nx = L->nx, np = L->dims.np;
  const int nf = g->dimf;
  const int ldv = L->nvf_max, lds = L->dims.ns_max > 0 ? L->dims.ns_max : 1;
  const bool impact = g->type == RTOC_GRID_IMPACT;
  const int ns = impact ? 0 : g->dims;
  const int* co = L->cdd.off;
  const int* d_o = L->dir.off;
  ContactDynamicsData data(robot);
  data.setContactDimension(nf);
  data.setSwitchingConstraintDimension(ns);
  get_mat(data.MJtJinv(), cdd_rec + co[RTOC_CDD_MJTJINV], ldv);
  get_mat(data.MJtJinv_dIDCdqv(), cdd_rec + co[RTOC_CDD_MJD], ldv);
  get_vec(data.MJtJinv_IDC(), cdd_rec + co[RTOC_CDD_MJIDC]);
  get_mat(data.Qafqv(), cdd_rec + co[RTOC_CDD_QAFQV], ldv);
  get_vec(data.laf(), cdd_rec + co[RTOC_CDD_LAF]);
  SplitDirection d(robot), dn(robot);
  d.setContactDimension(nf);
  d.setSwitchingConstraintDimension(ns);
  get_vec(d.dx, dir_rec + d_o[RTOC_DIR_DX]);
  get_vec(dn.dlmdgmm, dir_next_rec + d_o[RTOC_DIR_DLMDGMM]);
  if (impact) {
    expandImpactDynamicsPrimal(data, d);
    expandImpactDynamicsDual(data, dn, d);
  } else {
    get_vec(d.du, dir_rec + d_o[RTOC_DIR_DU]);
    get_mat(data.Qafu_full(), cdd_rec + co[RTOC_CDD_QAFU], ldv);
    get_vec(data.haf(), cdd_rec + co[RTOC_CDD_HAF]);
    if (np > 0) {
      get_mat(data.Qxu_passive, cdd_rec + co[RTOC_CDD_QXUP], nx);
      get_mat(data.Quu_passive_topRight, cdd_rec + co[RTOC_CDD_QUUPTR], np);
      get_vec(data.lu_passive, cdd_rec + co[RTOC_CDD_LUP]);
    }
    if (ns > 0) {
      get_mat(data.Phia(), cdd_rec + co[RTOC_CDD_PHIA], lds);
      get_vec(d.dxi(), dir_rec + d_o[RTOC_DIR_DXI]);
    }
    double dts = 0.0;
    if (g->num_grids_in_phase > 0)
      dts = (dir_rec[d_o[RTOC_DIR_DTS] + 1] - dir_rec[d_o[RTOC_DIR_DTS]]) / (double)g->num_grids_in_phase;
    expandContactDynamicsPrimal(data, d);
    expandContactDynamicsDual(g->dt, dts, data, dn, d);
    if (np > 0) put_vec(d.dnu_passive, dir_rec + d_o[RTOC_DIR_DNUP]);
  }
  put_vec(d.daf(), dir_rec + d_o[RTOC_DIR_DAF]);
  put_vec(d.dbetamu(), dir_rec + d_o[RTOC_DIR_DBETAMU]);
  put_vec(data.laf(), cdd_rec + co[RTOC_CDD_LAF]);
  (void)nv;
  return 0;
}

// correctCostateDirection (state_equation.cpp:90-96) on one direction record; se3_rec = the RTOC_BUF_SE3 record
// (Fqq_inv, Fqq_prev_inv).  (correctLinearizeStateEquation / correctLinearizeImpactStateEquation recompute those
// inverses with Pinocchio's dSubtractConfiguration inside the same function -- state_equation.cpp:76-79 -- so they
// cannot run here; their 6x6 products are pinned by the closed form in tests/test_state_equation_correction.py.)
int ref_correct_costate(const rtoc_layout* L, const double* se3_rec, double* dir_rec) {
  Robot robot(L->dims.nv, L->dims.nu, std::vector<ContactType>());
  StateEquationData data(robot);
  get_mat(data.Fqq_prev_inv, se3_rec + RTOC_SE3_FQQ_PREV_INV, 6);
  SplitDirection d(robot);
  get_vec(d.dlmdgmm, dir_rec + L->dir.off[RTOC_DIR_DLMDGMM]);
  correctCostateDirection(data, d);
  put_vec(d.dlmdgmm, dir_rec + L->dir.off[RTOC_DIR_DLMDGMM]);
  return 0;
}

// SplitSolution::integrate (src/core/split_solution.cpp:58-90) on one packed record pair.  q_integrated: the result of
// robot.integrateConfiguration for a floating base (injected; NULL for a fixed base, where it is q + step dq).
int ref_split_solution_integrate(const rtoc_layout* L, const rtoc_grid* g, int contact_dim, double step, const double* dir_rec,
                                 double* sol_rec, const double* q_integrated) {
  Robot robot = make_robot(L, contact_dim);
  const int nv = L->dims.nv, nu = L->dims.nu, np = L->dims.np, nq = nv + (np == 6 ? 1 : 0);
  const bool impact = g->type == RTOC_GRID_IMPACT;
  const int ns = impact ? 0 : g->dims;
  SplitSolution s(robot);
  SplitDirection d(robot);
  {
    const int nact = contact_dim > 0 ? g->dimf / contact_dim : 0;
    if (impact) {
      ImpactStatus is = robot.createImpactStatus();
      for (int c = 0; c < nact; ++c) is.activateImpact(c);
      s.setContactStatus(is);
    } else {
      ContactStatus cs = robot.createContactStatus();
      for (int c = 0; c < nact; ++c) cs.activateContact(c);
      s.setContactStatus(cs);
    }
  }
  d.setContactDimension(g->dimf);
  s.setSwitchingConstraintDimension(ns), d.setSwitchingConstraintDimension(ns);
  const int* so = L->sol.off;
  const int* dof = L->dir.off;
  for (int i = 0; i < nq; ++i) s.q(i) = sol_rec[so[RTOC_SOL_Q] + i];
  for (int i = 0; i < nv; ++i) {
    s.v(i) = sol_rec[so[RTOC_SOL_V] + i];
    (impact ? s.dv(i) : s.a(i)) = sol_rec[so[RTOC_SOL_A] + i];
    s.lmd(i) = sol_rec[so[RTOC_SOL_LMD] + i], s.gmm(i) = sol_rec[so[RTOC_SOL_GMM] + i], s.beta(i) = sol_rec[so[RTOC_SOL_BETA] + i];
    d.dq()(i) = dir_rec[dof[RTOC_DIR_DX] + i], d.dv()(i) = dir_rec[dof[RTOC_DIR_DX] + nv + i];
    d.daf()(i) = dir_rec[dof[RTOC_DIR_DAF] + i];
    d.dlmd()(i) = dir_rec[dof[RTOC_DIR_DLMDGMM] + i], d.dgmm()(i) = dir_rec[dof[RTOC_DIR_DLMDGMM] + nv + i];
    d.dbetamu()(i) = dir_rec[dof[RTOC_DIR_DBETAMU] + i];
  }
  for (int i = 0; i < nu; ++i) s.u(i) = sol_rec[so[RTOC_SOL_U] + i], d.du(i) = dir_rec[dof[RTOC_DIR_DU] + i];
  for (int i = 0; i < g->dimf; ++i) {
    s.f_stack()(i) = sol_rec[so[RTOC_SOL_F] + i], s.mu_stack()(i) = sol_rec[so[RTOC_SOL_MU] + i];
    d.daf()(nv + i) = dir_rec[dof[RTOC_DIR_DAF] + nv + i], d.dbetamu()(nv + i) = dir_rec[dof[RTOC_DIR_DBETAMU] + nv + i];
  }
  for (int i = 0; i < np; ++i) s.nu_passive(i) = sol_rec[so[RTOC_SOL_NUP] + i], d.dnu_passive(i) = dir_rec[dof[RTOC_DIR_DNUP] + i];
  for (int i = 0; i < ns; ++i) s.xi_stack()(i) = sol_rec[so[RTOC_SOL_XI] + i], d.dxi()(i) = dir_rec[dof[RTOC_DIR_DXI] + i];
  if (np == 6) {
    Eigen::VectorXd qi(nq);
    for (int i = 0; i < nq; ++i) qi(i) = q_integrated[i];
    robot.inject("integrateConfiguration", qi);
  }
  s.integrate(robot, step, d, impact);
  for (int i = 0; i < nq; ++i) sol_rec[so[RTOC_SOL_Q] + i] = s.q(i);
  for (int i = 0; i < nv; ++i) {
    sol_rec[so[RTOC_SOL_V] + i] = s.v(i);
    sol_rec[so[RTOC_SOL_A] + i] = impact ? s.dv(i) : s.a(i);
    sol_rec[so[RTOC_SOL_LMD] + i] = s.lmd(i), sol_rec[so[RTOC_SOL_GMM] + i] = s.gmm(i), sol_rec[so[RTOC_SOL_BETA] + i] = s.beta(i);
  }
  for (int i = 0; i < nu; ++i) sol_rec[so[RTOC_SOL_U] + i] = s.u(i);
  for (int i = 0; i < g->dimf; ++i) sol_rec[so[RTOC_SOL_F] + i] = s.f_stack()(i), sol_rec[so[RTOC_SOL_MU] + i] = s.mu_stack()(i);
  for (int i = 0; i < np; ++i) sol_rec[so[RTOC_SOL_NUP] + i] = s.nu_passive(i);
  for (int i = 0; i < ns; ++i) sol_rec[so[RTOC_SOL_XI] + i] = s.xi_stack()(i);
  return 0;
}

int ref_version(void) { return 1; }

}  // extern "C"

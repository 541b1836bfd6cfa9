// contact_eval_kkt.hpp -- the cost part of {Intermediate,Impact,Terminal}Stage::evalKKT on the device, for the contact path.
//
// kkt_matrix.setZero() / kkt_residual.setZero() (reference src/ocp/intermediate_stage.cpp:100-103) and
// cost_->quadratizeStageCost / ImpactCost / TerminalCost for a ConfigurationSpaceCost
// (src/cost/configuration_space_cost.cpp:274-324 stage, :381-470 impact, :343-378 terminal): diagonal weights on
// q - q_ref, v - v_ref, a, u - u_ref (dv on impact grids), with q - q_ref on the configuration manifold:
//   qdiff = log6(M_ref^-1 M) for a free-flyer base (Robot::subtractConfiguration(q, q_ref)), joints Euclidean
//   lq += dt J^T W qdiff,  Qqq += dt J^T W J,  J = Jlog6(M_ref^-1 M) (dSubtractConfiguration_dqf; forward mode through the log)
// The state equation (state_equation_lin.hpp) and the contact dynamics (rigid_body.hpp) then ADD their terms, in the order
// of the reference.  Records: the un-condensed contact-path convention (Qaa / la / lf ... in RTOC_BUF_CDD).
// cost table (device): 12 arrays of M = nv + 1 doubles: q_ref | v_ref | u_ref | wq | wv | wa | wu | wq_T | wv_T | wq_I | wv_I | wdv_I
#pragma once
#include "state_equation_lin.hpp"

namespace rtoc {

struct CostArgs {
  const double* sol;
  const double* cost;
  double* kkt;
  double* cdd;
  const rtoc_grid* grid;
  int nstages, batch, nv, nu, nf_max, ns_max, floating;
  int sol_stride, kkt_stride, cdd_stride;
  int o_q, o_v, o_a, o_u;
  rtoc_record_layout kl, cl;
  const double* dt_inst;  // per-instance time steps or nullptr (grid_dt)
  double* cost_out;       // [batch][nstages] value of the stage / impact / terminal cost (evalOCP: line search), or nullptr
};

// kkt_matrix.setZero() / kkt_residual.setZero() of every grid point as one stream over the (contiguous) KKT records -- 16 B per
// lane, a workgroup per record and trip: 6 TB/s, where the same zeros written field by field from inside contact_cost_kernel
// (whose waves then sat on their slots through the cost evaluation) reached 2.5 TB/s -- AND the constant part of
// quadratizeStageCost / ImpactCost / TerminalCost of a ConfigurationSpaceCost, the diagonals dt W of Qqq (joints), Qvv and Quu:
// written from the cost kernel they were 8-byte stores 37 doubles apart, a partial line each (0.4 of its 0.9 ms) -- and, for the
// same reason, the constant part of linearizeStateEquation: Fqq = I, Fqv = dt I (state_equation.cpp:29-40; the base corner of a
// floating base is overwritten by state_equation_lin_kernel).
struct InitArgs {
  double* kkt;
  const double* cost;      // the table of contact_cost_kernel
  const rtoc_grid* grid;
  const double* dt_inst;
  int nstages, batch, nv, nu, floating, kkt_stride, o_qxx, o_quu, o_fxx;
};
static __global__ __launch_bounds__(256) void init_records_kernel(InitArgs a) {
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  const int nv = a.nv, nu = a.nu, nx = 2 * nv, M = nv + 1, nb = a.floating ? 6 : 0;
  const double *wq = a.cost + 3 * M, *wv = wq + M, *wu = wv + 2 * M, *wqT = wu + M, *wvT = wqT + M, *wqI = wvT + M, *wvI = wqI + M;
  const int half = a.kkt_stride / 2, qxx0 = a.o_qxx, qxx1 = a.o_qxx + nx * nx, quu0 = a.o_quu, quu1 = a.o_quu + nu * nu;
  const int fxx0 = a.o_fxx, fxx1 = a.o_fxx + nx * nx;
  const long long nrec = (long long)a.batch * a.nstages;
  for (long long rec = blockIdx.x; rec < nrec; rec += gridDim.x) {
    const int b = (int)(rec / a.nstages), st = (int)(rec % a.nstages);
    const bool impact = a.grid[st].type == RTOC_GRID_IMPACT, terminal = st == a.nstages - 1;
    const double scale = (impact || terminal) ? 1.0 : grid_dt(a.grid, a.dt_inst, b, a.nstages, st);
    const double* const Wq = terminal ? wqT : impact ? wqI : wq;
    const double* const Wv = terminal ? wvT : impact ? wvI : wv;
    const bool sto = !terminal && !impact;
    const double dt = impact ? 0.0 : scale;   // terminal records: no state equation
    dbl2* const p2 = reinterpret_cast<dbl2*>(a.kkt + rec * a.kkt_stride);
    for (int t = threadIdx.x; t < half; t += 256) {
      dbl2 v = {0.0, 0.0};
      const int w = 2 * t;   // the pair (w, w + 1) of the record; fields start on multiples of 8 doubles
      if (w >= qxx0 && w < qxx1) {
        const int d = w - qxx0, c = d / nx, r = d - c * nx;   // column-major: rows r, r + 1 of column c (nx is even)
        if (r == c || r + 1 == c) {
          const double dv = c < nv ? (c >= nb ? scale * Wq[c] : 0.0) : scale * Wv[c - nv];   // the base block: contact_cost_kernel
          if (r == c) v.x = dv;
          else v.y = dv;
        }
      } else if (!terminal && w >= fxx0 && w < fxx1) {
        const int d = w - fxx0, c = d / nx, r = d - c * nx;   // rows r, r + 1 of column c; the top half of Fxx only
        if (r < nv) v.x = c == r ? 1.0 : (c == nv + r ? dt : 0.0);
        if (r + 1 < nv) v.y = c == r + 1 ? 1.0 : (c == nv + r + 1 ? dt : 0.0);
      } else if (sto && w >= quu0 && w < quu1) {
        const int d = w - quu0, c = d / nu, r = d - c * nu;
        // nu may be odd: the pair can straddle two columns
        if (r == c) v.x = scale * wu[c];
        const int r1 = r + 1 < nu ? r + 1 : 0, c1 = r + 1 < nu ? c : c + 1;
        if (r1 == c1 && c1 < nu && w + 1 < quu1) v.y = scale * wu[c1];
      }
      p2[t] = v;
    }
  }
}

// COST_GP grid points per wave, COST_LW lanes each: the arithmetic of a grid point lives in a handful of lanes (the Log6 of the base
// placement and its derivative in six), and one wave per grid point spent ~1.5k instructions on mostly idle lanes -- four grid
// points share the instruction stream now (the per-joint loops take two trips at nv = 18).  A wave whose last grid points lie
// beyond the batch repeats the last one (identical values to identical addresses).
constexpr int COST_GP = 4, COST_LW = 64 / COST_GP;
static __global__ __launch_bounds__(64) void contact_cost_kernel(CostArgs a) {
  using namespace selin;
  __shared__ double Js[COST_GP][36], wds[COST_GP][8];
  const int lane = threadIdx.x % COST_LW, grp = threadIdx.x / COST_LW;   // `lane`: within the grid point's COST_LW lanes
  double* const J = Js[grp];
  double* const wd = wds[grp];
  const long long nitems = (long long)a.batch * a.nstages;
  long long item = (long long)blockIdx.x * COST_GP + grp;
  item = item < nitems ? item : nitems - 1;
  const int b = (int)(item / a.nstages), st = (int)(item % a.nstages);
  const rtoc_grid g = a.grid[st];
  const bool impact = g.type == RTOC_GRID_IMPACT, terminal = st == a.nstages - 1;
  const int nv = a.nv, nu = a.nu, nx = 2 * nv, nb = a.floating ? 6 : 0, M = nv + 1, np = nv - nu;
  const size_t rec = (size_t)b * a.nstages + st;
  const double* const s = a.sol + rec * a.sol_stride;
  double* const kr = a.kkt + rec * a.kkt_stride;
  double* const cr = a.cdd + rec * a.cdd_stride;
  const double *qr = a.cost, *vr = qr + M, *ur = vr + M, *wq = ur + M, *wv = wq + M, *wa = wv + M, *wu = wa + M, *wqT = wu + M,
               *wvT = wqT + M, *wqI = wvT + M, *wvI = wqI + M, *wdvI = wvI + M;
  const double scale = (impact || terminal) ? 1.0 : grid_dt(a.grid, a.dt_inst, b, a.nstages, st);
  const double* const Wq = terminal ? wqT : impact ? wqI : wq;
  const double* const Wv = terminal ? wvT : impact ? wvI : wv;
  // ---- setZero of the RTOC_BUF_CDD fields the stages below accumulate into (the KKT record: init_records_kernel) ----
  // 16-byte stores (the fields start on 64-byte boundaries: rtoc_record_finish pads every field to 8 doubles)
  auto zero = [&](double* p, int n) {
    double2* const p2 = reinterpret_cast<double2*>(p);
    const int n2 = n >> 1;
    for (int e = lane; e < n2; e += COST_LW) p2[e] = make_double2(0.0, 0.0);
    if ((n & 1) && lane == 0) p[n - 1] = 0.0;
  };
  // the CDD fields the stages below accumulate into, except what the cost terms write themselves (QAA, LA, HA of a grid point
  // with accelerations): the fill needs no ordering against them and runs at the END of the kernel, behind the loads and the
  // arithmetic.  The KKT record was zeroed as a whole by init_records_kernel, which also wrote the diagonals dt W of Qqq (joints),
  // Qvv and Quu.
  auto zero_cdd = [&]() {
    if (a.ns_max > 0) zero(cr + a.cl.off[RTOC_CDD_PHIA], a.ns_max * nv);
    if (terminal) {
      zero(cr + a.cl.off[RTOC_CDD_QAA], nv);
      zero(cr + a.cl.off[RTOC_CDD_LA], nv);
      zero(cr + a.cl.off[RTOC_CDD_HA], nv);
    }
    zero(cr + a.cl.off[RTOC_CDD_LUP], 8);
    if (a.nf_max > 0) {
      zero(cr + a.cl.off[RTOC_CDD_QFF], a.nf_max * a.nf_max);
      zero(cr + a.cl.off[RTOC_CDD_QQF], nv * a.nf_max);
      zero(cr + a.cl.off[RTOC_CDD_LF], a.nf_max);
      zero(cr + a.cl.off[RTOC_CDD_HF], a.nf_max);
    }
  };
  const double *q = s + a.o_q, *v = s + a.o_v, *acc = s + a.o_a, *u = s + a.o_u;
  double* const Qxx = kr + a.kl.off[RTOC_KKT_QXX];
  double* const lx = kr + a.kl.off[RTOC_KKT_LX];
  // ---- joints of q, v, a, u ----
  // The STO sensitivities of the stage cost on intermediate grids (intermediate_stage.cpp:103-108): h = cost / dt and
  // hx, hu, ha = lx, lu, la / dt BEFORE constraints and dynamics add their terms -- i.e. the cost gradient without the dt
  const bool sto = !terminal && !impact;
  double* const hx = kr + a.kl.off[RTOC_KKT_HX];
  double hval = 0.0;   // this lane's share of cost / dt (cost itself on impact / terminal grids) = 1/2 sum of weight * difference^2
  for (int i = lane; i < nv; i += COST_LW) {
    if (i >= nb) {
      const double dq = q[(nb ? 1 : 0) + i] - qr[(nb ? 1 : 0) + i];
      lx[i] = scale * Wq[i] * dq;
      if (sto) hx[i] = Wq[i] * dq;
      hval += 0.5 * Wq[i] * dq * dq;
    }
    const double dv = v[i] - vr[i];
    lx[nv + i] = scale * Wv[i] * dv;
    if (sto) hx[nv + i] = Wv[i] * dv;
    hval += 0.5 * Wv[i] * dv * dv;
    if (!terminal) {
      const double w = impact ? wdvI[i] : scale * wa[i];   // a on contact grids, dv on impact grids (both in the A slot)
      cr[a.cl.off[RTOC_CDD_LA] + i] = w * acc[i];
      cr[a.cl.off[RTOC_CDD_QAA] + i] = w;
      cr[a.cl.off[RTOC_CDD_HA] + i] = sto ? wa[i] * acc[i] : 0.0;
      hval += 0.5 * (impact ? wdvI[i] : wa[i]) * acc[i] * acc[i];
    }
  }
  if (sto)
    for (int i = lane; i < nu; i += COST_LW) {
      const double du = u[i] - ur[i];
      kr[a.kl.off[RTOC_KKT_LU] + i] = scale * wu[i] * du;
      kr[a.kl.off[RTOC_KKT_HU] + i] = wu[i] * du;
      hval += 0.5 * wu[i] * du * du;
    }
  (void)np;
  // ---- the free-flyer base of q: qdiff = log6(M_ref^-1 M), J = Jlog6 ----
  if (a.floating) {
    M3 Rx;
    V3 px;
    rel(quat_R(qr + 3), rbd::ldv3(qr), quat_R(q + 3), rbd::ldv3(q), Rx, px);
    if (lane < 6) {
      SV val, der;
      rbd::log6_fwd(Rx, px, unit_twist(lane), val, der);
      const double c[6] = {der.l.x, der.l.y, der.l.z, der.a.x, der.a.y, der.a.z}, d[6] = {val.l.x, val.l.y, val.l.z, val.a.x, val.a.y, val.a.z};
#pragma unroll
      for (int r = 0; r < 6; ++r) J[r + 6 * lane] = c[r];
      wd[lane] = Wq[lane] * d[lane];
    }
    __syncthreads();
    for (int e = lane; e < 36; e += COST_LW) {
      const int r = e % 6, c = e / 6;
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) t += J[k + 6 * r] * Wq[k] * J[k + 6 * c];
      Qxx[r + (size_t)c * nx] = scale * t;
    }
    if (lane < 6) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) t += J[k + 6 * lane] * wd[k];
      lx[lane] = scale * t;
      if (sto) hx[lane] = t;
      hval += 0.5 * wd[lane] * wd[lane] / (Wq[lane] != 0.0 ? Wq[lane] : 1.0);   // 1/2 Wq d^2, wd = Wq d
    }
  }
  if (sto || a.cost_out) {
#pragma unroll
    for (int off = COST_LW / 2; off > 0; off >>= 1) hval += __shfl_xor(hval, off, 64);   // within the grid point's lanes
    if (lane == 0 && sto) kr[a.kl.off[RTOC_KKT_SCAL] + RTOC_KKT_SCAL_H] = hval;
    // the value of the cost itself (evalStageCost / evalImpactCost / evalTerminalCost): what evalOCP sums for the line search
    if (lane == 0 && a.cost_out) a.cost_out[rec] = scale * hval;
  }
  zero_cdd();
}

}  // namespace rtoc

// state_equation_lin.hpp -- linearisation of the (forward-Euler) state equation on the device.
//
// Replaces linearizeStateEquation (reference src/dynamics/state_equation.cpp:29-66) on intermediate / lift grids and
// linearizeImpactStateEquation (src/dynamics/impact_state_equation.cpp:27-57) on impact grids, and produces what the
// correctLinearize* steps (state_equation.cpp:69-90, impact :60-75; rtoc_condense applies them) take from
// StateEquationData: Fqq_inv and Fqq_prev_inv (RTOC_BUF_SE3).  For a floating base the configuration difference and its
// Jacobians are Pinocchio's difference / dDifference on SE(3):
//   Fq (base) = log6(M_next^-1 M)                      [+ dt v]            (Robot::subtractConfiguration(q, q_next))
//   Fqq       = d/dM      = Jlog6(X1),  X1 = M_next^-1 M                   (dSubtractConfiguration_dqf)
//   Fqq_prev  = d/dM of log6(M^-1 M_prev) = -Jlog6(X0) Ad_{X0^-1},  X0 = M^-1 M_prev   (dSubtractConfiguration_dq0(q_prev, q))
//   Fqq_inv   = [ -Jlog6(X1) Ad_{X1^-1} ]^-1           (dSubtractConfiguration_dq0(q, q_next), state_equation.cpp:78-79)
// Jlog6 x twist is evaluated by forward mode through the log (rbd::log6_fwd); Ad_{X^-1} is the motion actInv.  Six lanes
// carry the six unit twists, lane 0 inverts the two 6 x 6 Jacobians by their block-triangular shape (3 x 3 cofactor inverses,
// like the reference's SE3JacobianInverse).  Everything else is elementwise.
// One wave per (instance, grid point).  Records: the un-condensed contact-path convention (la in CDD.la).
#pragma once
#include "rigid_body.hpp"

namespace rtoc {

struct SeLinArgs {
  const double* sol;
  const double* x0;      // [batch][nq + nv]: q_prev of grid point 0 (the initial state), may be nullptr -> q_prev = q
  double* kkt;
  double* cdd;
  double* se3;           // may be nullptr on a fixed base
  double* dx0;           // may be nullptr: computeInitialStateDirection (state_equation.cpp:99-109) into RTOC_BUF_DX0
  const rtoc_grid* grid;
  int nstages, batch, nv, floating;
  int zeroed;            // the KKT record was initialised just before (init_records_kernel: zeros, Fqq = I, Fqv = dt I): only the base corner is written
  int sol_stride, kkt_stride, cdd_stride;
  int o_q, o_v, o_a, o_lmd, o_gmm;
  int o_fxx, o_fx, o_lx, o_hx, o_ffx, o_scal;
  int o_la, o_ha;
  const double* dt_inst;  // per-instance time steps or nullptr (grid_dt)
};

namespace selin {
using rbd::M3;
using rbd::SV;
using rbd::V3;
__device__ __forceinline__ M3 quat_R(const double* q) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  M3 R;
  R.m[0] = 1 - 2 * (y * y + z * z), R.m[1] = 2 * (x * y - z * w), R.m[2] = 2 * (x * z + y * w);
  R.m[3] = 2 * (x * y + z * w), R.m[4] = 1 - 2 * (x * x + z * z), R.m[5] = 2 * (y * z - x * w);
  R.m[6] = 2 * (x * z - y * w), R.m[7] = 2 * (y * z + x * w), R.m[8] = 1 - 2 * (x * x + y * y);
  return R;
}
// X = A^-1 B for placements A = (Ra, pa), B = (Rb, pb)
__device__ __forceinline__ void rel(const M3& Ra, V3 pa, const M3& Rb, V3 pb, M3& R, V3& p) {
  M3 Rat;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Rat.m[3 * r + c] = Ra.m[3 * c + r];
  R = rbd::mul(Rat, Rb);
  p = rbd::mulT(Ra, pb - pa);
}
__device__ __forceinline__ SV unit_twist(int k) {
  return SV{rbd::mk(k == 0, k == 1, k == 2), rbd::mk(k == 3, k == 4, k == 5)};
}
// in-place inverse of a 6 x 6 column-major matrix (one lane); returns false if singular to working precision
__device__ inline bool inv6(double* A) {
  double B[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) B[e] = (e % 7 == 0) ? 1.0 : 0.0;
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    double best = fabs(A[c + 6 * c]);
    for (int r = c + 1; r < 6; ++r)
      if (fabs(A[r + 6 * c]) > best) best = fabs(A[r + 6 * c]), piv = r;
    if (!(best > 1e-300)) return false;
    if (piv != c)
      for (int k = 0; k < 6; ++k) {
        double t = A[c + 6 * k];
        A[c + 6 * k] = A[piv + 6 * k], A[piv + 6 * k] = t;
        t = B[c + 6 * k];
        B[c + 6 * k] = B[piv + 6 * k], B[piv + 6 * k] = t;
      }
    const double d = 1.0 / A[c + 6 * c];
    for (int k = 0; k < 6; ++k) A[c + 6 * k] *= d, B[c + 6 * k] *= d;
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double f = A[r + 6 * c];
      for (int k = 0; k < 6; ++k) A[r + 6 * k] -= f * A[c + 6 * k], B[r + 6 * k] -= f * B[c + 6 * k];
    }
  }
  for (int e = 0; e < 36; ++e) A[e] = B[e];
  return true;
}
// Inverse of a 6 x 6 block upper-triangular matrix [[A, B], [0, D]] (column-major, 3 x 3 blocks) -- the shape of Jlog6 and
// of Ad in the (linear, angular) ordering, hence of every SE(3) difference Jacobian here: [[A^-1, -A^-1 B D^-1], [0, D^-1]]
// with the 3 x 3 inverses by cofactors (what the reference's SE3JacobianInverse exploits, se3_jacobian_inverse.hxx).  All
// indices are compile-time constants: the arrays stay in registers (the general inv6 above pivots, i.e. indexes at run time,
// which puts its 72 doubles into scratch memory).
__device__ __forceinline__ void inv3(const double (&m)[3][3], double (&o)[3][3]) {
  const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1], c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2], c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  const double id = 1.0 / (m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02);
  o[0][0] = c00 * id, o[1][0] = c01 * id, o[2][0] = c02 * id;
  o[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id, o[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id, o[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
  o[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id, o[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id, o[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
}
__device__ __forceinline__ void inv6_block_ut(const double* J, double* out) {
  double A[3][3], B[3][3], D[3][3], Ai[3][3], Di[3][3], T[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) A[r][c] = J[r + 6 * c], B[r][c] = J[r + 6 * (c + 3)], D[r][c] = J[r + 3 + 6 * (c + 3)];
  inv3(A, Ai);
  inv3(D, Di);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) T[r][c] = Ai[r][0] * B[0][c] + Ai[r][1] * B[1][c] + Ai[r][2] * B[2][c];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      out[r + 6 * c] = Ai[r][c];
      out[r + 3 + 6 * c] = 0.0;
      out[r + 3 + 6 * (c + 3)] = Di[r][c];
      out[r + 6 * (c + 3)] = -(T[r][0] * Di[0][c] + T[r][1] * Di[1][c] + T[r][2] * Di[2][c]);
    }
}
}  // namespace selin

static __global__ __launch_bounds__(64) void state_equation_lin_kernel(SeLinArgs a) {
  using namespace selin;
  __shared__ double J[3][36];  // Fqq, Fqq_prev, d/dq0 of (q (-) q_next): 6 x 6, column-major
  const int lane = threadIdx.x;
  const int b = blockIdx.x / a.nstages, st = blockIdx.x % a.nstages;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  const bool impact = g.type == RTOC_GRID_IMPACT;
  if (st == a.nstages - 1) {
    // linearizeTerminalStateEquation (src/dynamics/terminal_state_equation.cpp:8-28): lq -= lmd (base: += Fqq_prev^T lmd),
    // lv -= gmm; Fqq_prev_inv for correctCostateDirection / correctLinearizeTerminalStateEquation
    const int nv = a.nv, nb = a.floating ? 6 : 0;
    const size_t rec = (size_t)b * a.nstages + st;
    const double* const s = a.sol + rec * a.sol_stride;
    double* const kr = a.kkt + rec * a.kkt_stride;
    for (int i = lane; i < nv; i += 64) {
      if (i >= nb) kr[a.o_lx + i] -= s[a.o_lmd + i];
      kr[a.o_lx + nv + i] -= s[a.o_gmm + i];
    }
    if (a.floating) {
      const double* q = s + a.o_q;
      const double* qp = s - a.sol_stride + a.o_q;
      M3 R0;
      V3 p0;
      rel(quat_R(q + 3), rbd::ldv3(q), quat_R(qp + 3), rbd::ldv3(qp), R0, p0);
      if (lane < 6) {
        SV val, der;
        rbd::log6_fwd(R0, p0, rbd::sv0() - rbd::act_inv(R0, p0, unit_twist(lane)), val, der);
        const double c0[6] = {der.l.x, der.l.y, der.l.z, der.a.x, der.a.y, der.a.z};
#pragma unroll
        for (int r = 0; r < 6; ++r) J[1][r + 6 * lane] = c0[r];
      }
      __syncthreads();
      if (lane < 6) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) t += J[1][r + 6 * lane] * s[a.o_lmd + r];
        kr[a.o_lx + lane] += t;
      }
      if (lane == 0 && a.se3) {
        double* const se = a.se3 + rec * RTOC_SE3_STRIDE;
        double A[36];
        inv6_block_ut(J[1], A);
#pragma unroll
        for (int e = 0; e < 36; ++e) se[e] = 0.0, se[36 + e] = A[e];
      }
    }
    return;
  }
  const double dt = impact ? 0.0 : grid_dt(a.grid, a.dt_inst, b, a.nstages, st);
  const int nv = a.nv, nx = 2 * nv, nb = a.floating ? 6 : 0, nq = nv + (a.floating ? 1 : 0);
  const size_t rec = (size_t)b * a.nstages + st;
  const double* const s = a.sol + rec * a.sol_stride;
  const double* const sn = s + a.sol_stride;
  const double* const qp = st > 0 ? s - a.sol_stride + a.o_q : (a.x0 ? a.x0 + (size_t)b * (nq + nv) : s + a.o_q);
  double* const kr = a.kkt + rec * a.kkt_stride;
  double* const cr = a.cdd + rec * a.cdd_stride;
  const double *q = s + a.o_q, *v = s + a.o_v, *acc = s + a.o_a, *lmd = s + a.o_lmd, *gmm = s + a.o_gmm;
  const double *qn = sn + a.o_q, *vn = sn + a.o_v, *lmdn = sn + a.o_lmd, *gmmn = sn + a.o_gmm;
  // ---- Fxx top half: Fqq = I (joints), Fqv = dt I; the bottom half belongs to the dynamics condensation ----
  if (a.zeroed) {
    // rtoc_contact_eval_kkt: init_records_kernel has just written the record, zeros and the diagonals Fqq = I, Fqv = dt I with
    // them (scattered 8-byte stores from here were a partial line each)
  } else {
    int r = lane % nx, c = lane / nx;
    for (int e = lane; e < nx * nx; e += 64) {
      if (r < nv) kr[a.o_fxx + e] = (c == r) ? 1.0 : (c == nv + r ? dt : 0.0);
      r += 64;
      while (r >= nx) r -= nx, ++c;
    }
  }
  // ---- joints and velocities ----
  for (int i = lane; i < nv; i += 64) {
    if (i >= nb) {
      kr[a.o_fx + i] = q[(nb ? 1 : 0) + i] + dt * v[i] - qn[(nb ? 1 : 0) + i];   // Fq (:17-18), joints; q carries 7 base entries
      kr[a.o_lx + i] += lmdn[i] - lmd[i];                                         // (:47-48 / :52)
    }
    kr[a.o_fx + nv + i] = v[i] + (impact ? acc[i] : dt * acc[i]) - vn[i];         // Fv (:19; impact :15)
    kr[a.o_lx + nv + i] += dt * lmdn[i] + gmmn[i] - gmm[i];                       // lv (:55; impact :53)
    cr[a.o_la + i] += impact ? gmmn[i] : dt * gmmn[i];                            // la (:56) / ldv (impact :54)
    if (!impact) {                                                                 // STO sensitivities (:58-63)
      kr[a.o_hx + nv + i] += lmdn[i];
      cr[a.o_ha + i] += gmmn[i];
      kr[a.o_ffx + i] = v[i];
      kr[a.o_ffx + nv + i] = acc[i];
    }
  }
  if (!impact) {
    double h = 0.0;
    for (int i = lane; i < nv; i += 64) h += lmdn[i] * v[i] + gmmn[i] * acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off, 64);
    if (lane == 0) kr[a.o_scal + RTOC_KKT_SCAL_H] += h;
  }
  if (st == 0 && a.dx0 && !a.floating && a.x0)
    for (int i = lane; i < nx; i += 64) a.dx0[(size_t)b * nx + i] = a.x0[(size_t)b * nx + i] - (i < nv ? q[i] : v[i - nv]);
  if (!a.floating) return;
  // ---- the free-flyer base: differences on SE(3) ----
  const M3 R = quat_R(q + 3), Rn = quat_R(qn + 3), Rp = quat_R(qp + 3);
  const V3 p = rbd::ldv3(q), pn = rbd::ldv3(qn), pp = rbd::ldv3(qp);
  M3 R1, R0;
  V3 p1, p0;
  rel(Rn, pn, R, p, R1, p1);  // X1 = M_next^-1 M
  rel(R, p, Rp, pp, R0, p0);  // X0 = M^-1 M_prev
  __shared__ double sLog[2][6];  // log6(X1), log6(X0)
  if (lane < 18) {
    // 18 lanes, one forward-mode evaluation through the log each: matrix m, column c
    //   m = 0: Fqq[:, c] = Jlog6(X1) e_c      m = 1: Fqq_prev[:, c] = -Jlog6(X0) Ad_{X0^-1} e_c      m = 2: d/dq0 of q (-) q_next
    const int m = lane / 6, c = lane % 6;
    const M3& Rm = m == 1 ? R0 : R1;
    const V3 pm = m == 1 ? p0 : p1;
    const SV e = unit_twist(c);
    const SV tw = m == 0 ? e : rbd::sv0() - rbd::act_inv(Rm, pm, e);
    SV val, der;
    rbd::log6_fwd(Rm, pm, tw, val, der);
    const double col[6] = {der.l.x, der.l.y, der.l.z, der.a.x, der.a.y, der.a.z};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      J[m][r + 6 * c] = col[r];
      if (m == 0) kr[a.o_fxx + r + (size_t)c * nx] = col[r];  // Fqq top-left corner
    }
    if (c == 0 && m < 2) {
      const double l6[6] = {val.l.x, val.l.y, val.l.z, val.a.x, val.a.y, val.a.z};
#pragma unroll
      for (int r = 0; r < 6; ++r) sLog[m][r] = l6[r];
      if (m == 0) {   // Fq (base) = log6(X1) + dt v
#pragma unroll
        for (int r = 0; r < 6; ++r) kr[a.o_fx + r] = l6[r] + dt * v[r];
      }
    }
  }
  __syncthreads();
  if (lane < 6) {
    // lq[:6] += Fqq^T lmd_next[:6] + Fqq_prev^T lmd[:6]   (:41-46)
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) t += J[0][r + 6 * lane] * lmdn[r] + J[1][r + 6 * lane] * lmd[r];
    kr[a.o_lx + lane] += t;
  }
  if (a.se3 && lane >= 32 && lane < 34) {
    // two lanes, one block-triangular inverse each: Fqq_inv (from the d/dq0 Jacobian), Fqq_prev_inv
    const int which = lane - 32;
    double* const se = a.se3 + rec * RTOC_SE3_STRIDE;
    double A[36];
    inv6_block_ut(J[which == 0 ? 2 : 1], A);
#pragma unroll
    for (int e = 0; e < 36; ++e) se[36 * which + e] = A[e];
    if (which == 1 && st == 0 && a.dx0 && a.x0) {
      // computeInitialStateDirection (:99-109): dq0 = q0 (-) s0.q = log6(M^-1 M0) = log6(X0) for the base, with the
      // -Fqq_prev_inv correction rtoc_compute_initial_state_direction would apply; joints and velocities plain
      double* const o = a.dx0 + (size_t)b * nx;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) t += A[r + 6 * c] * sLog[1][c];
        o[r] = -t;
      }
      const double* x0 = a.x0 + (size_t)b * (nq + nv);
      for (int i = 6; i < nv; ++i) o[i] = x0[1 + i] - q[1 + i];
      for (int i = 0; i < nv; ++i) o[nv + i] = x0[nq + i] - v[i];
    }
  }
}

}  // namespace rtoc

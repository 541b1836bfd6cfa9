// contact_constraints.hpp -- the friction-cone rows of the contact path evaluated on the device.
//
// FrictionCone / ImpactFrictionCone (reference src/constraints/friction_cone.cpp:100-191, impact_friction_cone.cpp) per
// ACTIVE contact of a grid point: the local contact force turned into the world frame, f_W = R_wf(q) f, the five rows
//     g = cone_local f_W,   cone_local = cone_world R_surface^T,   cone_world = [0 0 -1; +-1 0 -mu/sqrt2; 0 +-1 -mu/sqrt2]
// (frictionConeResidual, friction_cone.hpp:102-120) and their Jacobians
//     dg/df = cone_local R_wf                                           (:180-182)
//     dg/dq = cone_local (w_j x f_W),  w_j the angular Jacobian column of the contact frame -- in the LOCAL frame as the
//             reference takes it (getJacobianTransformFromLocalToWorld over getFrameJacobian(LOCAL), robot.hxx:247-287;
//             friction_cone.cpp:170-176), or world-aligned (the derivative of R_wf f): RTOC_OPT_CONE_JACOBIAN
//   INIT       setSlack + setSlackAndDualPositive: slack = -g clipped at sqrt(barrier), dual = barrier / slack  (:100-116, pdipm.hxx:12-23)
//   LINEARIZE  evalConstraint + evalDerivatives: residual = g + slack, cmpl = slack dual - barrier, the Jacobians into the
//              RTOC_BUF_CONE record (what rtoc_condense / rtoc_expand read), lq += dg/dq^T dual, lf += dg/df^T dual
// The rows' ConstraintComponentData live where rtoc_set_friction_cones puts them (RTOC_BUF_CON, compacted by active contact).
// One wave per (instance, grid point), lane j = dof j: the lane carries its column of the bodies' Jacobians down the tree.
#pragma once
#include "state_equation_lin.hpp"

namespace rtoc {

struct CcArgs {
  const rbd::DevModel* model;
  const double* sol;
  double* kkt;
  double* cdd;
  double* con;
  double* cone;
  const rtoc_grid* grid;
  const unsigned* active;
  const double* rotations;  // [nstages][ncontacts][9] or nullptr
  const double* mu;         // [ncontacts]
  int nstages, batch, nv, njoints, ncontacts, nlevels, mode;
  int contact_dim, row0, cone_stride, dgdf_off, impact_cones;
  int exact_jacobian;   // RTOC_OPT_CONE_JACOBIAN: 0 = w_local x f_W as the reference composes it, 1 = w_world x f_W
  double barrier;
  int sol_stride, kkt_stride, cdd_stride, con_stride;
  int o_q, o_f, o_lx, o_lf;
  rtoc_record_layout nl;
};
enum { CC_INIT = 0, CC_LINEARIZE = 1 };

__host__ __device__ constexpr size_t cc_lds_bytes(int nlevels, int njoints, int ncontacts) {
  return sizeof(double) * ((size_t)nlevels * (32 + 6 * 64) + njoints * rbd::JP + ncontacts * rbd::CP + (RTOC_MAX_JOINTS + 8));
}

static __global__ __launch_bounds__(64) void contact_cone_kernel(CcArgs a) {
  using namespace selin;
  using rbd::CP;
  using rbd::JP;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = blockIdx.x / nst1, st = blockIdx.x % nst1;   // the terminal grid point has no rows
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  if (g.dimf == 0 || (g.type == RTOC_GRID_IMPACT && !a.impact_cones)) return;
  const unsigned act = a.active[st];
  const int nv = a.nv, nb = a.njoints, ncon = a.ncontacts, nlev = a.nlevels, cd = a.contact_dim;
  double* const lval = smem;
  double* const ltan = lval + (size_t)nlev * 32;
  double* const sjm = ltan + (size_t)nlev * 6 * 64;
  double* const scm = sjm + nb * JP;
  double* const sq = scm + ncon * CP;
  const size_t rec = (size_t)b * a.nstages + st;
  const double* const s = a.sol + rec * a.sol_stride;
  double* const kr = a.kkt ? a.kkt + rec * a.kkt_stride : nullptr;
  double* const cr = a.cdd ? a.cdd + rec * a.cdd_stride : nullptr;
  double* const nr = a.con + rec * a.con_stride;
  double* const gr = a.cone ? a.cone + rec * a.cone_stride : nullptr;
  const int* const no = a.nl.off;
  {
    const double* const gj = &a.model->joint[0][0];
    const double* const gc = &a.model->contact[0][0];
    for (int e = lane; e < nb * JP; e += 64) sjm[e] = gj[e];
    for (int e = lane; e < ncon * CP; e += 64) scm[e] = gc[e];
    for (int e = lane; e < a.model->m.nq; e += 64) sq[e] = s[a.o_q + e];
  }
  __syncthreads();
  const int j = lane;
  const bool lane_on = j < nv;
  const bool lin = a.mode == CC_LINEARIZE;
  auto LV = [&](int lev, int k) -> double& { return lval[lev * 32 + k]; };
  auto LT = [&](int lev, int k) -> double& { return ltan[((size_t)lev * 6 + k) * 64 + lane]; };
  auto JM = [&](int i, int k) -> const double& { return sjm[i * JP + k]; };
  double lq = 0.0;
  for (int i = 0; i < nb; ++i) {
    const int d = (int)JM(i, 31), iq = (int)JM(i, 29), iv = (int)JM(i, 30);
    const bool ff = (int)JM(i, 28) == RTOC_JOINT_FREE_FLYER;
    const bool own = lane_on && j >= iv && j < iv + (ff ? 6 : 1);
    M3 Rj;
    V3 pj = rbd::mk(0, 0, 0);
    if (ff) {
      Rj = quat_R(sq + iq + 3);
      pj = rbd::ldv3(sq + iq);
    } else {
      const V3 ax = rbd::ldv3(&JM(i, 12));
      const double th = sq[iq], c = cos(th), sn = sin(th), t = 1.0 - c;
      Rj.m[0] = t * ax.x * ax.x + c, Rj.m[1] = t * ax.x * ax.y - sn * ax.z, Rj.m[2] = t * ax.x * ax.z + sn * ax.y;
      Rj.m[3] = t * ax.x * ax.y + sn * ax.z, Rj.m[4] = t * ax.y * ax.y + c, Rj.m[5] = t * ax.y * ax.z - sn * ax.x;
      Rj.m[6] = t * ax.x * ax.z - sn * ax.y, Rj.m[7] = t * ax.y * ax.z + sn * ax.x, Rj.m[8] = t * ax.z * ax.z + c;
    }
    const M3 Rp = rbd::ldm3(&JM(i, 0));
    const M3 R = rbd::mul(Rp, Rj);
    const V3 p = rbd::mul(Rp, pj) + rbd::ldv3(&JM(i, 9));
    M3 oR = R;
    V3 wj = rbd::mk(0, 0, 0);   // body-frame angular velocity of body i per unit rate of dof j
    if (d > 0) {
      oR = rbd::mul(rbd::ldm3(&LV(d - 1, 12)), R);
      wj = rbd::mulT(R, rbd::mk(LT(d - 1, 3), LT(d - 1, 4), LT(d - 1, 5)));
    }
    if (own) wj = wj + (ff ? unit_twist(j - iv).a : rbd::ldv3(&JM(i, 12)));
#pragma unroll
    for (int k = 0; k < 9; ++k) LV(d, 12 + k) = oR.m[k];
    LT(d, 3) = wj.x, LT(d, 4) = wj.y, LT(d, 5) = wj.z;
    int k = 0;  // index among the active contacts
    for (int c = 0; c < ncon; ++c) {
      const bool on = (act >> c) & 1u;
      if (on && (int)scm[c * CP + 14] == i) {
        const M3 Rwf = rbd::mul(oR, rbd::ldm3(&scm[c * CP]));
        const V3 fW = rbd::mul(Rwf, rbd::ldv3(s + a.o_f + k * cd));
        M3 Rs;
#pragma unroll
        for (int e = 0; e < 9; ++e) Rs.m[e] = a.rotations ? a.rotations[((size_t)st * ncon + c) * 9 + e] : ((e % 4 == 0) ? 1.0 : 0.0);
        const double m = a.mu[c] * 0.70710678118654752440;
        // rows of cone_local = cone_world R_surface^T: row r = R_surface * (row r of cone_world)
        V3 row[5];
        row[0] = rbd::mul(Rs, rbd::mk(0, 0, -1));
        row[1] = rbd::mul(Rs, rbd::mk(1, 0, -m));
        row[2] = rbd::mul(Rs, rbd::mk(-1, 0, -m));
        row[3] = rbd::mul(Rs, rbd::mk(0, 1, -m));
        row[4] = rbd::mul(Rs, rbd::mk(0, -1, -m));
        const int r0 = a.row0 + 5 * k;
        if (lane < 5) {
          double gval = 0.0;   // row `lane` by selection (no run-time index into row[])
#pragma unroll
          for (int r = 0; r < 5; ++r)
            if (lane == r) gval = rbd::dot(row[r], fW);
          if (!lin) {
            double slack = -gval;
            const double sb = sqrt(a.barrier);
            if (slack < sb) slack = sb;
            nr[no[RTOC_CON_SLACK] + r0 + lane] = slack;
            nr[no[RTOC_CON_DUAL] + r0 + lane] = a.barrier / slack;
          } else {
            const double slack = nr[no[RTOC_CON_SLACK] + r0 + lane], dual = nr[no[RTOC_CON_DUAL] + r0 + lane];
            nr[no[RTOC_CON_RESIDUAL] + r0 + lane] = gval + slack;
            nr[no[RTOC_CON_CMPL] + r0 + lane] = slack * dual - a.barrier;
          }
        }
        if (lin) {
          double dual[5];
#pragma unroll
          for (int r = 0; r < 5; ++r) dual[r] = nr[no[RTOC_CON_DUAL] + r0 + r];
          if (lane < 3) {   // column `lane` of dg/df = cone_local R_wf, and lf
            const V3 col = rbd::mk(Rwf.m[lane], Rwf.m[3 + lane], Rwf.m[6 + lane]);
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
              const double e = rbd::dot(row[r], col);
              gr[a.dgdf_off + k * 15 + r + 5 * lane] = e;
              acc += e * dual[r];
            }
            cr[a.o_lf + k * cd + lane] += acc;
          }
          if (lane_on) {    // column j of dg/dq
            // the reference crosses the LOCAL-frame angular Jacobian column with the WORLD-frame force (robot.hxx:247-253,
            // 275-287: getFrameJacobian(..., pinocchio::LOCAL, ...)); the derivative of R_wf f is w_world x f_W
            const V3 wxf = rbd::cross(a.exact_jacobian ? rbd::mul(oR, wj) : rbd::mulT(rbd::ldm3(&scm[c * CP]), wj), fW);
#pragma unroll
            for (int r = 0; r < 5; ++r) {
              const double e = rbd::dot(row[r], wxf);
              gr[k * 5 * nv + r + 5 * j] = e;
              lq += e * dual[r];
            }
          }
        }
      }
      k += on ? 1 : 0;
    }
  }
  if (lin && lane_on) kr[a.o_lx + j] += lq;
}

}  // namespace rtoc

namespace rtoc {

// The same rows (LINEARIZE) from the kinematics the rigid-body pre-pass has already computed (rbd_values_kernel: the world
// rotation of every body, slots 12..20 of its block): no tree walk.  The world-aligned angular Jacobian column of a contact
// frame for dof j is oR_body(j) axis_j if the dof lies on the path from the root to the contact's body, else zero.
// One wave per (instance, grid point), lane j = dof j.
struct CvArgs {
  CcArgs c;
  const double* vals;   // [batch * nstages][njoints][64]
};

static __global__ __launch_bounds__(64) void contact_cone_vals_kernel(CvArgs v) {
  using namespace selin;
  using rbd::CP;
  const CcArgs& a = v.c;
  const int lane = threadIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = blockIdx.x / nst1, st = blockIdx.x % nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  if (g.dimf == 0 || (g.type == RTOC_GRID_IMPACT && !a.impact_cones)) return;
  const unsigned act = a.active[st];
  const int nv = a.nv, nb = a.njoints, ncon = a.ncontacts, cd = a.contact_dim;
  const size_t rec = (size_t)b * a.nstages + st;
  const double* const s = a.sol + rec * a.sol_stride;
  double* const kr = a.kkt + rec * a.kkt_stride;
  double* const cr = a.cdd + rec * a.cdd_stride;
  double* const nr = a.con + rec * a.con_stride;
  double* const gr = a.cone + rec * a.cone_stride;
  const double* const vb = v.vals + rec * (size_t)nb * rbd::VAL_SLOTS;
  const int* const no = a.nl.off;
  const int j = lane;
  const bool lane_on = j < nv;
  // this dof's rotation axis in the world frame
  V3 wj = rbd::mk(0, 0, 0);
  if (lane_on) {
    const double* const blk = vb + (size_t)a.model->dof_body[j] * rbd::VAL_SLOTS;
    wj = rbd::mul(rbd::ldm3(blk + 12), rbd::ldv3(a.model->dof_axis[j]));
  }
  double lq = 0.0;
  int k = 0;
  for (int c = 0; c < ncon; ++c) {
    if (!((act >> c) & 1u)) continue;
    const double* const cm = &a.model->contact[c][0];
    const double* const blk = vb + (size_t)(int)cm[14] * rbd::VAL_SLOTS;
    const M3 Rwf = rbd::mul(rbd::ldm3(blk + 12), rbd::ldm3(cm));
    const V3 fW = rbd::mul(Rwf, rbd::ldv3(s + a.o_f + k * cd));
    M3 Rs;
#pragma unroll
    for (int e = 0; e < 9; ++e) Rs.m[e] = a.rotations ? a.rotations[((size_t)st * ncon + c) * 9 + e] : ((e % 4 == 0) ? 1.0 : 0.0);
    const double m = a.mu[c] * 0.70710678118654752440;
    V3 row[5];
    row[0] = rbd::mul(Rs, rbd::mk(0, 0, -1));
    row[1] = rbd::mul(Rs, rbd::mk(1, 0, -m));
    row[2] = rbd::mul(Rs, rbd::mk(-1, 0, -m));
    row[3] = rbd::mul(Rs, rbd::mk(0, 1, -m));
    row[4] = rbd::mul(Rs, rbd::mk(0, -1, -m));
    const int r0 = a.row0 + 5 * k;
    double dual[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) dual[r] = nr[no[RTOC_CON_DUAL] + r0 + r];
    if (lane < 5) {
      double gval = 0.0, dl = 0.0;   // row `lane` by selection: a run-time index would put row[] and dual[] into scratch
#pragma unroll
      for (int r = 0; r < 5; ++r)
        if (lane == r) gval = rbd::dot(row[r], fW), dl = dual[r];
      const double slack = nr[no[RTOC_CON_SLACK] + r0 + lane];
      nr[no[RTOC_CON_RESIDUAL] + r0 + lane] = gval + slack;
      nr[no[RTOC_CON_CMPL] + r0 + lane] = slack * dl - a.barrier;
    }
    if (lane < 3) {
      const V3 col = rbd::mk(Rwf.m[lane], Rwf.m[3 + lane], Rwf.m[6 + lane]);
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const double e = rbd::dot(row[r], col);
        gr[a.dgdf_off + k * 15 + r + 5 * lane] = e;
        acc += e * dual[r];
      }
      cr[a.o_lf + k * cd + lane] += acc;
    }
    if (lane_on) {
      const bool path = (a.model->contact_dofs[c] >> j) & 1ull;
      // reference: LOCAL-frame angular Jacobian column x WORLD-frame force (see contact_cone_kernel); exact: w_world x f_W
      const V3 wxf = path ? rbd::cross(a.exact_jacobian ? wj : rbd::mulT(Rwf, wj), fW) : rbd::mk(0, 0, 0);
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const double e = rbd::dot(row[r], wxf);
        gr[k * 5 * nv + r + 5 * j] = e;
        lq += e * dual[r];
      }
    }
    ++k;
  }
  if (lane_on) kr[a.o_lx + j] += lq;
}

}  // namespace rtoc

namespace rtoc {

// ContactWrenchCone / ImpactWrenchCone (reference src/constraints/contact_wrench_cone.cpp:114-204): 17 rows per ACTIVE surface
// contact, g = cone f with the constant 17 x 6 cone matrix of the sole (computeCone / updateCone, :282-313; the host builds
// it with rtoc_wrench_cone_matrix and hands a table over) acting on the local 6-d contact wrench -- no kinematics:
//   INIT       cone matrix into the RTOC_BUF_CONE record (what rtoc_condense / rtoc_expand read), slack = -g clipped at
//              sqrt(barrier), dual = barrier / slack
//   LINEARIZE  residual = g + slack, cmpl = slack dual - barrier, lf += cone^T dual
// One wave per (instance, grid point), lane r < 17 = row r of a contact.
struct WcArgs {
  const double* sol;
  double* cdd;
  double* con;
  double* cone;
  const double* table;   // [ncontacts][17 x 6], column-major (ld 17)
  const rtoc_grid* grid;
  const unsigned* active;
  int nstages, batch, ncontacts, mode, row0, cone_stride, impact_cones;
  double barrier;
  int sol_stride, cdd_stride, con_stride;
  int o_f, o_lf;
  rtoc_record_layout nl;
};

static __global__ __launch_bounds__(64) void wrench_cone_eval_kernel(WcArgs a) {
  __shared__ double sd[RTOC_WRENCH_ROWS];
  const int lane = threadIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = blockIdx.x / nst1, st = blockIdx.x % nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  if (g.dimf == 0 || (g.type == RTOC_GRID_IMPACT && !a.impact_cones)) return;
  const unsigned act = a.active[st];
  const size_t rec = (size_t)b * a.nstages + st;
  const double* const s = a.sol + rec * a.sol_stride;
  double* const cr = a.cdd ? a.cdd + rec * a.cdd_stride : nullptr;
  double* const nr = a.con + rec * a.con_stride;
  double* const gr = a.cone + rec * a.cone_stride;
  const int* const no = a.nl.off;
  int k = 0;
  for (int c = 0; c < a.ncontacts; ++c) {
    if (!((act >> c) & 1u)) continue;
    const double* const A = a.table + (size_t)c * RTOC_WRENCH_ROWS * 6;
    const int r0 = a.row0 + RTOC_WRENCH_ROWS * k;
    if (a.mode == CC_INIT)
      for (int e = lane; e < RTOC_WRENCH_ROWS * 6; e += 64) gr[k * RTOC_WRENCH_ROWS * 6 + e] = A[e];
    if (lane < RTOC_WRENCH_ROWS) {
      double gval = 0.0;
#pragma unroll
      for (int t = 0; t < 6; ++t) gval += A[lane + RTOC_WRENCH_ROWS * t] * s[a.o_f + 6 * k + t];
      if (a.mode == CC_INIT) {
        double slack = -gval;
        const double sb = sqrt(a.barrier);
        if (slack < sb) slack = sb;
        nr[no[RTOC_CON_SLACK] + r0 + lane] = slack;
        nr[no[RTOC_CON_DUAL] + r0 + lane] = a.barrier / slack;
      } else {
        const double slack = nr[no[RTOC_CON_SLACK] + r0 + lane], dual = nr[no[RTOC_CON_DUAL] + r0 + lane];
        nr[no[RTOC_CON_RESIDUAL] + r0 + lane] = gval + slack;
        nr[no[RTOC_CON_CMPL] + r0 + lane] = slack * dual - a.barrier;
        sd[lane] = dual;
      }
    }
    if (a.mode == CC_LINEARIZE) {
      __syncthreads();
      if (lane < 6) {
        double acc = 0.0;
        for (int r = 0; r < RTOC_WRENCH_ROWS; ++r) acc += A[r + RTOC_WRENCH_ROWS * lane] * sd[r];
        cr[a.o_lf + 6 * k + lane] += acc;
      }
      __syncthreads();
    }
    ++k;
  }
}

}  // namespace rtoc

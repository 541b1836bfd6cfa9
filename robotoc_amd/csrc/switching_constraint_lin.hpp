// switching_constraint_lin.hpp -- linearizeSwitchingConstraint on the device.
//
// Replaces linearizeSwitchingConstraint (reference src/dynamics/switching_constraint.cpp:26-70) on the grid point two ahead
// of an impact (GridInfo::switching_constraint): the contacts that become active at the impact must sit at their desired
// positions at the configuration predicted two steps ahead,
//     q+ = q (+) dq,   dq = (dt1 + dt2) v + dt1 dt2 a,   P = position(q+) - desired          (:14-23)
//     (surface contacts: P = Log6(X_desired^-1 X_frame(q+)), six rows, Pq = Jlog6 J_frame,local: surface_contact.hxx:106-128)
//     Pq = R_of J_frame,lin at q+ (point_contact.hxx:133-142),  Phiq = Pq dIntegrate_dq(q, dq),  Phiv = (dt1+dt2) Pq dIntegrate_dv(q, dq),
//     Phia = dt1 dt2 Pq dIntegrate_dv(q, dq)                                                  (:41-51)
// with Pinocchio's dIntegrate on the free-flyer base restated: dIntegrate_dq = Ad_{exp(dq_b)}^-1 (the motion actInv),
// dIntegrate_dv = Jexp6(dq_b) = Jlog6(exp6(dq_b))^-1 (six forward-mode evaluations through the log, one 6 x 6 inverse);
// identities on the joints.  Then the multiplier terms (:52-53) and the STO sensitivities (:54-62).
// Base-block transport: see RTOC_OPT_SWITCHING_TRANSPORT (rtoc.h) -- the reference's composition by default.
// Mapping: one wave per (instance, switching grid point), one lane per dof: the lane carries its column of every body's
// Jacobian along the depth-first walk (body-frame twist per unit rate of the dof), values once per tree level in LDS.
#pragma once
#include "state_equation_lin.hpp"

namespace rtoc {

struct SwLinArgs {
  const rbd::DevModel* model;
  const double* sol;
  double* kkt;
  double* cdd;
  const rtoc_grid* grid;
  const unsigned* active;
  const double* positions;  // [nstages][ncontacts][3] or nullptr
  const double* rotations;  // [nstages][ncontacts][9] or nullptr (surface contacts)
  int nstages, batch, nv, nq, njoints, ncontacts, nlevels, floating, ns_max;
  int exact_transport;  // RTOC_OPT_SWITCHING_TRANSPORT
  int nsel, sel[16];    // the grid points with a switching constraint (nsel == 0: all grid points are launched)
  int sol_stride, kkt_stride, cdd_stride;
  int o_q, o_v, o_a, o_xi;
  int o_phix, o_phit, o_pres, o_lx, o_hx, o_scal;
  int o_phia, o_la, o_ha;
  const double* dt_inst;  // per-instance time steps or nullptr (grid_dt)
};

__host__ __device__ constexpr size_t sw_lds_bytes(int nlevels, int njoints, int ncontacts) {
  return sizeof(double) * ((size_t)nlevels * (32 + 6 * 64) + njoints * rbd::JP + ncontacts * rbd::CP + 3 * (RTOC_MAX_JOINTS + 8) +
                           6 * RTOC_MAX_CONTACTS * 8 + 2 * 36 + 6 * RTOC_MAX_CONTACTS * (RTOC_MAX_JOINTS + 8));
}

static __global__ __launch_bounds__(64) void switching_constraint_lin_kernel(SwLinArgs a) {
  using namespace selin;
  using rbd::CP;
  using rbd::JP;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  // launched over the grid points that carry a switching constraint only (sel), or over all of them (nsel == 0)
  const int nst1 = a.nsel > 0 ? a.nsel : a.nstages - 1;
  const int b = blockIdx.x / nst1, st = a.nsel > 0 ? a.sel[blockIdx.x % nst1] : blockIdx.x % nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  if (!g.switching_constraint || st + 2 >= a.nstages) return;
  const int nv = a.nv, nx = 2 * nv, nb = a.njoints, ncon = a.ncontacts, nlev = a.nlevels, ns = g.dims, LDSW = a.ns_max;
  const double dt1 = grid_dt(a.grid, a.dt_inst, b, a.nstages, st), dt2 = grid_dt(a.grid, a.dt_inst, b, a.nstages, st + 1);
  const unsigned impact = a.active[st + 2];  // ImpactStatus of the impact two grid points ahead
  double* const lval = smem;                               // [nlev][32]: R 9, p 3, oR 9, op 3, body index
  double* const ltan = lval + (size_t)nlev * 32;           // [nlev][6][64]
  double* const sjm = ltan + (size_t)nlev * 6 * 64;
  double* const scm = sjm + nb * JP;
  double* const sqp = scm + ncon * CP;                     // q+
  double* const sdq = sqp + RTOC_MAX_JOINTS + 8;           // dq
  double* const sxi = sdq + RTOC_MAX_JOINTS + 8;
  double* const spq = sxi + RTOC_MAX_JOINTS + 8;           // Pq base block [rows][8], then P residual etc.
  double* const sT = spq + 6 * RTOC_MAX_CONTACTS * 8;      // Tq (36) | Jr (36), column-major
  double* const sPq = sT + 72;                             // Pq [3 ncontacts][nv] row-major
  const size_t rec = (size_t)b * a.nstages + st;
  const double* const s = a.sol + rec * a.sol_stride;
  double* const kr = a.kkt + rec * a.kkt_stride;
  double* const cr = a.cdd + rec * a.cdd_stride;
  {
    const double* const gj = &a.model->joint[0][0];
    const double* const gc = &a.model->contact[0][0];
    for (int e = lane; e < nb * JP; e += 64) sjm[e] = gj[e];
    for (int e = lane; e < ncon * CP; e += 64) scm[e] = gc[e];
  }
  for (int i = lane; i < nv; i += 64) sdq[i] = (dt1 + dt2) * s[a.o_v + i] + dt1 * dt2 * s[a.o_a + i];   // (:19)
  for (int i = lane; i < ns; i += 64) sxi[i] = s[a.o_xi + i];
  __syncthreads();
  // ---- q+ = q (+) dq (:20): joints additively, the base by the SE(3) exponential; X = exp6(dq_b) for the transports ----
  const int nbase = a.floating ? 6 : 0;
  for (int i = lane; i < nv - nbase; i += 64) sqp[(nbase ? 7 : 0) + i] = s[a.o_q + (nbase ? 7 : 0) + i] + sdq[nbase + i];
  M3 E;     // rotation of exp6(dq_b)
  V3 pe = rbd::mk(0, 0, 0);
#pragma unroll
  for (int e = 0; e < 9; ++e) E.m[e] = (e % 4 == 0) ? 1.0 : 0.0;
  M3 Rb;    // base rotation at q+
  V3 pb = rbd::mk(0, 0, 0);
  if (a.floating) {
    const V3 vl = rbd::mk(sdq[0], sdq[1], sdq[2]), w = rbd::mk(sdq[3], sdq[4], sdq[5]);
    const double th = sqrt(rbd::dot(w, w));
    double A, B;
    if (th < 1e-8) {
      A = 0.5, B = 1.0 / 6.0;
    } else {
      A = (1.0 - cos(th)) / (th * th), B = (th - sin(th)) / (th * th * th);
    }
    const V3 wxv = rbd::cross(w, vl);
    pe = vl + A * wxv + B * rbd::cross(w, wxv);
    // exp(w) = I + sin t / t [w]x + (1 - cos t) / t^2 [w]x^2
    const double sa = th < 1e-8 ? 1.0 - th * th / 6.0 : sin(th) / th;
    const double wx[9] = {0, -w.z, w.y, w.z, 0, -w.x, -w.y, w.x, 0};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double w2 = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) w2 += wx[3 * r + k] * wx[3 * k + c];
        E.m[3 * r + c] = (r == c ? 1.0 : 0.0) + sa * wx[3 * r + c] + A * w2;
      }
    const M3 R0 = quat_R(s + a.o_q + 3);
    Rb = rbd::mul(R0, E);
    pb = rbd::ldv3(s + a.o_q) + rbd::mul(R0, pe);
  }
  __syncthreads();
  // ---- depth-first walk at q+: values per level, this lane's Jacobian column (body-frame twist per unit rate of dof j) ----
  const int j = lane;
  const bool lane_on = j < nv;
  auto LV = [&](int lev, int k) -> double& { return lval[lev * 32 + k]; };
  auto LT = [&](int lev, int k) -> double& { return ltan[((size_t)lev * 6 + k) * 64 + lane]; };
  auto JM = [&](int i, int k) -> const double& { return sjm[i * JP + k]; };
  for (int i = 0; i < nb; ++i) {
    const int d = (int)JM(i, 31), iq = (int)JM(i, 29), iv = (int)JM(i, 30);
    const bool ff = (int)JM(i, 28) == RTOC_JOINT_FREE_FLYER;
    const bool own = lane_on && j >= iv && j < iv + (ff ? 6 : 1);
    M3 Rj;
    V3 pj = rbd::mk(0, 0, 0);
    if (ff) {
      Rj = Rb;
      pj = pb;
    } else {
      const V3 ax = rbd::ldv3(&JM(i, 12));
      const double th = sqp[iq], c = cos(th), sn = sin(th), t = 1.0 - c;
      Rj.m[0] = t * ax.x * ax.x + c, Rj.m[1] = t * ax.x * ax.y - sn * ax.z, Rj.m[2] = t * ax.x * ax.z + sn * ax.y;
      Rj.m[3] = t * ax.x * ax.y + sn * ax.z, Rj.m[4] = t * ax.y * ax.y + c, Rj.m[5] = t * ax.y * ax.z - sn * ax.x;
      Rj.m[6] = t * ax.x * ax.z - sn * ax.y, Rj.m[7] = t * ax.y * ax.z + sn * ax.x, Rj.m[8] = t * ax.z * ax.z + c;
    }
    const M3 Rp = rbd::ldm3(&JM(i, 0));
    const M3 R = rbd::mul(Rp, Rj);
    const V3 p = rbd::mul(Rp, pj) + rbd::ldv3(&JM(i, 9));
    M3 oR = R;
    V3 op = p;
    SV Jc = rbd::sv0();
    if (d > 0) {
      const M3 oRp = rbd::ldm3(&LV(d - 1, 12));
      oR = rbd::mul(oRp, R);
      op = rbd::ldv3(&LV(d - 1, 21)) + rbd::mul(oRp, p);
      Jc = rbd::act_inv(R, p, SV{rbd::mk(LT(d - 1, 0), LT(d - 1, 1), LT(d - 1, 2)), rbd::mk(LT(d - 1, 3), LT(d - 1, 4), LT(d - 1, 5))});
    }
    if (own) Jc = Jc + (ff ? unit_twist(j - iv) : SV{rbd::mk(0, 0, 0), rbd::ldv3(&JM(i, 12))});
#pragma unroll
    for (int k = 0; k < 9; ++k) LV(d, k) = R.m[k], LV(d, 12 + k) = oR.m[k];
    LV(d, 9) = p.x, LV(d, 10) = p.y, LV(d, 11) = p.z, LV(d, 21) = op.x, LV(d, 22) = op.y, LV(d, 23) = op.z;
    LT(d, 0) = Jc.l.x, LT(d, 1) = Jc.l.y, LT(d, 2) = Jc.l.z, LT(d, 3) = Jc.a.x, LT(d, 4) = Jc.a.y, LT(d, 5) = Jc.a.z;
    // impacting contacts carried by this body: P rows and this lane's column of Pq
    int r0 = 0;
    for (int c = 0; c < ncon; ++c) {
      const bool on = (impact >> c) & 1u;
      const bool surf = (int)scm[c * CP + 15] == RTOC_CONTACT_SURFACE;
      if (on && (int)scm[c * CP + 14] == i) {
        const M3 Rf = rbd::ldm3(&scm[c * CP]);
        const V3 pf = rbd::ldv3(&scm[c * CP + 9]);
        const M3 oRf = rbd::mul(oR, Rf);
        const V3 pw = op + rbd::mul(oR, pf);
        const V3 pr = a.positions ? rbd::ldv3(a.positions + ((size_t)(st + 2) * ncon + c) * 3) : rbd::mk(0, 0, 0);
        const SV jf = rbd::act_inv(Rf, pf, Jc);   // this lane's column of the LOCAL frame Jacobian
        double* const Pres = spq + 6 * RTOC_MAX_CONTACTS * 4;
        if (!surf) {
          const V3 col = rbd::mul(oRf, jf.l);     // R_of J_frame,lin (point_contact.hxx:133-142)
          if (lane == 0) Pres[r0] = pw.x - pr.x, Pres[r0 + 1] = pw.y - pr.y, Pres[r0 + 2] = pw.z - pr.z;
          if (lane_on) sPq[(r0 + 0) * nv + j] = col.x, sPq[(r0 + 1) * nv + j] = col.y, sPq[(r0 + 2) * nv + j] = col.z;
        } else {
          // P = Log6(X_desired^-1 X_frame), Pq = Jlog6(X_diff) J_frame,local (surface_contact.hxx:106-128): forward mode
          M3 Rdt;   // transpose of the desired rotation (identity if none was set)
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
              Rdt.m[3 * r + cc] = a.rotations ? a.rotations[((size_t)(st + 2) * ncon + c) * 9 + 3 * cc + r] : (r == cc ? 1.0 : 0.0);
          SV lg, dlg;
          rbd::log6_fwd(rbd::mul(Rdt, oRf), rbd::mul(Rdt, pw - pr), jf, lg, dlg);
          const double pv[6] = {lg.l.x, lg.l.y, lg.l.z, lg.a.x, lg.a.y, lg.a.z}, dv6[6] = {dlg.l.x, dlg.l.y, dlg.l.z, dlg.a.x, dlg.a.y, dlg.a.z};
#pragma unroll
          for (int t = 0; t < 6; ++t) {
            if (lane == 0) Pres[r0 + t] = pv[t];
            if (lane_on) sPq[(r0 + t) * nv + j] = dv6[t];
          }
        }
      }
      r0 += on ? (surf ? 6 : 3) : 0;
    }
  }
  // ---- transports of the base block: Tq = Ad_{exp(dq_b)}^-1, Jr = Jlog6(exp6(dq_b))^-1 ----
  if (a.floating && lane < 6) {
    const SV t = rbd::act_inv(E, pe, unit_twist(lane));
    const double tc[6] = {t.l.x, t.l.y, t.l.z, t.a.x, t.a.y, t.a.z};
    SV val, der;
    rbd::log6_fwd(E, pe, unit_twist(lane), val, der);
    const double jc[6] = {der.l.x, der.l.y, der.l.z, der.a.x, der.a.y, der.a.z};
#pragma unroll
    for (int r = 0; r < 6; ++r) sT[r + 6 * lane] = tc[r], sT[36 + r + 6 * lane] = jc[r];
  }
  __syncthreads();
  if (a.floating && lane == 0) {   // Jlog6 is block upper-triangular in the (linear, angular) ordering: 3 x 3 cofactor inverses
    double A[36];
    inv6_block_ut(sT + 36, A);
#pragma unroll
    for (int e = 0; e < 36; ++e) sT[36 + e] = A[e];
  }
  __syncthreads();
  // ---- Phiq, Phiv, Phia, P, Phit and the multiplier / STO terms ----
  const double* const P = spq + 6 * RTOC_MAX_CONTACTS * 4;
  if (lane_on) {
    double lq = 0.0, lvv = 0.0, laa = 0.0, pqxi = 0.0;
    for (int r = 0; r < ns; ++r) {
      double pq = sPq[r * nv + j], pqq = pq, pqv = pq;
      if (a.floating && j < 6) {
        pqq = 0.0, pqv = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          // the reference hands pinocchio::dIntegrateTransport the TRANSPOSED Jacobian (robot.hxx:69-72, :88-91), which
          // left-multiplies by dIntegrate: Phi^T = dIntegrate Pq^T, i.e. Phi = Pq dIntegrate^T; the chain rule is Pq dIntegrate
          const int e = a.exact_transport ? k + 6 * j : j + 6 * k;
          pqq += sPq[r * nv + k] * sT[e], pqv += sPq[r * nv + k] * sT[36 + e];
        }
      }
      kr[a.o_phix + r + (size_t)j * LDSW] = pqq;                           // Phiq
      kr[a.o_phix + r + (size_t)(nv + j) * LDSW] = (dt1 + dt2) * pqv;      // Phiv (:45-46)
      cr[a.o_phia + r + (size_t)j * LDSW] = dt1 * dt2 * pqv;               // Phia
      lq += pqq * sxi[r], lvv += (dt1 + dt2) * pqv * sxi[r], laa += dt1 * dt2 * pqv * sxi[r], pqxi += pq * sxi[r];
    }
    kr[a.o_lx + j] += lq;                 // lx += Phix^T xi (:52)
    kr[a.o_lx + nv + j] += lvv;
    cr[a.o_la + j] += laa;                // la += Phia^T xi (:53)
    kr[a.o_hx + nv + j] += 2.0 * pqxi;    // hv += 2 Pq^T xi (:61)
    cr[a.o_ha + j] += 2.0 * dt1 * pqxi;   // ha += 2 dt1 Pq^T xi (:62)
  }
  // Phit = Pq (2 (v + dt1 a)) (:56-57); h += xi . Phit (:58); Qtt += 2 (Pq^T xi) . a (:60)
  double hacc = 0.0, qacc = 0.0;
  for (int r = lane; r < ns; r += 64) {
    double t = 0.0, ta = 0.0;
    for (int k = 0; k < nv; ++k) t += sPq[r * nv + k] * 2.0 * (s[a.o_v + k] + dt1 * s[a.o_a + k]), ta += sPq[r * nv + k] * s[a.o_a + k];
    kr[a.o_phit + r] = t;
    kr[a.o_pres + r] = P[r];
    hacc += sxi[r] * t;
    qacc += 2.0 * sxi[r] * ta;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) hacc += __shfl_xor(hacc, off, 64), qacc += __shfl_xor(qacc, off, 64);
  if (lane == 0) {
    kr[a.o_scal + RTOC_KKT_SCAL_H] += hacc;
    kr[a.o_scal + RTOC_KKT_SCAL_QTT] += qacc;
  }
}

}  // namespace rtoc

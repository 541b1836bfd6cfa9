// friction_cone.hpp -- PDIPM rows of the linearised friction cones (5 rows per active point contact).
//
// Replaces FrictionCone::condenseSlackAndDual / expandSlackAndDual (reference
// src/constraints/friction_cone.cpp:194-268; ImpactFrictionCone has the same algebra on the impulse)
// and their share of updateSlack / updateDual (constraints_impl.hxx:167-182).  Unlike the joint
// limits these rows have dense Jacobians: dg_dq (5 x nv) and dg_df (5 x 3) per contact, computed by
// the CPU-side evalDerivatives (:143-191) and handed over in the RTOC_BUF_CONE record, compacted
// over the ACTIVE contacts of the grid point (k-th active contact -> block k, force offset
// k * contact_dim).  Their slack/dual/residual/... live in the tail of the RTOC_BUF_CON record:
// rows nc_max - 5*max_contacts + 5k + j.
//
// Condensation adds to Qqq, Qqf, Qff, lq, lf BEFORE the contact-dynamics condensation consumes
// them (Constraints::condenseSlackAndDual runs first, intermediate_stage.cpp:134-136).
// A few kflop per grid point, read-modify-write of a 18x18 block: one wave per grid point.
#pragma once
#include "device_utils.hpp"
#include "../../include/rtoc.h"

namespace rtoc {

struct ConeArgs {
  double* kkt;
  double* cdd;
  double* con;
  const double* cone;
  const double* dir;
  const rtoc_grid* grid;
  unsigned long long* steps;  // [batch][2] bit patterns (expand) / doubles (update)
  int nstages, batch;
  int max_contacts, contact_dim, row0;
  int cone_stride, dgdf_off;
  double tau;
  rtoc_record_layout kl, cl, nl, dl;
};

template <int NV, int NF>
__global__ __launch_bounds__(64) void cone_condense_kernel(ConeArgs a) {
  constexpr int NX = 2 * NV, NFP = NF > 0 ? NF : 1;
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  const int nact = g.dimf / a.contact_dim;
  if (nact == 0) return;
  const size_t rec = (size_t)b * a.nstages + st;
  double* kr = a.kkt + rec * a.kl.stride;
  double* cr = a.cdd + rec * a.cl.stride;
  double* nr = a.con + rec * a.nl.stride;
  const double* cone = a.cone + rec * a.cone_stride;
  double* Qxx = kr + a.kl.off[RTOC_KKT_QXX];
  double* lx = kr + a.kl.off[RTOC_KKT_LX];
  double* Qff = cr + a.cl.off[RTOC_CDD_QFF];
  double* Qqf = cr + a.cl.off[RTOC_CDD_QQF];
  double* lf = cr + a.cl.off[RTOC_CDD_LF];
  __shared__ double dq[5 * NV], df[15], cond[5], rr[5];
  for (int k = 0; k < nact; ++k) {  // contacts in order, like the reference's loop (:199-233)
    const int r0 = a.row0 + 5 * k, stack = k * a.contact_dim;
    for (int e = lane; e < 5 * NV; e += 64) dq[e] = cone[(size_t)k * 5 * NV + e];
    if (lane < 15) df[lane] = cone[a.dgdf_off + k * 15 + lane];
    if (lane < 5) {
      const double slack = nr[a.nl.off[RTOC_CON_SLACK] + r0 + lane], dual = nr[a.nl.off[RTOC_CON_DUAL] + r0 + lane];
      const double c = (dual * nr[a.nl.off[RTOC_CON_RESIDUAL] + r0 + lane] - nr[a.nl.off[RTOC_CON_CMPL] + r0 + lane]) / slack;
      nr[a.nl.off[RTOC_CON_COND] + r0 + lane] = c;  // computeCondensingCoeffcient<5> (:202)
      cond[lane] = c;
      rr[lane] = dual / slack;  // (:211-212)
    }
    __syncthreads();
    // lq += dg_dq^T cond ; lf += dg_df^T cond (:206-208)
    if (lane < NV) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 5; ++j) acc += dq[j + 5 * lane] * cond[j];
      lx[lane] += acc;
    } else if (lane < NV + 3) {
      const int m = lane - NV;
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 5; ++j) acc += df[j + 5 * m] * cond[j];
      lf[stack + m] += acc;
    }
    // Qqq += dg_dq^T (r dg_dq) (:215-216)
    for (int e = lane; e < NV * NV; e += 64) {
      const int r = e % NV, c = e / NV;
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 5; ++j) acc += dq[j + 5 * r] * (rr[j] * dq[j + 5 * c]);
      Qxx[r + (size_t)c * NX] += acc;
    }
    // Qqf[:, stack..+3] += dg_dq^T (r dg_df) (:217-218) ; Qff block += dg_df^T (r dg_df) (:219-220)
    for (int e = lane; e < NV * 3 + 9; e += 64) {
      if (e < NV * 3) {
        const int r = e % NV, m = e / NV;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 5; ++j) acc += dq[j + 5 * r] * (rr[j] * df[j + 5 * m]);
        Qqf[r + (size_t)(stack + m) * NV] += acc;
      } else {
        const int m = (e - NV * 3) % 3, n = (e - NV * 3) / 3;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 5; ++j) acc += df[j + 5 * m] * (rr[j] * df[j + 5 * n]);
        Qff[(stack + m) + (size_t)(stack + n) * NFP] += acc;
      }
    }
    __syncthreads();
  }
}

// expandSlackAndDual (:238-268) + fraction-to-boundary (pdipm.hxx:121-142)
template <int NV, int NF>
__global__ __launch_bounds__(64) void cone_expand_kernel(ConeArgs a) {
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  const int nact = g.dimf / a.contact_dim;
  if (nact == 0) return;
  const size_t rec = (size_t)b * a.nstages + st;
  double* nr = a.con + rec * a.nl.stride;
  const double* cone = a.cone + rec * a.cone_stride;
  const double* dr = a.dir + rec * a.dl.stride;
  const double* dqv = dr + a.dl.off[RTOC_DIR_DX];
  const double* dfv = dr + a.dl.off[RTOC_DIR_DAF] + NV;
  double fp = 1.0, fd = 1.0;
  if (lane < 5 * nact) {
    const int k = lane / 5, j = lane % 5, r = a.row0 + lane, stack = k * a.contact_dim;
    const double* dq = cone + (size_t)k * 5 * NV;
    const double* df = cone + a.dgdf_off + k * 15;
    double accq = 0.0, accf = 0.0;
    for (int c = 0; c < NV; ++c) accq += dq[j + 5 * c] * dqv[c];
    for (int m = 0; m < 3; ++m) accf += df[j + 5 * m] * dfv[stack + m];
    const double slack = nr[a.nl.off[RTOC_CON_SLACK] + r], dual = nr[a.nl.off[RTOC_CON_DUAL] + r];
    const double dslack = -accq - accf - nr[a.nl.off[RTOC_CON_RESIDUAL] + r];  // (:252-254)
    const double ddual = -(dual * dslack + nr[a.nl.off[RTOC_CON_CMPL] + r]) / slack;
    nr[a.nl.off[RTOC_CON_DSLACK] + r] = dslack;
    nr[a.nl.off[RTOC_CON_DDUAL] + r] = ddual;
    const double fs = -a.tau * (slack / dslack), fdd = -a.tau * (dual / ddual);
    if (fs > 0.0 && fs < 1.0) fp = fs;
    if (fdd > 0.0 && fdd < 1.0) fd = fdd;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    fp = fmin(fp, __shfl_xor(fp, off, 64));
    fd = fmin(fd, __shfl_xor(fd, off, 64));
  }
  if (lane == 0) {
    atomicMin(&a.steps[2 * b + 0], (unsigned long long)__double_as_longlong(fp));
    atomicMin(&a.steps[2 * b + 1], (unsigned long long)__double_as_longlong(fd));
  }
}

// updateSlack / updateDual of the cone rows (constraints_impl.hxx:167-182)
__global__ __launch_bounds__(64) void cone_update_kernel(ConeArgs a) {
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const int nact = a.grid[st].dimf / a.contact_dim;
  if (lane >= 5 * nact) return;
  double* nr = a.con + ((size_t)b * a.nstages + st) * a.nl.stride;
  const double* steps = reinterpret_cast<const double*>(a.steps);
  const int r = a.row0 + lane;
  nr[a.nl.off[RTOC_CON_SLACK] + r] += steps[2 * b] * nr[a.nl.off[RTOC_CON_DSLACK] + r];
  nr[a.nl.off[RTOC_CON_DUAL] + r] += steps[2 * b + 1] * nr[a.nl.off[RTOC_CON_DDUAL] + r];
}

}  // namespace rtoc

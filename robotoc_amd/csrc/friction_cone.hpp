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
  int max_contacts, contact_dim, row0, rows_per_contact;
  int cone_stride, dgdf_off;
  double tau;
  rtoc_record_layout kl, cl, nl, dl;
};

// the LDS writes of this wave become visible to its other lanes (one wave per grid point: no s_barrier needed)
__device__ __forceinline__ void cone_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// scratch doubles the condensation bodies need (LDS, owned by one wave for the duration of the call)
template <int NV, int NF>
struct ConeScratch {
  static constexpr int MAXC = NF / 3 > 0 ? NF / 3 : 1, MAXW = NF / 6 > 0 ? NF / 6 : 1;
  static constexpr int FRICTION = MAXC * (5 * NV + 15 + 5 + 5);
  static constexpr int WRENCH = NF >= 6 ? MAXW * (RTOC_WRENCH_ROWS * 6 + 2 * RTOC_WRENCH_ROWS) : 0;  // a surface contact has 6 force components
  static constexpr int DOUBLES = FRICTION > WRENCH ? FRICTION : WRENCH;
};

// One wave, one (instance b, grid point st).  Also called from mjtjinv_kernel (condense.hpp), where the
// rows ride along with the first kernel of the split condensation instead of costing a launch of their own.
template <int NV, int NF>
__device__ __forceinline__ void cone_condense_body(const ConeArgs& a, const int b, const int st, const int lane,
                                                   double* const scratch) {
  constexpr int NX = 2 * NV, NFP = NF > 0 ? NF : 1, MAXC = NF / 3 > 0 ? NF / 3 : 1;
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
  // Everything the rows need is fetched in ONE round trip (all contacts), the Hessian / gradient
  // entries are accumulated over the contacts in registers -- in contact order, like the reference's
  // loop (:199-233) -- and get one read-modify-write each: two HBM latencies per grid point instead
  // of two per contact.
  double(*const dq)[5 * NV] = reinterpret_cast<double(*)[5 * NV]>(scratch);
  double(*const df)[15] = reinterpret_cast<double(*)[15]>(scratch + MAXC * 5 * NV);
  double(*const cond)[5] = reinterpret_cast<double(*)[5]>(scratch + MAXC * (5 * NV + 15));
  double(*const rr)[5] = reinterpret_cast<double(*)[5]>(scratch + MAXC * (5 * NV + 15 + 5));
  for (int e = lane; e < nact * 5 * NV; e += 64) dq[e / (5 * NV)][e % (5 * NV)] = cone[e];
  for (int e = lane; e < nact * 15; e += 64) df[e / 15][e % 15] = cone[a.dgdf_off + e];
  if (lane < 5 * nact) {
    const int r = a.row0 + lane;
    const double slack = nr[a.nl.off[RTOC_CON_SLACK] + r], dual = nr[a.nl.off[RTOC_CON_DUAL] + r];
    const double c = (dual * nr[a.nl.off[RTOC_CON_RESIDUAL] + r] - nr[a.nl.off[RTOC_CON_CMPL] + r]) / slack;
    nr[a.nl.off[RTOC_CON_COND] + r] = c;  // computeCondensingCoeffcient<5> (:202)
    cond[lane / 5][lane % 5] = c;
    rr[lane / 5][lane % 5] = dual / slack;  // (:211-212)
  }
  // read-modify-write targets: issued now, consumed after the sums
  constexpr int NQ = (NV * NV + 63) / 64;
  double cq[NQ];
#pragma unroll
  for (int p = 0; p < NQ; ++p) {
    const int e = lane + 64 * p;
    cq[p] = Qxx[(e < NV * NV ? e % NV : 0) + (size_t)(e < NV * NV ? e / NV : 0) * NX];
  }
  const double clx = lx[lane < NV ? lane : 0];
  cone_wave_sync();
  // lq += dg_dq^T cond (:206)
  if (lane < NV) {
    double v = clx;
    for (int k = 0; k < nact; ++k) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 5; ++j) acc += dq[k][j + 5 * lane] * cond[k][j];
      v += acc;
    }
    lx[lane] = v;
  }
  // lf += dg_df^T cond (:207-208): one lane per (contact, component)
  if (lane >= 32 && lane < 32 + 3 * nact) {
    const int k = (lane - 32) / 3, m = (lane - 32) % 3;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 5; ++j) acc += df[k][j + 5 * m] * cond[k][j];
    lf[k * a.contact_dim + m] += acc;
  }
  // Qqq += sum_k dg_dq^T (r dg_dq) (:215-216)
#pragma unroll
  for (int p = 0; p < NQ; ++p) {
    const int e = lane + 64 * p;
    if (e < NV * NV) {
      const int r = e % NV, c = e / NV;
      double v = cq[p];
      for (int k = 0; k < nact; ++k) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 5; ++j) acc += dq[k][j + 5 * r] * (rr[k][j] * dq[k][j + 5 * c]);
        v += acc;
      }
      Qxx[r + (size_t)c * NX] = v;
    }
  }
  // Qqf[:, stack..+3] += dg_dq^T (r dg_df) (:217-218) ; Qff block += dg_df^T (r dg_df) (:219-220):
  // every (contact, entry) is a distinct memory location
  for (int e = lane; e < nact * (NV * 3 + 9); e += 64) {
    const int k = e / (NV * 3 + 9), w = e % (NV * 3 + 9), stack = k * a.contact_dim;
    if (w < NV * 3) {
      const int r = w % NV, m = w / NV;
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 5; ++j) acc += dq[k][j + 5 * r] * (rr[k][j] * df[k][j + 5 * m]);
      Qqf[r + (size_t)(stack + m) * NV] += acc;
    } else {
      const int m = (w - NV * 3) % 3, n = (w - NV * 3) / 3;
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 5; ++j) acc += df[k][j + 5 * m] * (rr[k][j] * df[k][j + 5 * n]);
      Qff[(stack + m) + (size_t)(stack + n) * NFP] += acc;
    }
  }
}

template <int NV, int NF>
__global__ __launch_bounds__(64) void cone_condense_kernel(ConeArgs a) {
  __shared__ double scratch[ConeScratch<NV, NF>::DOUBLES];
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  cone_condense_body<NV, NF>(a, b, st, threadIdx.x, scratch);
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
static __global__ __launch_bounds__(64) void cone_update_kernel(ConeArgs a) {
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const int nact = a.grid[st].dimf / a.contact_dim;
  if (lane >= a.rows_per_contact * nact) return;  // <= 5*4 friction rows, <= 17*2 wrench rows
  double* nr = a.con + ((size_t)b * a.nstages + st) * a.nl.stride;
  const double* steps = reinterpret_cast<const double*>(a.steps);
  const int r = a.row0 + lane;
  nr[a.nl.off[RTOC_CON_SLACK] + r] += steps[2 * b] * nr[a.nl.off[RTOC_CON_DSLACK] + r];
  nr[a.nl.off[RTOC_CON_DUAL] + r] += steps[2 * b + 1] * nr[a.nl.off[RTOC_CON_DDUAL] + r];
}

// ---------------------------------------------------------------------------------------------
// Contact wrench cones (ContactWrenchCone, reference src/constraints/contact_wrench_cone.cpp): 17 rows
// per active surface contact, g = cone * f with the 17 x 6 cone matrix of the RTOC_BUF_CONE record
// (wrench layout, rtoc_layout.h).  Only Qff and lf are touched (:209-238); at most 3 contacts
// (51 rows) fit the one wave that serves a grid point.
// ---------------------------------------------------------------------------------------------
template <int NV, int NF>
__device__ __forceinline__ void wrench_condense_body(const ConeArgs& a, const int b, const int st, const int lane,
                                                     double* const scratch) {
  constexpr int NFP = NF > 0 ? NF : 1, MAXC = NF / 6 > 0 ? NF / 6 : 1, WR = RTOC_WRENCH_ROWS;
  static_assert(MAXC * WR <= 64, "one lane per wrench-cone row");
  const int nact = a.grid[st].dimf / 6;
  if (nact == 0) return;
  const size_t rec = (size_t)b * a.nstages + st;
  double* cr = a.cdd + rec * a.cl.stride;
  double* nr = a.con + rec * a.nl.stride;
  const double* cone = a.cone + rec * a.cone_stride;
  double* Qff = cr + a.cl.off[RTOC_CDD_QFF];
  double* lf = cr + a.cl.off[RTOC_CDD_LF];
  double(*const J)[WR * 6] = reinterpret_cast<double(*)[WR * 6]>(scratch);
  double(*const cond)[WR] = reinterpret_cast<double(*)[WR]>(scratch + MAXC * WR * 6);
  double(*const rr)[WR] = reinterpret_cast<double(*)[WR]>(scratch + MAXC * WR * 7);
  for (int e = lane; e < nact * WR * 6; e += 64) J[e / (WR * 6)][e % (WR * 6)] = cone[e];
  if (lane < WR * nact) {
    const int r = a.row0 + lane;
    const double slack = nr[a.nl.off[RTOC_CON_SLACK] + r], dual = nr[a.nl.off[RTOC_CON_DUAL] + r];
    const double c = (dual * nr[a.nl.off[RTOC_CON_RESIDUAL] + r] - nr[a.nl.off[RTOC_CON_CMPL] + r]) / slack;
    nr[a.nl.off[RTOC_CON_COND] + r] = c;       // computeCondensingCoeffcient<17> (:228)
    cond[lane / WR][lane % WR] = c;
    rr[lane / WR][lane % WR] = dual / slack;  // (:224-225)
  }
  // inactive rows keep cond = 0 like data.cond.setZero() (:213)
  if (lane >= WR * nact && lane < WR * a.max_contacts) nr[a.nl.off[RTOC_CON_COND] + a.row0 + lane] = 0.0;
  cone_wave_sync();
  for (int e = lane; e < nact * 36; e += 64) {
    const int kk = e / 36, ww = e % 36, mm = ww % 6, nn = ww / 6, stack = kk * 6;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < WR; ++j) acc += J[kk][j + WR * mm] * (rr[kk][j] * J[kk][j + WR * nn]);
    Qff[(stack + mm) + (size_t)(stack + nn) * NFP] += acc;  // (:226-227)
  }
  if (lane < 6 * nact) {
    const int kk = lane / 6, mm = lane % 6;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < WR; ++j) acc += J[kk][j + WR * mm] * cond[kk][j];
    lf[kk * 6 + mm] += acc;  // (:229-230)
  }
}

template <int NV, int NF>
__global__ __launch_bounds__(64) void wrench_condense_kernel(ConeArgs a) {
  __shared__ double scratch[ConeScratch<NV, NF>::DOUBLES];
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  wrench_condense_body<NV, NF>(a, b, st, threadIdx.x, scratch);
}

// expandSlackAndDual (:241-270) + fraction-to-boundary (pdipm.hxx:121-142)
template <int NV, int NF>
__global__ __launch_bounds__(64) void wrench_expand_kernel(ConeArgs a) {
  constexpr int WR = RTOC_WRENCH_ROWS;
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const int nact = a.grid[st].dimf / 6;
  if (nact == 0) return;
  const size_t rec = (size_t)b * a.nstages + st;
  double* nr = a.con + rec * a.nl.stride;
  const double* cone = a.cone + rec * a.cone_stride;
  const double* dfv = a.dir + rec * a.dl.stride + a.dl.off[RTOC_DIR_DAF] + NV;
  double fp = 1.0, fd = 1.0;
  if (lane < WR * nact) {
    const int k = lane / WR, j = lane % WR, r = a.row0 + lane;
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < 6; ++m) acc += cone[k * WR * 6 + j + WR * m] * dfv[k * 6 + m];
    const double slack = nr[a.nl.off[RTOC_CON_SLACK] + r], dual = nr[a.nl.off[RTOC_CON_DUAL] + r];
    const double dslack = -acc - nr[a.nl.off[RTOC_CON_RESIDUAL] + r];  // (:260-262)
    const double ddual = -(dual * dslack + nr[a.nl.off[RTOC_CON_CMPL] + r]) / slack;
    nr[a.nl.off[RTOC_CON_DSLACK] + r] = dslack;
    nr[a.nl.off[RTOC_CON_DDUAL] + r] = ddual;
    const double fs = -a.tau * (slack / dslack), fdd = -a.tau * (dual / ddual);
    if (fs > 0.0 && fs < 1.0) fp = fs;
    if (fdd > 0.0 && fdd < 1.0) fd = fdd;
  } else if (lane < WR * a.max_contacts) {  // dslack.fill(1), ddual.fill(1) (:247-248)
    nr[a.nl.off[RTOC_CON_DSLACK] + a.row0 + lane] = 1.0;
    nr[a.nl.off[RTOC_CON_DDUAL] + a.row0 + lane] = 1.0;
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

}  // namespace rtoc

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
  int impact_cones;  // RTOC_OPT_IMPACT_CONES: 0 = no rows on impact grids (a Constraints object without ImpactFrictionCone)
  double tau;
  rtoc_record_layout kl, cl, nl, dl;
  long long* prof;
};
#ifdef RTOC_ENABLE_PROF
#define RTOC_KPROF(k) do { if (a.prof && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0) a.prof[(k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define RTOC_KPROF(k) do { } while (0)
#endif

// active contacts of a grid point that carry cone rows
__device__ __forceinline__ int cone_nact(const rtoc_grid& g, int contact_dim, int impact_cones) {
  return (g.type == RTOC_GRID_IMPACT && !impact_cones) ? 0 : g.dimf / contact_dim;
}

// the LDS writes of this wave become visible to its other lanes (one wave per grid point: no s_barrier needed)
__device__ __forceinline__ void cone_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// The Hessian / gradient contributions of the rows of one grid point are ONE Gram product: with G the stacked
// Jacobian of the active rows (K rows; columns = the primal entries they touch), R = diag(dual / slack) and
// c the condensing coefficients,  [H | g] += G^T [R G | c].  G is staged K-major in LDS (leading dimension
// LD = 16 T, the rider column c right after the NC Jacobian columns), so that both MFMA operands of every
// k-step are the same conflict-free row read.  Scratch: 4 KS x LD for G and 4 KS for diag R.
template <int K, int NC>
struct ConeGram {
  static constexpr int KS = (K + 3) / 4, T = (NC + 1 + 15) / 16, LD = 16 * T, DOUBLES = 4 * KS * LD + 4 * KS;
};

// scratch doubles the condensation bodies need (LDS, owned by one wave for the duration of the call)
template <int NV, int NF>
struct ConeScratch {
  static constexpr int MAXC = NF / 3 > 0 ? NF / 3 : 1, MAXW = NF / 6 > 0 ? NF / 6 : 1;
  static constexpr int FRICTION = ConeGram<5 * MAXC, NV + NF>::DOUBLES;
  static constexpr int WRENCH = NF >= 6 ? ConeGram<RTOC_WRENCH_ROWS * MAXW, NF>::DOUBLES : 0;  // a surface contact has 6 force components
  static constexpr int DOUBLES = FRICTION > WRENCH ? FRICTION : WRENCH;
};

// acc[ta][tb] = tile (ta, tb) of G^T [R G | c]; lane (li, q) holds rows 16 ta + drow(q, r), column 16 tb + li.
// `need(ta, tb)` (compile-time foldable) drops tiles nobody reads.
template <int K, int NC, class Need>
__device__ __forceinline__ void cone_gram_tiles(const double* const Gs, const double* const rs, const int lane,
                                                d4 (&acc)[ConeGram<K, NC>::T][ConeGram<K, NC>::T], Need need) {
  using CG = ConeGram<K, NC>;
  const int li = lane & 15, q = lane >> 4;
#pragma unroll
  for (int ta = 0; ta < CG::T; ++ta)
#pragma unroll
    for (int tb = 0; tb < CG::T; ++tb) acc[ta][tb] = zero4();
#pragma unroll
  for (int s = 0; s < CG::KS; ++s) {
    const int kk = 4 * s + q;
    const double r = rs[kk];
    double gv[CG::T], bv[CG::T];
#pragma unroll
    for (int t = 0; t < CG::T; ++t) {
      gv[t] = Gs[kk * CG::LD + 16 * t + li];
      bv[t] = gv[t] * ((16 * t + li == NC) ? 1.0 : r);
    }
#pragma unroll
    for (int ta = 0; ta < CG::T; ++ta)
#pragma unroll
      for (int tb = 0; tb < CG::T; ++tb)
        if (need(ta, tb)) acc[ta][tb] = mfma16(gv[ta], bv[tb], acc[ta][tb]);
  }
}

// One wave, one (instance b, grid point st).  Also called from mjtjinv_kernel (condense.hpp), where the
// rows ride along with the first kernel of the split condensation instead of costing a launch of their own.
template <int NV, int NF, int CD>
__device__ __forceinline__ void cone_condense_body_cd(const ConeArgs& a, const int b, const int st, const int lane,
                                                      double* const scratch) {
  constexpr int NX = 2 * NV, NFP = NF > 0 ? NF : 1, MAXC = NF / 3 > 0 ? NF / 3 : 1;
  constexpr int K = 5 * MAXC, NC = NV + NF;
  using CG = ConeGram<K, NC>;
  constexpr int T = CG::T, LD = CG::LD;
  static_assert(4 * CG::KS <= 64, "one lane per cone row");
  const int li = lane & 15, q = lane >> 4;
  const size_t rec = (size_t)b * a.nstages + st;
  double* const kr = a.kkt + rec * a.kl.stride;
  double* const cr = a.cdd + rec * a.cl.stride;
  double* const nr = a.con + rec * a.nl.stride;
  const double* const cone = a.cone + rec * a.cone_stride;
  constexpr int cd = CD;  // compile-time: the entry -> target map below divides by it 8 times per entry
  // ---- every load of the grid point is requested here, from addresses that do not depend on the grid
  // descriptor (the records are max-size): one HBM round trip, the descriptor's own fetch inside it.
  constexpr int ND = (MAXC * 5 * NV + 63) / 64, NG = (MAXC * 15 + 63) / 64;
  double gdq[ND], gdf[NG];
#pragma unroll
  for (int p = 0; p < ND; ++p) {
    const int e = lane + 64 * p;
    gdq[p] = cone[e < a.max_contacts * 5 * NV ? e : 0];  // the record holds max_contacts blocks
  }
#pragma unroll
  for (int p = 0; p < NG; ++p) {
    const int e = lane + 64 * p;
    gdf[p] = cone[a.dgdf_off + (e < a.max_contacts * 15 ? e : 0)];
  }
  const int rw = a.row0 + (lane < 5 * a.max_contacts ? lane : 0);
  const double slack = nr[a.nl.off[RTOC_CON_SLACK] + rw], dual = nr[a.nl.off[RTOC_CON_DUAL] + rw],
               resid = nr[a.nl.off[RTOC_CON_RESIDUAL] + rw], cmpl = nr[a.nl.off[RTOC_CON_CMPL] + rw];
  // Entry (row, col) of G^T [R G | c] -> the Hessian / gradient entry it is added to (friction_cone.cpp):
  //   (q, q) Qqq (:215-216)   (q, f) Qqf (:217-218)   (f, f) Qff, diagonal 3x3 block of the contact (:219-220)
  //   (q, rider) lq (:206)    (f, rider) lf (:207-208);  nullptr: nothing (Qfq is not stored, and the blocks
  // between different contacts are zero).  `kc` = the contact whose activity gates the write, -1: always.
  auto target = [&](const int row, const int col, int& kc) -> double* {
    kc = -1;
    const int fr = row - NV, fc = col - NV;
    const int kr_ = fr / cd, mr = fr % cd, kc_ = fc / cd, mc = fc % cd;
    const bool frow = row >= NV && row < NC && mr < 3 && kr_ < MAXC, fcol = col >= NV && col < NC && mc < 3 && kc_ < MAXC;
    if (row < NV) {
      if (col < NV) return kr + a.kl.off[RTOC_KKT_QXX] + row + (size_t)col * NX;
      if (col == NC) return kr + a.kl.off[RTOC_KKT_LX] + row;
      if (fcol) {
        kc = kc_;
        return cr + a.cl.off[RTOC_CDD_QQF] + row + (size_t)fc * NV;
      }
      return nullptr;
    }
    if (!frow) return nullptr;
    kc = kr_;
    if (col == NC) return cr + a.cl.off[RTOC_CDD_LF] + fr;
    if (fcol && kc_ == kr_) return cr + a.cl.off[RTOC_CDD_QFF] + fr + (size_t)fc * NFP;
    return nullptr;
  };
  auto need = [](const int ta, const int tb) { return !(16 * ta >= NV && 16 * tb + 15 < NV); };  // not all-(f, q)
  double ce[T][T][4];
#pragma unroll
  for (int ta = 0; ta < T; ++ta)
#pragma unroll
    for (int tb = 0; tb < T; ++tb)
      if (need(ta, tb)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int kc;
          const double* const p = target(16 * ta + drow(q, r), 16 * tb + li, kc);
          ce[ta][tb][r] = *(p ? p : cr);
        }
      }
  const int nact = cone_nact(a.grid[st], cd, a.impact_cones);
  if (nact == 0) return;
  RTOC_KPROF(25);
  // ---- G, diag R and the rider column into LDS
  double* const Gs = scratch;
  double* const rs = scratch + 4 * CG::KS * LD;
  for (int e = lane; e < 4 * CG::KS * LD; e += 64) Gs[e] = 0.0;
  cone_wave_sync();
#pragma unroll
  for (int p = 0; p < ND; ++p) {  // dg_dq of contact k: 5 x NV, column-major
    const int e = lane + 64 * p, k = e / (5 * NV), w = e % (5 * NV);
    if (e < nact * 5 * NV) Gs[(5 * k + w % 5) * LD + w / 5] = gdq[p];
  }
#pragma unroll
  for (int p = 0; p < NG; ++p) {  // dg_df of contact k: 5 x 3, on the contact's own force columns
    const int e = lane + 64 * p, k = e / 15, w = e % 15;
    if (e < nact * 15) Gs[(5 * k + w % 5) * LD + NV + k * cd + w / 5] = gdf[p];
  }
  if (lane < 4 * CG::KS) {
    const bool on = lane < 5 * nact;
    if (on) {
      const double c = (dual * resid - cmpl) / slack;
      nr[a.nl.off[RTOC_CON_COND] + rw] = c;  // computeCondensingCoeffcient<5> (:202)
      Gs[lane * LD + NC] = c;
    }
    rs[lane] = on ? dual / slack : 0.0;  // (:211-212)
  }
  cone_wave_sync();
  RTOC_KPROF(26);
  d4 acc[T][T];
  cone_gram_tiles<K, NC>(Gs, rs, lane, acc, need);
  RTOC_KPROF(27);
  {
#pragma unroll
    for (int ta = 0; ta < T; ++ta)
#pragma unroll
      for (int tb = 0; tb < T; ++tb)
        if (need(ta, tb)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int kc;
            double* const p = target(16 * ta + drow(q, r), 16 * tb + li, kc);
            if (p && kc < nact) *p = ce[ta][tb][r] + acc[ta][tb][r];
          }
        }
  }
  RTOC_KPROF(28);
}

// contact_dim: 3 (point contacts) or 6 (surface contacts, friction on the force part) -- rtoc_set_friction_cones
template <int NV, int NF>
__device__ __forceinline__ void cone_condense_body(const ConeArgs& a, const int b, const int st, const int lane,
                                                   double* const scratch) {
  if (a.contact_dim == 3)
    cone_condense_body_cd<NV, NF, 3>(a, b, st, lane, scratch);
  else
    cone_condense_body_cd<NV, NF, 6>(a, b, st, lane, scratch);
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
  const int nact = cone_nact(g, a.contact_dim, a.impact_cones);
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
  const int nact = cone_nact(a.grid[st], a.contact_dim, a.impact_cones);
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
  constexpr int K = WR * MAXC, NC = NF;
  using CG = ConeGram<K, NC>;
  constexpr int T = CG::T, LD = CG::LD;
  static_assert(4 * CG::KS <= 64, "one lane per wrench-cone row");
  const int li = lane & 15, q = lane >> 4;
  const size_t rec = (size_t)b * a.nstages + st;
  double* const cr = a.cdd + rec * a.cl.stride;
  double* const nr = a.con + rec * a.nl.stride;
  const double* const cone = a.cone + rec * a.cone_stride;
  // all loads up front, from addresses independent of the grid descriptor (cf. cone_condense_body)
  constexpr int NJ = (MAXC * WR * 6 + 63) / 64;
  double gj[NJ];
#pragma unroll
  for (int p = 0; p < NJ; ++p) {
    const int e = lane + 64 * p;
    gj[p] = cone[e < a.max_contacts * WR * 6 ? e : 0];  // the record holds max_contacts blocks
  }
  const int rw = a.row0 + (lane < WR * a.max_contacts ? lane : 0);
  const double slack = nr[a.nl.off[RTOC_CON_SLACK] + rw], dual = nr[a.nl.off[RTOC_CON_DUAL] + rw],
               resid = nr[a.nl.off[RTOC_CON_RESIDUAL] + rw], cmpl = nr[a.nl.off[RTOC_CON_CMPL] + rw];
  // (f, f) within one contact -> Qff (:226-227) ; (f, rider) -> lf (:229-230); `kc` = the gating contact
  auto target = [&](const int row, const int col, int& kc) -> double* {
    kc = row / 6;
    if (row >= NF || kc >= MAXC) return nullptr;
    if (col == NC) return cr + a.cl.off[RTOC_CDD_LF] + row;
    if (col < NF && col / 6 == kc) return cr + a.cl.off[RTOC_CDD_QFF] + row + (size_t)col * NFP;
    return nullptr;
  };
  auto need = [](const int, const int) { return true; };
  double ce[T][T][4];
#pragma unroll
  for (int ta = 0; ta < T; ++ta)
#pragma unroll
    for (int tb = 0; tb < T; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int kc;
        const double* const p = target(16 * ta + drow(q, r), 16 * tb + li, kc);
        ce[ta][tb][r] = *(p ? p : cr);
      }
  const int nact = cone_nact(a.grid[st], 6, a.impact_cones);
  // inactive rows keep cond = 0 like data.cond.setZero() (:213)
  if (lane >= WR * nact && lane < WR * a.max_contacts) nr[a.nl.off[RTOC_CON_COND] + a.row0 + lane] = 0.0;
  if (nact == 0) return;
  double* const Gs = scratch;
  double* const rs = scratch + 4 * CG::KS * LD;
  for (int e = lane; e < 4 * CG::KS * LD; e += 64) Gs[e] = 0.0;
  cone_wave_sync();
#pragma unroll
  for (int p = 0; p < NJ; ++p) {  // the 17 x 6 cone matrix of contact k on the contact's own force columns
    const int e = lane + 64 * p, k = e / (WR * 6), w = e % (WR * 6);
    if (e < nact * WR * 6) Gs[(WR * k + w % WR) * LD + 6 * k + w / WR] = gj[p];
  }
  if (lane < 4 * CG::KS) {
    const bool on = lane < WR * nact;
    if (on) {
      const double c = (dual * resid - cmpl) / slack;
      nr[a.nl.off[RTOC_CON_COND] + rw] = c;  // computeCondensingCoeffcient<17> (:228)
      Gs[lane * LD + NC] = c;
    }
    rs[lane] = on ? dual / slack : 0.0;  // (:224-225)
  }
  cone_wave_sync();
  d4 acc[T][T];
  cone_gram_tiles<K, NC>(Gs, rs, lane, acc, need);
#pragma unroll
  for (int ta = 0; ta < T; ++ta)
#pragma unroll
    for (int tb = 0; tb < T; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int kc;
        double* const p = target(16 * ta + drow(q, r), 16 * tb + li, kc);
        if (p && kc < nact) *p = ce[ta][tb][r] + acc[ta][tb][r];
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
  const int nact = cone_nact(a.grid[st], 6, a.impact_cones);
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

// device_utils.hpp -- shared device helpers for the gfx950 kernels.
//
// CDNA4 specifics used here:
//   * v_mfma_f64_16x16x4_f64: one wavefront (64 lanes) computes a 16x16 fp64 tile
//     with K=4.  Operand layout (cdna_hip_programming.md 3): lane l supplies
//     A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; result register r of lane l
//     holds D[row = (l>>4) + 4*r][col = l&15].
//   * 64-wide wavefronts, LDS-resident operands, ds_read_b64 fragments.
#pragma once
#include "../../include/rtoc_layout.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtoc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Row (within the 16x16 tile) of result register r for a lane with q = lane>>4.
__device__ __forceinline__ constexpr int drow(int q, int r) { return q + 4 * r; }

__device__ __forceinline__ d4 zero4() {
  d4 z = {0.0, 0.0, 0.0, 0.0};
  return z;
}

// Leading dimension for LDS-resident NX x NX operands: even, and ld/2 odd, so that the
// 16 rows x 2 k-columns a half-wave touches in one ds_read_b64 fall in distinct banks.
__host__ __device__ constexpr int lds_ld(int n) { return (n % 4 == 2) ? n : n + 2; }  // n even

template <int NT>
__device__ __forceinline__ void copy_g2s_flat(double* __restrict__ dst, const double* __restrict__ src,
                                              int n, int tid) {
  // n doubles, src 16B aligned; dst contiguous
  const int n2 = n >> 1;
  const d2* s2 = reinterpret_cast<const d2*>(src);
  d2* t2 = reinterpret_cast<d2*>(dst);
#pragma unroll 8
  for (int e = tid; e < n2; e += NT) t2[e] = s2[e];
  if ((n & 1) && tid == 0) dst[n - 1] = src[n - 1];
}

// Column-major ROWS x COLS matrix copies (ld = ROWS in HBM, LD in LDS), 16 B per lane.
// Lane mapping "column groups": CPL = ROWS/2 lanes cover one column, CPP = NT/CPL columns are
// moved per pass.  (row pair, column offset) are computed once per call, every pass then uses
// constant strides: no per-element div/mod, LDS and HBM addresses are affine in the pass index.
template <int NT, int ROWS>
struct MatMap {
  static_assert(ROWS % 2 == 0, "even rows required for 16 B moves");
  static constexpr int CPL = ROWS / 2;
  static_assert(CPL <= NT, "one pass must cover at least one column");
  static constexpr int CPP = NT / CPL;
  static constexpr int passes(int cols) { return (cols + CPP - 1) / CPP; }
};

template <int NT, int ROWS, int COLS, int LD>
__device__ __forceinline__ void copy_g2s_mat(double* __restrict__ dst, const double* __restrict__ src,
                                             int tid) {
  using M = MatMap<NT, ROWS>;
  static_assert(LD % 2 == 0, "even ld");
  const int r2 = tid % M::CPL, c0 = tid / M::CPL;
  const d2* s2 = reinterpret_cast<const d2*>(src) + r2 + c0 * M::CPL;
  double* d = dst + 2 * r2 + c0 * LD;
  const bool act = c0 < M::CPP;
#pragma unroll
  for (int k = 0; k < M::passes(COLS); ++k) {
    if (act && (k * M::CPP + c0 < COLS))
      *reinterpret_cast<d2*>(d + k * M::CPP * LD) = s2[k * M::CPP * M::CPL];
  }
}

template <int NT, int ROWS, int COLS, int LD>
__device__ __forceinline__ void copy_s2g_mat(double* __restrict__ dst, const double* __restrict__ src,
                                             int tid) {
  using M = MatMap<NT, ROWS>;
  static_assert(LD % 2 == 0, "even ld");
  const int r2 = tid % M::CPL, c0 = tid / M::CPL;
  d2* t2 = reinterpret_cast<d2*>(dst) + r2 + c0 * M::CPL;
  const double* sp = src + 2 * r2 + c0 * LD;
  const bool act = c0 < M::CPP;
#pragma unroll
  for (int k = 0; k < M::passes(COLS); ++k) {
    if (act && (k * M::CPP + c0 < COLS))
      t2[k * M::CPP * M::CPL] = *reinterpret_cast<const d2*>(sp + k * M::CPP * LD);
  }
}

template <int NT>
__device__ __forceinline__ void copy_s2g_flat(double* __restrict__ dst, const double* __restrict__ src,
                                              int n, int tid) {
  for (int e = tid; e < n; e += NT) dst[e] = src[e];
}

// n doubles (n even or the tail handled), 16 B per lane, LDS -> HBM
template <int NT>
__device__ __forceinline__ void copy_s2g_flat16(double* __restrict__ dst, const double* __restrict__ src,
                                                int n, int tid) {
  const int n2 = n >> 1;
  const d2* s2 = reinterpret_cast<const d2*>(src);
  d2* t2 = reinterpret_cast<d2*>(dst);
#pragma unroll 8
  for (int e = tid; e < n2; e += NT) t2[e] = s2[e];
  if ((n & 1) && tid == 0) dst[n - 1] = src[n - 1];
}

__device__ __forceinline__ double shfl_d(double v, int src_lane) { return __shfl(v, src_lane, 64); }

// Broadcast of one lane's double to the whole wave through the scalar unit (v_readlane_b32 x2):
// a few cycles, versus the ~100-cycle LDS crossbar round trip of ds_bpermute.  `src_lane` must be
// wave-uniform (it is a compile-time constant at every call site).
__device__ __forceinline__ double readlane_d(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to fp64 accuracy: hardware estimate + two Newton steps (no separate sqrt and divide)
__device__ __forceinline__ double rsqrt_d(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * __builtin_fma(-h * y, y, 1.5);
  y = y * __builtin_fma(-h * y, y, 1.5);
  return y;
}

__device__ __forceinline__ bool is_bad(double v) { return !(fabs(v) <= 1.79769313486231570815e308); }

// Time step of grid point `st` of instance `b`: the batch shares the grid STRUCTURE (rtoc_set_grid), but with switching-time
// optimisation every instance moves its own event times, i.e. has its own time steps (TimeDiscretization::correctTimeSteps,
// time_discretization.cpp:179-221): `dt_inst` = [batch][nstages] table written by sto_time_steps_kernel (sto.hpp), or
// nullptr -> the shared grid's dt.  One unconditional load through a selected address (a predicated load would compile to a
// branch with its own s_waitcnt: one more dependent round trip ahead of everything).
__device__ __forceinline__ double grid_dt(const rtoc_grid* grid, const double* dt_inst, int b, int nstages, int st) {
  const double* const p = dt_inst ? dt_inst + (size_t)b * nstages + st : &grid[st].dt;
  return *p;
}

// Field offsets of the KKT / Riccati records for a robot known at compile time: the same
// rtoc_compute_layout() the host uses, evaluated as a constant expression, so that every offset is
// an instruction immediate instead of a scalar register (the runtime table cost ~60 SGPRs and made
// the compiler spill scalars through v_writelane / v_readlane all over the stage loop).
template <int NV, int NU, int NS>
struct StaticLayout {
  static constexpr rtoc_layout make() {
    rtoc_dims d = {NV, NU, 0, NS, NS, 0};
    rtoc_layout L = {};
    rtoc_compute_layout(&d, &L);
    return L;
  }
};

}  // namespace rtoc

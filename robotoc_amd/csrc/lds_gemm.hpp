// lds_gemm.hpp -- small dense products between LDS-resident matrices with COMPILE-TIME shapes.
//
// The condensation kernel multiplies ~15 pairs of matrices whose extents are bounded by the robot
// (nv, nu, nf_max) and known when the kernel is instantiated.  With the extents, leading dimensions
// and transposition flags as template parameters every LDS address is an instruction immediate and
// every bounds test folds away except on edge tiles -- the run-time-shaped wave_gemm spent ~10k VALU
// instructions per work item on address arithmetic and masks against 262 MFMAs.
//
// Inactive rows / columns (dimf < max_dimf) are handled by ZERO PADDING in LDS, not by run-time
// extents: a product over the full compile-time K is exact when the inactive part is zero.
//
//   C(i,j) = beta*C(i,j) + alpha * sum_k A(i,k) B(k,j)      i < M, j < N, k < K
//   A(i,k) = A[i*ARS + k*ACS], B(k,j) = B[k*BRS + j*BCS], C(i,j) = C[i*CRS + j*CCS]
//
// 16x16 output tiles are dealt to the NW waves of the work item by tile index parity; one tile = one
// dependent chain of ceil(K/4) v_mfma_f64_16x16x4_f64 (back-to-back dependent f64 MFMAs issue at full
// rate on gfx950), its 2*ceil(K/4) operands fetched up front.
#pragma once
#include "device_utils.hpp"

namespace rtoc {

template <int M, int N, int K, int ARS, int ACS, int BRS, int BCS>
struct TileOps {
  static constexpr int KS = (K + 3) / 4;
  // operands of tile (tm, tn) for every k-step
  static __device__ __forceinline__ void load(const double* __restrict__ A, const double* __restrict__ B,
                                              int tm, int tn, int li, int q, double (&av)[KS],
                                              double (&bv)[KS]) {
    const int i = tm * 16 + li, j = tn * 16 + li;
    const bool iok = i < M, jok = j < N;
    const double* ap = A + (iok ? i : 0) * ARS + q * ACS;
    const double* bp = B + (jok ? j : 0) * BCS + q * BRS;
    // Rows i >= M of A / columns j >= N of B only reach output rows / columns the epilogue never sees: they read the
    // (finite) clamped row / column 0 and need no select.  Only the k-steps that run past K are zeroed (in both
    // operands: whatever lies behind the operand in LDS may not be finite).
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks * 4 + 3 < K) {
        av[ks] = ap[ks * 4 * ACS];
        bv[ks] = bp[ks * 4 * BRS];
      } else {
        const bool kok = ks * 4 + q < K;
        const double a = ap[(kok ? ks : 0) * 4 * ACS];
        const double b = bp[(kok ? ks : 0) * 4 * BRS];
        av[ks] = kok ? a : 0.0;
        bv[ks] = kok ? b : 0.0;
      }
    }
  }
};

// acc += A B for one tile
template <int KS>
__device__ __forceinline__ d4 tile_mma(d4 acc, const double (&av)[KS], const double (&bv)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) acc = mfma16(av[ks], bv[ks], acc);
  return acc;
}

// Full product into LDS (or any memory): every wave of the work item calls it; barriers are the
// caller's business.  `E` is an epilogue functor (row, col, value, slot, reg) -> void; (slot, reg) are
// compile-time after unrolling and index the registers of prefetch_tiles (tile t -> slot t / NW).
template <int NW, int M, int N, int K, int ARS, int ACS, int BRS, int BCS, class E>
__device__ __forceinline__ void lds_gemm(const double* __restrict__ A, const double* __restrict__ B, int tid,
                                         E&& epilogue) {
  using T = TileOps<M, N, K, ARS, ACS, BRS, BCS>;
  constexpr int TM = (M + 15) / 16, TN = (N + 15) / 16;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
#pragma unroll
  for (int t = 0; t < TM * TN; ++t) {
    if ((t % NW) != wave) continue;  // wave-uniform
    const int tm = t / TN, tn = t % TN;
    double av[T::KS], bv[T::KS];
    T::load(A, B, tm, tn, li, q, av, bv);
    const d4 acc = tile_mma<T::KS>(zero4(), av, bv);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * 16 + drow(q, r), col = tn * 16 + li;
      if (row < M && col < N) epilogue(row, col, acc[r], t / NW, r);
    }
  }
}

// Two products into the same tiles: acc = A1 B1 (M x N x K1), acc2 = A2 B2 restricted to rows < M2
// (the Schur updates with their Qqf corrections).  Epilogue (row, col, v1, v2, slot, reg).
template <int NW, int M, int N, int K1, int A1RS, int A1CS, int B1RS, int B1CS, int M2, int K2, int A2RS,
          int A2CS, int B2RS, int B2CS, class E>
__device__ __forceinline__ void lds_gemm2(const double* __restrict__ A1, const double* __restrict__ B1,
                                          const double* __restrict__ A2, const double* __restrict__ B2,
                                          bool second, int tid, E&& epilogue) {
  using T1 = TileOps<M, N, K1, A1RS, A1CS, B1RS, B1CS>;
  using T2 = TileOps<M2, N, K2, A2RS, A2CS, B2RS, B2CS>;
  constexpr int TM = (M + 15) / 16, TN = (N + 15) / 16;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
#pragma unroll
  for (int t = 0; t < TM * TN; ++t) {
    if ((t % NW) != wave) continue;
    const int tm = t / TN, tn = t % TN;
    double av[T1::KS], bv[T1::KS];
    T1::load(A1, B1, tm, tn, li, q, av, bv);
    const d4 acc = tile_mma<T1::KS>(zero4(), av, bv);
    d4 acc2 = zero4();
    if (tm * 16 < M2 && second) {
      double av2[T2::KS], bv2[T2::KS];
      T2::load(A2, B2, tm, tn, li, q, av2, bv2);
      acc2 = tile_mma<T2::KS>(zero4(), av2, bv2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * 16 + drow(q, r), col = tn * 16 + li;
      // (the second product's operand rows >= M2 are not zeroed on load, TileOps::load: drop them here)
      if (row < M && col < N) epilogue(row, col, acc[r], ((tm + 1) * 16 <= M2 || row < M2) ? acc2[r] : 0.0, t / NW, r);
    }
  }
}

}  // namespace rtoc

// riccati_backward_rw2.hpp -- the register-wide backward kernel (riccati_backward_rw.hpp) for T = 5 state tiles (iCub nv = 35,
// nx = 70): TWO wavefronts per OCP instance, each with the whole register file of a SIMD, two instances per CU.
//
// One wave cannot hold P+ (25 tiles) AND F (15 upper tiles): 320 of 512 registers before the first operand, and the compiler spills
// 540 (6.7 ms per 1024 instances against the tile-split kernel's 4.8).  What does fit is P+ (200) + HALF of F + one column of W.  So
// both waves keep the WHOLE value function in registers, and the stage is split twice:
//
//  * the O(n^2 nu) part by ROLE.  Wave 0 forms the rows v of PB and G = Quu + Bv^T PB[v, :], factorises it (Y = L^-1) and solves for
//    t and k; wave 1 stores P, forms all of PB and H^T = Qxu^T + PB^T A directly in the layout Z^T = Y H^T reads -- PB's C tiles are the
//    A operand as they stand, the structured rows of A synthesised fragments, Qxu^T (by DMA, in wave 1's idle G / Y place) the start
//    value of the accumulators; no transposes.  The 29 x 29 factorisation, whose column steps leave the matrix pipe idle, runs beside
//    wave 1's 256 MFMAs.  They meet ONCE (two barriers): Y, t and A^T z cross over, H^T goes to wave 0 through LDS.
//  * the O(n^3) part by COLUMN TILES of F: wave 0 owns {0, 1, 2}, wave 1 {3, 4} (6 and 9 upper tiles: 252 and 228 MFMAs in the column
//    loop), each its tiles of Z^T, Z Z^T, s and the Qxx start values; wave 0, whose way is the shorter one, forms K for all column tiles,
//    wave 1 meets wave 0's column tiles of Z^T one at a time and lets them die.  At the stage end the fifteen tiles of F meet in LDS -- in
//    the place of A, which is dead by then -- and each wave rebuilds its own copy of P+ = sym(F) from there (direct and transposed reads),
//    forming z = s+ - P+ Fx of the next grid point as the elements pass.
//
// Five workgroup barriers per stage.  Same lane algebra as riccati_backward_rw.hpp, which tests/rw_lane_model.py verifies for this
// shape (the direct H^T product: direct_ht).  Shared LDS: the dense rows of A / the F exchange, two strips, the grid table, Bv / wave
// 1's column tiles of H^T; per wave: G / Y / scratch / vectors (wave 0), Qxu / wave 0's column tiles of H^T (wave 1) -- the landing
// zone of the wave's Qxx panels.  79 KB per instance.
//
// Same scope as riccati_backward_rw.hpp (rw_applies, rtoc_capi.hip): structured Fxx, no switching-time optimisation, switching-
// constraint grid points as one-stage launches of the tile-split kernel, batches larger than the device's CU count.
#pragma once
#include "riccati_backward_rw.hpp"

namespace rtoc {

template <int NV, int NU, int NS>
constexpr bool KL_QXU_EVEN() { return StaticLayout<NV, NU, NS>::make().kkt.off[RTOC_KKT_QXU] % 2 == 0; }

template <int NV, int NU, int NS>
struct Rw2Lds {
  using C = RwCfg<NV, NU>;
  static constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  static constexpr int VOFF_LX = SL.kkt.off[RTOC_KKT_LX] - SL.kkt.off[RTOC_KKT_FX], VOFF_LU = SL.kkt.off[RTOC_KKT_LU] - SL.kkt.off[RTOC_KKT_FX];
  static constexpr int CV = (VOFF_LU + NU + 1) / 2, PS = (CV + 63) / 64;   // 16-byte chunks / DMA pieces of the strip Fx | lx | lu
  static constexpr int NTR = C::T * (C::T + 1) / 2;                          // upper tiles of F
  // ---- shared by the two waves ----
  static constexpr int OFF_A = 0;                                            // the dense rows of A; at the stage end: the fifteen tiles of F
  static constexpr int SH_A = (C::NX * C::LDA > NTR * C::SCR_TILE) ? C::NX * C::LDA : NTR * C::SCR_TILE;
  static constexpr int OFF_ST = SH_A;
  static constexpr int STRIP = C::pad8(2 * CV);                              // TWO strips, by the parity of the grid point: the next one's is
  static constexpr int OFF_GRID = OFF_ST + 2 * STRIP;                        // requested at the stage top.  Behind them: grid-point kinds (ints)
  static constexpr int OFF_BV = OFF_GRID + RV_MAX_STAGES / 2;                // Fvu, flat as in the record (NV x NU, by DMA)
  static constexpr int OFF_PW = OFF_BV + C::pad8(NV * NU + 1);
  // ---- per wave (offsets inside its block) ----
  static constexpr int P_S = 0;                                              // s+
  static constexpr int P_G = P_S + C::pad8(C::NX);                           // Quu (by DMA) -> G, factorised in place; from here on: the Qxx panels' zone
  static constexpr int P_Y = P_G + C::pad8(NU * NU);
  static constexpr int P_LINV = P_Y + C::pad8(NU * NU);
  static constexpr int P_SCR = P_LINV + C::pad8(NU);
  static constexpr int NSCR = 1;                                             // transpose tiles of a wave (the Cholesky's 256 doubles fit one)
  static constexpr int P_Z = P_SCR + NSCR * C::SCR_TILE;
  static constexpr int P_W0 = P_Z + C::pad8(C::NX);
  static constexpr int P_LUP = P_W0 + C::pad8(C::NX);
  static constexpr int P_T = P_LUP + C::pad8(NU);
  static constexpr int LDQ_ = lds_ld(C::NX);
  static constexpr int PWD = (P_T + C::pad8(NU) > P_G + 2 * 16 * LDQ_) ? P_T + C::pad8(NU) : P_G + 2 * 16 * LDQ_;   // (room for two Qxx panels)
  static constexpr int LDQ = lds_ld(C::NX), PANEL = 16 * LDQ;
  static constexpr int P_Q = P_G;
  static_assert(P_Q + 2 * PANEL <= PWD, "two Qxx panels in a wave's dead zone");
  static constexpr int DOUBLES = OFF_PW + 2 * PWD;
  static constexpr int BYTES = DOUBLES * 8;
  static_assert(BYTES <= 80 * 1024, "two instances per CU");
  static_assert(PS * 128 <= STRIP + 128, "the last DMA piece of a strip is partial");
  static_assert(NSCR * C::SCR_TILE >= 256, "the blocked Cholesky's scratch");
};

// cycle stamps of instance 0 (tools/phase_profile_rv.py; make PROF=1): wave 0 in slots 0..15, wave 1 in 16..31 of its stage's row
#ifdef RTOC_ENABLE_PROF
#define RW2_PROF(k) do { if (a.prof && b == 0 && lane0 == 0) a.prof[st * 32 + (k) + 16 * W] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define RW2_PROF(k) do { } while (0)
#endif

// the wave-uniform barrier of the two waves of an instance (their own LDS traffic first)
__device__ __forceinline__ void rw2_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// W: which of the two waves of the instance this is (compile-time: each wave's register arrays are sized for its own share)
template <int NV, int NU, int NS, int W>
__device__ __forceinline__ void rw2_body(BwdArgs a) {
  using C = RwCfg<NV, NU>;
  using M = Rw2Lds<NV, NU, NS>;
  static_assert(C::T == 5 || C::T == 4, "column tiles {0 .. (T-1)/2} | the rest");
  // column tiles of F (and of K, and row tiles of the P stores) this wave owns
  constexpr int TSPLIT = (C::T - 1) / 2;   // T = 5: wave 0 owns {0, 1, 2} (6 upper tiles), wave 1 {3, 4} (9)
  constexpr int LDH = 16 * (TSPLIT + 1);   // leading dimension of the H^T hand-over
  constexpr int LDB = 16 * (C::T - TSPLIT - 1);   // ... of wave 1's column tiles
#ifndef RW2_PSPLIT
#define RW2_PSPLIT 0
#endif
#ifndef RW2_S_BY_WAVE1
#define RW2_S_BY_WAVE1 1
#endif
  constexpr bool S_W1 = RW2_S_BY_WAVE1 != 0;      // s of all column tiles by wave 1 (they all pass through its registers) / each wave its own
  constexpr int PSPLIT = RW2_PSPLIT;                       // row tiles of the P stores wave 0 takes
  constexpr int RID = C::NX % 16;                 // the idle column of the last column tile that carries lu' -> t -> k
  static_assert(RID != 0, "an idle column in the last column tile");
  static_assert(NU * LDH <= M::P_Z - M::P_G, "H^T of wave 0's column tiles fits the place of wave 1's G, Y and scratch");
  static_assert(NU * LDB <= C::pad8(NV * NU + 1), "H^T of wave 1's column tiles fits the place of Bv");
  static_assert(C::NX * NU <= M::P_Z && KL_QXU_EVEN<NV, NU, NS>(), "Qxu fits wave 1's block ahead of z");
  auto own = [](int t) constexpr { return (t <= TSPLIT) == (W == 0); };
  constexpr int NX = C::NX, NP_ = C::NP, T = C::T, TU = C::TU, KG = C::KG, KGU = C::KGU, G0 = C::G0, G1 = C::G1, NUC = C::NUC;
  constexpr int TS = C::TS, LS = C::LS, QS = C::QS, RS = C::RS, LDA = C::LDA, HL = C::HL, SCR_LD = C::SCR_LD, SCR_TILE = C::SCR_TILE;
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt, RL = SL.ric;
  static_assert(M::VOFF_LX > 0 && M::VOFF_LU > M::VOFF_LX && M::VOFF_LX % 2 == 0 && M::VOFF_LU % 2 == 0, "Fx, lx, lu lie behind one another in the record");
  static_assert(KL.off[RTOC_KKT_FXX] % 2 == 0 && KL.off[RTOC_KKT_FX] % 2 == 0 && KL.off[RTOC_KKT_QUU] % 2 == 0 && KL.off[RTOC_KKT_QXX] % 2 == 0 && KL.stride % 2 == 0 && NX % 2 == 0, "16-byte chunks");
  constexpr int ST_LXO = M::VOFF_LX, ST_LUO = M::VOFF_LU;   // offsets inside a strip (Fx at 0)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const LdsAddr ldsp(smem);   // (32-bit LDS addresses for the DMA destinations)
  double* const sA = smem + M::OFF_A;
  double* const pw0 = smem + M::OFF_PW;                // wave 0's block: G, Y = L^-1, t live there (wave 0 factorises)
  double* const pw1 = pw0 + M::PWD;                    // wave 1's block: the place of its (unused) G / Y carries H^T to wave 0
  double* const pw_ = (W == 0) ? pw0 : pw1;            // this wave's own block
  double* const sG = pw0 + M::P_G;
  double* const sY = pw0 + M::P_Y;
  double* const scr = pw_ + M::P_SCR;
  double* const sS = pw0 + M::P_S;                     // s+: ONE copy, each wave writes its own column tiles
  double* const sHX = pw1 + M::P_G;                    // H^T[u][x], x < LDH: the column tiles of wave 0
  double* const sHB = smem + M::OFF_BV;                // H^T[u][x - LDH], the column tiles of wave 1: in the place of Bv (dead behind PB)
  const double* const sQxu = pw1;                      // Qxu (flat, as in the record), by wave 1's DMA: under its s+ / G / Y / scratch
  double* const sZ = pw0 + M::P_Z;                     // z: ONE copy, formed where P+ is rebuilt, each wave its column tiles
  double* const sW0 = pw_ + M::P_W0;
  double* const sLup = pw_ + M::P_LUP;
  double* const sT = pw_ + M::P_T;
  int* const sGrid = reinterpret_cast<int*>(smem + M::OFF_GRID);

  const int b = a.first + (int)blockIdx.x;
  if (b >= a.batch) return;
  const int lane0 = threadIdx.x & 63;
  int lane = lane0 & 63, li = lane & 15, q = lane >> 4;
  const int N = a.nstages - 1;
  const size_t kinst = (size_t)b * a.nstages * KL.stride;
  const size_t rinst = (size_t)b * a.nstages * RL.stride;
  const int hi = a.seg_hi, lo = a.seg_lo;
  unsigned stat = 0;

  // compile-time predicates of the k groups
  auto group_dense = [](int g) { return g < G1 || g >= G0; };
  auto group_all_dense = [](int g) { return 4 * g + 3 < NP_ || (4 * g >= NV && 4 * g + 3 < NX); };
  auto qsum = [](double v) __attribute__((always_inline)) {   // sum over the four lanes (li, 0..3)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
  };

  // ---- DMA: the dense k groups of A (rows [0, 4 G1) and [4 G0, 4 KG) of every column) into the padded compact layout; whole
  //      padded columns per piece, so that a piece differs from the next by a scalar ----
  constexpr int NPC_A = (NX + 64 / HL - 1) / (64 / HL);   // DMA instructions of one A
  auto issue_dma_A = [&](int stage) __attribute__((always_inline)) {
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride + KL.off[RTOC_KKT_FXX];
    constexpr int CPP = 64 / HL, NPC = NPC_A;
    const unsigned col0 = (unsigned)lane / HL, ch0 = (unsigned)lane - col0 * HL;
    const unsigned rl = 2 * ch0;                                                        // compact row of the chunk
    const unsigned srow = (rl < 4 * G1) ? rl : ((rl < 4 * C::NDG) ? rl + 4 * (G0 - G1) : 0);   // (the padding chunk loads anything)
    const unsigned voff = col0 * NX + srow;
    if (col0 < CPP) {
#pragma unroll
      for (int p = 0; p < NPC; ++p) {
        if ((p + 1) * CPP <= NX || p * CPP + (int)col0 < NX)
          __builtin_amdgcn_global_load_lds(kp + p * CPP * NX + voff, ldsp(sA + p * CPP * LDA), 16, 0, 0);
      }
    }
  };
  auto issue_dma_bv = [&](int stage) __attribute__((always_inline)) {
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride + KL.off[RTOC_KKT_FVU];
    constexpr int CBV = (NV * NU + 1) / 2, PBV = (CBV + 63) / 64;
#pragma unroll
    for (int p = 0; p < PBV; ++p) {
      const int n = lane + 64 * p;
      if (64 * (p + 1) <= CBV || n < CBV) __builtin_amdgcn_global_load_lds(kp + 2 * n, ldsp(smem + M::OFF_BV + 128 * p), 16, 0, 0);
    }
  };
  auto issue_dma_qxu = [&](int stage) __attribute__((always_inline)) {   // (wave 1: the start value of H^T)
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride + KL.off[RTOC_KKT_QXU];
    constexpr int CX = NX * NU / 2, PX = (CX + 63) / 64;
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      const int n = lane + 64 * p;
      if (64 * (p + 1) <= CX || n < CX) __builtin_amdgcn_global_load_lds(kp + 2 * n, ldsp(pw1 + 128 * p), 16, 0, 0);
    }
  };
  auto issue_dma_strip = [&](int stage) __attribute__((always_inline)) {
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride + KL.off[RTOC_KKT_FX];
#pragma unroll
    for (int p = 0; p < M::PS; ++p) {
      const int n = lane + 64 * p;
      if (64 * (p + 1) <= M::CV || n < M::CV)
        __builtin_amdgcn_global_load_lds(kp + 2 * n, ldsp(smem + M::OFF_ST + (stage & 1) * M::STRIP + 128 * p), 16, 0, 0);
    }
  };

  // ---- value function of grid point hi + 1 -> registers / LDS: the terminal one (P_N = Qxx_N, s_N = -lx_N,
  //      riccati_recursion.cpp:37-38) or the one a previous segment left in the Riccati records ----
  d4 pp[T][T];
  {
    const bool term = (hi == N - 1);
    const double* psrc = term ? (a.kkt + kinst + (size_t)N * KL.stride + KL.off[RTOC_KKT_QXX])
                              : (a.ric + rinst + (size_t)(hi + 1) * RL.stride + RL.off[RTOC_RIC_P]);
    const double* ssrc = term ? (a.kkt + kinst + (size_t)N * KL.stride + KL.off[RTOC_KKT_LX])
                              : (a.ric + rinst + (size_t)(hi + 1) * RL.stride + RL.off[RTOC_RIC_S]);
#pragma unroll
    for (int kt = 0; kt < T; ++kt)
#pragma unroll
      for (int mt = 0; mt < T; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * kt + 4 * r + q, j = 16 * mt + li;
          const bool ok = i < NX && j < NX;
          const double v = psrc[ok ? j + i * NX : 0];   // (symmetric: the mirror element, li along the contiguous index)
          pp[kt][mt][r] = ok ? v : 0.0;
        }
    for (int e = lane; e < NX; e += 64) sS[e] = term ? -ssrc[e] : ssrc[e];
    if (term && W == 0) {
      double* rr = a.ric + rinst + (size_t)N * RL.stride;
      const d2* s2 = reinterpret_cast<const d2*>(psrc);
      d2* t2 = reinterpret_cast<d2*>(rr + RL.off[RTOC_RIC_P]);
      for (int e = lane; e < NX * NX / 2; e += 64) t2[e] = s2[e];
      for (int e = lane; e < NX; e += 64) rr[RL.off[RTOC_RIC_S] + e] = -ssrc[e];
    }
  }
  // ---- operands that come straight from HBM, requested one stage ahead: Bv fragments (B operand of PB, A operand of G), Quu in the C
  //      layout.  RAW: lanes beyond the matrices load a clamped address and are masked where the value is USED ----
  // (Bv is NOT requested a stage ahead here: twenty registers carried over the stage boundary, where P+ is being rebuilt, were spilled
  //  load by load -- each with a full wait; they are requested at the stage top, ahead of z)
  auto issue_bq = [&](int stage) __attribute__((always_inline)) {
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride;
    // Quu -> the LDS place of G by DMA (flat: G's leading dimension is NU); the G product adds itself onto it
    constexpr int CQ = (NU * NU + 1) / 2, PQ = (CQ + 63) / 64;
    const double* gp = kp + KL.off[RTOC_KKT_QUU];
#pragma unroll
    for (int p = 0; p < PQ; ++p) {
      const int n = lane + 64 * p;
      if (64 * (p + 1) <= CQ || n < CQ) __builtin_amdgcn_global_load_lds(gp + 2 * n, ldsp(sG + 128 * p), 16, 0, 0);
    }
  };
  // Qxx comes in by DMA too, one column tile (16 columns, all rows: 8 KB, contiguous in the record) at a time into one of two padded
  // panels in the zone that is dead between the policy products and the transposes; F accumulates from zero and takes its start
  // value panel by panel at the end of the column loop's iterations (the same sum in another order; the symmetrisation of
  // brrf.cpp:85 folded in): panel p holds the transposed elements of the tiles (p, t >= p) -- read with li along the contiguous
  // index -- and the direct elements of the tiles (c < p, p).  No registers wait for these 32 KB.
  constexpr int LDQ = M::LDQ;
  d4 f[T][T];
  auto issue_qxx_panel = [&](const double* kr_, int p) __attribute__((always_inline)) {
    const double* qb_ = kr_ + KL.off[RTOC_KKT_QXX] + (size_t)16 * p * NX;
    double* dst = pw_ + M::P_Q + (p & 1) * M::PANEL;
#pragma unroll
    for (int col = 0; col < 16; ++col) {
      if (16 * p + col >= NX) continue;
      if (lane < NX / 2) __builtin_amdgcn_global_load_lds(qb_ + col * NX + 2 * lane, ldsp(dst + col * LDQ), 16, 0, 0);
    }
  };
  auto qxx_seed_panel = [&](int p) __attribute__((always_inline)) {
    const double* pan = pw_ + M::P_Q + (p & 1) * M::PANEL;
#pragma unroll
    for (int t = p; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // tile (p, t): element (16p + 4r + q, 16t + li) <- Qxx[16t + li][16p + 4r + q], column 4r + q of the panel
        if (!own(t)) continue;
        const int i = 16 * p + 4 * r + q, j = 16 * t + li;
        if (16 * p + 4 * r >= NX) continue;
        const bool ok = ((16 * p + 4 * r + 3 < NX) || i < NX) && ((16 * t + 15 < NX) || j < NX);
        const double v = pan[(ok ? j : 0) + (ok ? 4 * r + q : 0) * LDQ];
        f[p][t][r] = __builtin_fma((t == p) ? 1.0 : 0.5, ok ? v : 0.0, f[p][t][r]);
      }
#pragma unroll
    for (int c = 0; c < p; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // tile (c, p): element (16c + 4r + q, 16p + li) <- Qxx[16c + 4r + q][16p + li], column li of the panel
        if (!own(p)) continue;
        const int j = 16 * p + li;
        const bool ok = (16 * p + 15 < NX) || j < NX;
        const double v = pan[16 * c + 4 * r + q + (ok ? li : 0) * LDQ];
        f[c][p][r] = __builtin_fma(0.5, ok ? v : 0.0, f[c][p][r]);
      }
  };
  if (W == 0) {
    issue_dma_strip(hi);
    issue_dma_bv(hi);
    issue_dma_A(hi);
    for (int e = lane; e < a.nstages; e += 64) sGrid[e] = a.grid[e].type | (a.grid[e].dims << 8);   // (the host keeps nstages <= RV_MAX_STAGES)
  }
  if (W == 0) issue_bq(hi);
  else issue_dma_qxu(hi);
  // z = s+ - P+ Fx (brrf.cpp:86) of the first grid point of the segment: per-lane partial sums over the rows a lane holds, q-reduction
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  rw2_sync();   // the strip of grid point hi has landed (wave 0's DMA)
  if (W == 0) {
    double fxr[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const double v = smem[M::OFF_ST + (hi & 1) * M::STRIP + 4 * g + q];
      fxr[g] = (4 * g + 3 < NX || 4 * g + q < NX) ? v : 0.0;
    }
#pragma unroll
    for (int mt = 0; mt < T; ++mt) {
      double part = 0.0;
#pragma unroll
      for (int g = 0; g < KG; ++g) part = __builtin_fma(pp[g / 4][mt][g % 4], fxr[g], part);
      part = qsum(part);
      const int j = 16 * mt + li;
      const double sv = sS[(j < NX) ? j : 0];
      if (q == 0 && j < NX) sZ[j] = sv - part;
    }
  }

  for (int st = hi; st >= lo; --st) {
    lane = lane0;
    asm volatile("" : "+v"(lane));   // opaque per stage: keeps LICM from pinning the per-lane addresses of every unrolled loop
    lane &= 63;
    li = lane & 15;
    q = lane >> 4;
    // what PB and G read has landed: wave 0's DMA of the strip, Bv and Quu.  A (35 KB) and Qxu are WAVE 1's requests: it waits for
    // them itself ahead of H, behind PB; wave 0 reads A behind the hand-over barrier, which wave 1 reaches later still
    if (W == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rw2_sync();
    const int gword = __builtin_amdgcn_readfirstlane(sGrid[st]);
    const bool impact = (gword & 0xff) == RTOC_GRID_IMPACT;   // (read behind the barrier below: wave 0 writes the table)
    const double* kr = a.kkt + kinst + (size_t)st * KL.stride;
    double* rr = a.ric + rinst + (size_t)st * RL.stride;

    const double* const sStrip = smem + M::OFF_ST + (st & 1) * M::STRIP;          // Fx | lx | lu of this grid point
    const double* const sStripN = smem + M::OFF_ST + ((st - 1) & 1) * M::STRIP;   // ... of the next one
    if (W == 0 && st > lo) issue_dma_strip(st - 1);   // (into the other strip: it has a whole stage to land)
    RW2_PROF(0);
    // H^T = Qxu^T + PB^T A (rows u = 16 tu + 4r + q, columns x = 16c + li): all of it in wave 1, handed to wave 0 through LDS.  Qxu^T
    // is the START VALUE of the accumulators, read from the LDS place wave 1's DMA brought it to
    d4 hT[TU][T];
    auto seed = [&](int tu) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < T; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u = 16 * tu + 4 * r + q, x = 16 * c + li;
          const bool ok = u < NU && x < NX;
          const double v = sQxu[ok ? x + u * NX : 0];
          hT[tu][c][r] = ok ? v : 0.0;
        }
    };
    // ---- P of grid point st + 1 -> HBM from the registers that hold it as P+ all stage long, at the stage top: the stores drain
    //      behind the PB / G / H products and never stand between a Qxx panel and its counted wait (element (i, j) through its
    //      mirror (j, i): li along the contiguous index).  Row tiles [0, PSPLIT) by wave 0, the rest by wave 1: what evens out
    //      their ways to the hand-over ----
    if (st < hi) {
      double* pw = a.ric + rinst + (size_t)(st + 1) * RL.stride + RL.off[RTOC_RIC_P];
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mt = 0; mt < T; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * t + 4 * r + q, j = 16 * mt + li;
            if ((t < PSPLIT) == (W == 0) && i < NX && j < NX) pw[j + i * NX] = pp[t][mt][r];
          }
    }
    RW2_PROF(1);
    // (1. z = s+ - P+ Fx of this grid point was formed where P+ was rebuilt: at the end of the previous stage / in the prologue)
    auto zrow = [&](int g) __attribute__((always_inline)) -> double {   // z in the row layout: z[4g + q] (re-read where it rides: no registers held)
      const double v = sZ[4 * g + q];
      return (4 * g + 3 < NX || 4 * g + q < NX) ? v : 0.0;
    };

    RW2_PROF(2);
    // ---- the O(n^2 nu) part, split by ROLE: wave 0 forms G (it needs the rows v of PB only), factorises it and solves for t, k;
    //      wave 1 forms all of PB and H^T beside it.  They meet once: Y = L^-1, t and A^T z cross over, H^T's column tiles of wave 0
    //      through the place of wave 1's G.  From there each wave forms Z^T, K, Z Z^T and s for its own column tiles (wave 1 Z^T for
    //      all of them: its tiles (c, t >= 3) of Z Z^T meet every column) ----
    constexpr int C0 = (W == 0) ? G0 / 4 : 0;   // first row tile of PB this wave forms
    d4 acc[T][TU];   // PB = P+[:, v] Bv: rows x = 16c + .., columns u = 16 tu + li; column NU: z
#pragma unroll
    for (int c = 0; c < T; ++c)
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) acc[c][tu] = zero4();
    if (!impact) {
      // ================= 2. PB, G = Quu + Bv^T PB[v, :] (rider column NU: Bv^T z_v) =================
      double bvf[KG - G0][TU];
#pragma unroll
      for (int g = G0; g < KG; ++g)
#pragma unroll
        for (int tu = 0; tu < TU; ++tu) {
          const int k = 4 * g + q - NV, u = 16 * tu + li;
          const bool ok = k >= 0 && k < NV && u < NU;
          const double v = smem[M::OFF_BV + (ok ? k + u * NV : 0)];   // Bv[k][u] (flat, as in the record)
          bvf[g - G0][tu] = ok ? v : 0.0;
        }
#pragma unroll
      for (int g = G0; g < KG; ++g)
#pragma unroll
        for (int tu = 0; tu < TU; ++tu)
#pragma unroll
          for (int c = C0; c < T; ++c) acc[c][tu] = mfma16(pp[g / 4][c][g % 4], bvf[g - G0][tu], acc[c][tu]);
      if constexpr (W == 0) {
        d4 gacc[TU][TU];
#pragma unroll
        for (int tr = 0; tr < TU; ++tr)
#pragma unroll
          for (int tu = 0; tu < TU; ++tu) gacc[tr][tu] = zero4();
#pragma unroll
        for (int g = G0; g < KG; ++g)
#pragma unroll
          for (int tu = 0; tu < TU; ++tu) {
            double bop = acc[g / 4][tu][g % 4];
            if (tu == TU - 1) bop = (li == NUC) ? zrow(g) : bop;
#pragma unroll
            for (int tr = 0; tr < TU; ++tr) gacc[tr][tu] = mfma16(bvf[g - G0][tr], bop, gacc[tr][tu]);
          }
#pragma unroll
        for (int tr = 0; tr < TU; ++tr)
#pragma unroll
          for (int tu = 0; tu < TU; ++tu)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int u0 = 16 * tr + 4 * r + q, u1 = 16 * tu + li;
              if (u0 < NU && u1 < NU) sG[u0 + u1 * NU] += gacc[tr][tu][r];
              if (tu == TU - 1 && u0 < NU && li == NUC) sLup[u0] = sStrip[ST_LUO + u0] - gacc[tr][tu][r];   // lu' = lu - Bv^T z_v
            }
        rv_lds_sync();
      }
    }
    RW2_PROF(3);
    double ca = 0.0, cc = 0.0;   // A[NP][NP], A[NP][NV + NP] (row NP is a corner-group row: staged)
    if (W == 1) {                // A and Qxu have landed (this wave's DMA)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      ca = sA[NP_ + NP_ * LDA];
      cc = sA[NP_ + (NV + NP_) * LDA];
    }
    // structured rows of A^T [.] for one column of C tiles (rows = state rows): dst[k] += ca src[k], dst[NV + k] += cc src[k],
    // k in [NP, NV); NV + k lies TS tiles, RS registers and QS q-groups below k (tests/rw_lane_model.py: struct_rows_add)
    auto struct_rows_add = [&](d4(&dst)[T], const d4(&src)[T], int cmax) __attribute__((always_inline)) {
      d4 rot[T];
#pragma unroll
      for (int c = 0; c < T; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = 4 * c + r;
          const bool used = 4 * e + 3 >= NP_ && 4 * e < NV;   // the register holds a row of [NP, NV)
          rot[c][r] = (QS != 0 && used) ? __shfl(src[c][r], (lane + 16 * (4 - QS)) & 63, 64) : src[c][r];
        }
#pragma unroll
      for (int c = 0; c < T; ++c) {
        if (c > cmax) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = 4 * c + r, i = 4 * e + q;
          if (4 * e + 3 >= NP_ && 4 * e < NV) {
            const double coef = (i >= NP_ && i < NV) ? ca : 0.0;
            dst[c][r] = __builtin_fma(coef, src[c][r], dst[c][r]);
          }
          if (4 * e + 3 >= NV + NP_ && 4 * e < NX) {
            const int ehi = 4 * (c - TS) + r - RS, elo = ehi - 1;
            const double vhi = (ehi >= 0 && ehi < 4 * T) ? rot[(ehi >= 0 ? ehi : 0) / 4][(ehi >= 0 ? ehi : 0) % 4] : 0.0;
            const double vlo = (QS != 0 && elo >= 0 && elo < 4 * T) ? rot[(elo >= 0 ? elo : 0) / 4][(elo >= 0 ? elo : 0) % 4] : 0.0;
            const double got = (QS == 0 || q >= QS) ? vhi : vlo;
            const int k = i - NV;
            const double coef = (k >= NP_ && k < NV && i < NX) ? cc : 0.0;
            dst[c][r] = __builtin_fma(coef, got, dst[c][r]);
          }
        }
      }
    };
    // A fragment of k group g, column tile c: A[4g + q][16c + li], the B operand of [.] A and the A operand of A^T [.] alike
    auto afrag = [&](int g, int c) __attribute__((always_inline)) -> double {
      const int j = 16 * c + li;
      const double v = sA[4 * C::cg(g) + q + ((16 * c + 15 < NX || j < NX) ? j : NX - 1) * LDA];
      const int k = 4 * g + q;
      const bool rowok = group_all_dense(g) || (k < NP_ || (k >= NV && k < NX));
      return (rowok && (16 * c + 15 < NX || j < NX)) ? v : 0.0;
    };
    // structured rows of k group g as a B fragment of column tile c: B[k = 4g + q][x = 16c + li] = ca (x == k) + cc (x == NV + k), k in [NP, NV)
    auto struct_hit = [](int g, int c) {
      const int k0 = (4 * g > NP_) ? 4 * g : NP_, k1 = (4 * g + 3 < NV - 1) ? 4 * g + 3 : NV - 1;
      if (k0 > k1) return false;
      return (k0 <= 16 * c + 15 && k1 >= 16 * c) || (NV + k0 <= 16 * c + 15 && NV + k1 >= 16 * c);
    };
    auto struct_frag = [&](int g, int c) __attribute__((always_inline)) -> double {
      const int k = 4 * g + q, x = 16 * c + li;
      const double v = (x == k) ? ca : (x == NV + k) ? cc : 0.0;
      return (k >= NP_ && k < NV) ? v : 0.0;
    };

    if constexpr (W == 1) {
      // z into the idle column NU of PB: row NU of H^T = PB^T A is then (A^T z)^T (on an impact grid point PB is that column alone)
#pragma unroll
      for (int g = 0; g < KG; ++g) acc[g / 4][TU - 1][g % 4] = (li == NUC) ? zrow(g) : acc[g / 4][TU - 1][g % 4];
      // ================= 4. H^T DIRECTLY in the layout Z^T = Y H^T reads: PB's C tiles are the A operand as they stand, A's fragments
      //                      the B operand; the structured rows k of A (ca e_k | cc e_NV+k) are synthesised fragments that meet at
      //                      most three column tiles.  No transposes, no rotations (tests/rw_lane_model.py: direct_ht).  On an
      //                      impact grid point (riccati_factorizer.cpp:178-197: no controls) the rider row alone =================
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) {
        if (impact && tu < TU - 1) continue;
        asm volatile("" ::: "memory");   // (A's fragments are re-read per control tile: held across the tiles they would not fit beside P+)
        if (!impact) {
          seed(tu);
        } else {
#pragma unroll
          for (int c = 0; c < T; ++c) hT[tu][c] = zero4();
        }
#pragma unroll
        for (int g = 0; g < KG; ++g) {
          const double aop = acc[g / 4][tu][g % 4];
#pragma unroll
          for (int c = 0; c < T; ++c) {
            const bool hit = struct_hit(g, c);
            if (!group_dense(g) && !hit) continue;
            double bop = group_dense(g) ? afrag(g, c) : 0.0;
            if (hit) bop += struct_frag(g, c);   // (afrag is zero on the structured rows)
            hT[tu][c] = mfma16(aop, bop, hT[tu][c]);
          }
        }
        if (tu == TU - 1) {   // the rider: w0 = A^T z; its row leaves H^T (rows u >= NU are zero otherwise: Bv's fragments are masked)
#pragma unroll
          for (int c = 0; c < T; ++c) {
            const int x = 16 * c + li;
            if (q == NUC % 4 && x < NX) sW0[x] = hT[tu][c][NUC / 4];
            hT[tu][c][NUC / 4] = (q == NUC % 4) ? 0.0 : hT[tu][c][NUC / 4];
          }
        }
      }
      if (!impact) {   // (behind the last read of Qxu: the hand-over takes its place)
        rv_lds_sync();
#pragma unroll
        for (int tu = 0; tu < TU; ++tu)
#pragma unroll
          for (int c = 0; c < T; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int u = 16 * tu + 4 * r + q;
              if (u < NU) {
                if (c <= TSPLIT) sHX[u * LDH + 16 * c + li] = hT[tu][c][r];
                else sHB[u * LDB + 16 * (c - TSPLIT - 1) + li] = hT[tu][c][r];
              }
            }
      }
      RW2_PROF(4);
    } else {
      // ================= 3. LLT(G) (riccati_factorizer.cpp:49), Y = L^-1.  t = Y lu' and k = -Y^T t ride in Z^T = Y H^T and
      //                      K = -Y^T Z^T: lu' is the idle column NX of H^T's last column tile (the hand-over below) =================
      if (!impact) {
        if (wave_llt_inv_blocked<NU, NU>(sG, sG, pw_ + M::P_LINV, sY, scr, lane)) stat |= RTOC_STAT_QUU_NOT_SPD;
        rv_lds_sync();
        RW2_PROF(5);
      }
    }
    RW2_PROF(6);
    // ---- the hand-over: behind the first barrier each wave takes what it needs of the other's block into registers / its own block,
    //      behind the second one nobody reads the other's block any more (its Qxx panels will land there) ----
    rw2_sync();
    double ya[TU][KGU];   // wave 1's copy of Y's fragments as the A operand of Z^T = Y H^T (five column tiles pass by them)
    auto yaf = [&](int tu, int gj) __attribute__((always_inline)) -> double {
      const int m = 16 * tu + li, k = 4 * gj + q;
      const bool ok = m < NU && k < NU;
      const double v = sY[(ok ? m : 0) + (ok ? k : 0) * NU];
      return ok ? v : 0.0;
    };
    auto ykf = [&](int tu, int gj) __attribute__((always_inline)) -> double {   // -Y[k][m], the A operand of K = -Y^T Z^T (wave 0)
      const int m = 16 * tu + li, k = 4 * gj + q;
      const bool ok = m < NU && k < NU;
      const double v = sY[(ok ? k : 0) + (ok ? m : 0) * NU];
      return ok ? -v : 0.0;
    };
    if constexpr (W == 1) {
      if (!impact) {
#pragma unroll
        for (int tu = 0; tu < TU; ++tu)
#pragma unroll
          for (int gj = 0; gj < KGU; ++gj) ya[tu][gj] = (gj < 4 * (tu + 1)) ? yaf(tu, gj) : 0.0;
        // lu' (wave 0 formed it beside G) -> column NX of H^T: in this wave's registers and in wave 0's copy of the last column tile
#pragma unroll
        for (int tu = 0; tu < TU; ++tu)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = 16 * tu + 4 * r + q;
            const double v = pw0[M::P_LUP + ((u < NU) ? u : 0)];
            hT[tu][T - 1][r] = (li == RID && u < NU) ? v : hT[tu][T - 1][r];
          }
        if (lane < NU) sHB[lane * LDB + 16 * (T - TSPLIT - 2) + RID] = pw0[M::P_LUP + lane];
      }
    } else {
      if (!impact) {
#pragma unroll
        for (int tu = 0; tu < TU; ++tu)
#pragma unroll
          for (int c = 0; c <= TSPLIT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int u = 16 * tu + 4 * r + q;
              const double v = sHX[((u < NU) ? u : 0) * LDH + 16 * c + li];
              hT[tu][c][r] = (u < NU) ? v : 0.0;
            }
      }
      if (!S_W1)
        for (int e = lane; e < NX; e += 64) sW0[e] = pw1[M::P_W0 + e];
      ca = sA[NP_ + NP_ * LDA];   // (wave 1 waited for A ahead of H)
      cc = sA[NP_ + (NV + NP_) * LDA];
    }
    rw2_sync();
    // The panels this wave needs: every p that meets one of its tiles -- (p, t >= p) with t its own, (c < p, p) with p its own --,
    // i.e. 0 .. its last column tile.  Panel k is consumed at slot k of the sequence [behind s; then, per column iteration of this
    // wave: behind its W product, at its end]; panel k + 2 takes its buffer there (two panels in flight).
    constexpr int NPAN = (W == 0) ? TSPLIT + 1 : T;
    auto panel_slot = [&](int k) __attribute__((always_inline)) {
      if (k >= NPAN) return;
      // panel k has landed: only the DMA instructions of panel k + 1 -- issued one slot ago, nothing behind them -- may be in flight
      if (k + 1 < NPAN && 16 * (k + 2) <= NX) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (k + 1 < NPAN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NX % 16) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      qxx_seed_panel(k);
      if (k + 2 < NPAN) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        issue_qxx_panel(kr, k + 2);
      }
    };
    // F accumulates from ZERO (-Z Z^T, then A^T W column by column) and takes its start value Qxx (upper tiles; off-diagonal ones
    // symmetrised: brrf.cpp:85 folded in) panel by panel.  Wave 1's block is dead from here (the hand-over has been read): its first
    // panel now; wave 0's behind its last read of Y
    if (W == 1) issue_qxx_panel(kr, 0);   // (panel 1 covers A^T z and t, which the update of s below still reads: behind it)
#pragma unroll
    for (int c = 0; c < T; ++c)
#pragma unroll
      for (int t = c; t < T; ++t)
        if (own(t)) f[c][t] = zero4();
    // Z^T = Y H^T of one column tile (Y lower triangular: the tile above the diagonal is skipped); K = -Y^T Z^T of it -> HBM (row-major)
    auto zt_col = [&](d4(&zc)[TU], const d4(&hc)[TU]) __attribute__((always_inline)) {
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) {
        zc[tu] = zero4();
#pragma unroll
        for (int gj = 0; gj < KGU; ++gj) {
          if (gj >= 4 * (tu + 1)) continue;
          zc[tu] = mfma16((W == 1) ? ya[tu][gj] : yaf(tu, gj), hc[gj / 4][gj % 4], zc[tu]);
        }
      }
    };
    auto t_out = [&](const d4(&zc)[TU]) __attribute__((always_inline)) {   // the rider of the last column tile: t = Y lu' -> this wave's sT
#pragma unroll
      for (int tu = 0; tu < TU; ++tu)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u = 16 * tu + 4 * r + q;
          if (li == RID && u < NU) sT[u] = zc[tu][r];
        }
    };
    double chk = 0.0;
    auto k_col = [&](int c, const d4(&zc)[TU]) __attribute__((always_inline)) {
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) {
        d4 kk = zero4();
#pragma unroll
        for (int gj = 0; gj < KGU; ++gj) {
          if (gj < 4 * tu) continue;
          kk = mfma16(ykf(tu, gj), zc[gj / 4][gj % 4], kk);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u = 16 * tu + 4 * r + q, x = 16 * c + li;
          if (u < NU && x < NX) rr[RL.off[RTOC_RIC_K] + u * NX + x] = kk[r];
          if (c == T - 1 && u < NU && li == RID) rr[RL.off[RTOC_RIC_KV] + u] = kk[r];   // the rider: k = -Y^T t
          chk = __builtin_fma(kk[r], 0.0, chk);
        }
      }
    };
    // s = A^T z - lx - H k = w0 - lx + Z t (brrf.cpp:86-90) of one column tile, by WAVE 1 for all of them (every column tile of Z^T
    // passes through its registers): column layout, per-lane partial sums + q-reduction; -> LDS (the s+ of the next grid point) and
    // HBM, with the (zero) switching-time fields of the record
    double tr_[KGU];   // t in the row layout
    auto s_col = [&](int c, const d4(&zc)[TU]) __attribute__((always_inline)) {
      double part = 0.0;
#pragma unroll
      for (int gu = 0; gu < KGU; ++gu) part = __builtin_fma(zc[gu / 4][gu % 4], tr_[gu], part);
      part = qsum(part);
      const int j = 16 * c + li;
      const int jc = (j < NX) ? j : 0;
      const double sn = sW0[jc] - sStrip[ST_LXO + jc] + part;
      if (q == 0 && j < NX) {
        sS[j] = sn;
        rr[RL.off[RTOC_RIC_S] + j] = sn;
      }
      if (q == 1 && j < NX) rr[RL.off[RTOC_RIC_PSI] + j] = 0.0;
      if (q == 2 && j < NX) rr[RL.off[RTOC_RIC_PHI] + j] = 0.0;
    };
    d4 zt[TU][T];   // Z^T = Y H^T: the column tiles this wave owns stay
    if (!impact) {
      if constexpr (W == 0) {
        // ================= 6. / 7. wave 0: Z^T of its column tiles, and K = -Y^T Z^T (riccati_factorizer.cpp:55) of ALL column tiles --
        //                          wave 1's pass through (H^T from the place of Bv); the idle column NX of the last one carries
        //                          lu' -> t = Y lu' -> k = -Y^T t =================
#pragma unroll
        for (int c = 0; c < T; ++c) {
          d4 hc[TU], zc[TU];
#pragma unroll
          for (int tu = 0; tu < TU; ++tu) {
            if (c <= TSPLIT) {
              hc[tu] = hT[tu][c];
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int u = 16 * tu + 4 * r + q;
                const double v = sHB[((u < NU) ? u : 0) * LDB + 16 * (c - TSPLIT - 1) + li];
                hc[tu][r] = (u < NU) ? v : 0.0;
              }
            }
          }
          zt_col(zc, hc);
          k_col(c, zc);
          if (!S_W1 && c == T - 1) t_out(zc);
          if (c <= TSPLIT) {
#pragma unroll
            for (int tu = 0; tu < TU; ++tu) zt[tu][c] = zc[tu];
          }
        }
        if (is_bad(chk)) stat |= RTOC_STAT_NAN;
      } else {
        // ================= 6. / 8. wave 1: Z^T of its own column tiles first (they stay; the rider of the last one is t), F -= Z Z^T
        //                          (brrf.cpp:82-84) among them; then the column tiles of wave 0 one at a time: each meets wave 1's
        //                          tiles of its row, gives its entries of s, and dies =================
#pragma unroll
        for (int c = TSPLIT + 1; c < T; ++c) {
          d4 hc[TU], zc[TU];
#pragma unroll
          for (int tu = 0; tu < TU; ++tu) hc[tu] = hT[tu][c];
          zt_col(zc, hc);
          if (c == T - 1) t_out(zc);
#pragma unroll
          for (int tu = 0; tu < TU; ++tu) zt[tu][c] = zc[tu];
        }
#pragma unroll
        for (int gu = 0; gu < KGU; ++gu)
#pragma unroll
          for (int c = TSPLIT + 1; c < T; ++c)
#pragma unroll
            for (int t = c; t < T; ++t) f[c][t] = mfma16(-zt[gu / 4][c][gu % 4], zt[gu / 4][t][gu % 4], f[c][t]);
        if (S_W1) {
          rv_lds_sync();
#pragma unroll
          for (int gu = 0; gu < KGU; ++gu) {
            const double v = sT[4 * gu + q];
            tr_[gu] = (4 * gu + q < NU) ? v : 0.0;
          }
        }
#pragma unroll
        for (int c = 0; c <= TSPLIT; ++c) {
          d4 hc[TU], zc[TU];
#pragma unroll
          for (int tu = 0; tu < TU; ++tu) hc[tu] = hT[tu][c];
          zt_col(zc, hc);
#pragma unroll
          for (int gu = 0; gu < KGU; ++gu)
#pragma unroll
            for (int t = TSPLIT + 1; t < T; ++t) f[c][t] = mfma16(-zc[gu / 4][gu % 4], zt[gu / 4][t][gu % 4], f[c][t]);
          if (S_W1) s_col(c, zc);
        }
      }
    } else {
#pragma unroll
      for (int tu = 0; tu < TU; ++tu)
#pragma unroll
        for (int c = 0; c < T; ++c)
          if (own(c)) zt[tu][c] = zero4();
    }
    RW2_PROF(8);
    if constexpr (W == 0) {
      // ================= 8. wave 0: G, Y and the scratch are dead: the first panel of Qxx; F -= Z Z^T (brrf.cpp:82-84) =================
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      issue_qxx_panel(kr, 0);
      if (!impact) {
#pragma unroll
        for (int gu = 0; gu < KGU; ++gu)
#pragma unroll
          for (int c = 0; c <= TSPLIT; ++c)
#pragma unroll
            for (int t = c; t <= TSPLIT; ++t) f[c][t] = mfma16(-zt[gu / 4][c][gu % 4], zt[gu / 4][t][gu % 4], f[c][t]);
      }
    }
    RW2_PROF(9);
    // ---- s of this grid point: wave 1's own column tiles (wave 0's passed by above); on an impact grid point s = A^T z - lx ----
    if constexpr (!S_W1) {
      rv_lds_sync();
#pragma unroll
      for (int gu = 0; gu < KGU; ++gu) {
        const double v = sT[4 * gu + q];
        tr_[gu] = (!impact && 4 * gu + q < NU) ? v : 0.0;
      }
#pragma unroll
      for (int c = 0; c < T; ++c) {
        if (!own(c)) continue;
        d4 zc[TU];
#pragma unroll
        for (int tu = 0; tu < TU; ++tu) zc[tu] = zt[tu][c];
        s_col(c, zc);
      }
      if (W == 0 && lane < 5) rr[RL.off[RTOC_RIC_SCAL] + lane] = 0.0;
    } else if constexpr (W == 1) {
      rv_lds_sync();
      d4 z0[TU];
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) z0[tu] = zero4();
      if (impact) {
#pragma unroll
        for (int gu = 0; gu < KGU; ++gu) tr_[gu] = 0.0;
#pragma unroll
        for (int c = 0; c <= TSPLIT; ++c) s_col(c, z0);
      }
#pragma unroll
      for (int c = TSPLIT + 1; c < T; ++c) {
        d4 zc[TU];
#pragma unroll
        for (int tu = 0; tu < TU; ++tu) zc[tu] = zt[tu][c];
        s_col(c, zc);
      }
    } else {
      if (lane < 5) rr[RL.off[RTOC_RIC_SCAL] + lane] = 0.0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (NPAN > 1) issue_qxx_panel(kr, 1);
    asm volatile("" ::: "memory");
    panel_slot(0);

    RW2_PROF(10);
    // ================= 9. column tile by column tile: W[:, t] = P+ A[:, t], F[c][t] += A^T[c] W[:, t] =================
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (!own(t)) continue;   // this wave's column tiles of F
      constexpr int T0 = (W == 0) ? 0 : TSPLIT + 1;
      const int ord = t - T0;   // which of this wave's iterations
      d4 w[T];
#pragma unroll
      for (int tm = 0; tm < T; ++tm) w[tm] = zero4();
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        if (!group_dense(g)) continue;
        const double bv = afrag(g, t);
#pragma unroll
        for (int tm = 0; tm < T; ++tm) w[tm] = mfma16(pp[g / 4][tm][g % 4], bv, w[tm]);
      }
      {
        // structured rows k of A: W[:, k] += ca P+[:, k] (same tile / lane), W[:, NV + k] += cc P+[:, k] (TS tiles, LS lanes to the left)
        const int j = 16 * t + li, k = j - NV;
        const double ca_l = (j >= NP_ && j < NV) ? ca : 0.0;
        const double cc_l = (k >= NP_ && k < NV && j < NX) ? cc : 0.0;
#pragma unroll
        for (int tm = 0; tm < T; ++tm)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (16 * t + 15 >= NP_ && 16 * t < NV) w[tm][r] = __builtin_fma(ca_l, pp[tm][t][r], w[tm][r]);
            if (16 * t + 15 >= NV + NP_ && 16 * t < NX) {
              double src = 0.0;
              if constexpr (LS == 0) {
                if (t - TS >= 0) src = pp[tm][(t - TS >= 0) ? t - TS : 0][r];
              } else {
                if (t - TS >= 0) src = dpp_from_left<LS>(pp[tm][(t - TS >= 0) ? t - TS : 0][r]);
                if (t - TS - 1 >= 0) src += dpp_from_right<16 - LS>(pp[tm][(t - TS - 1 >= 0) ? t - TS - 1 : 0][r]);
              }
              w[tm][r] = __builtin_fma(cc_l, src, w[tm][r]);
            }
          }
      }
      panel_slot(1 + 2 * ord);
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        if (!group_dense(g)) continue;
        const double bw = w[g / 4][g % 4];
#pragma unroll
        for (int c = 0; c <= t; ++c) f[c][t] = mfma16(afrag(g, c), bw, f[c][t]);
      }
      {
        d4 col[T];
#pragma unroll
        for (int c = 0; c < T; ++c) col[c] = (c <= t) ? f[c][t] : zero4();
        struct_rows_add(col, w, t);
#pragma unroll
        for (int c = 0; c <= t; ++c) f[c][t] = col[c];
      }
      panel_slot(2 + 2 * ord);
    }
    RW2_PROF(11);
    // ================= 10. the fifteen tiles of F meet in LDS, in the place of A; each wave rebuilds P+ = sym(F) from there ==========
    auto masked = [&](int c, int t, int r, double v) -> double {
      const int i = 16 * c + 4 * r + q, j = 16 * t + li;
      if (16 * c + 4 * r >= NX) return 0.0;
      const bool rowok = (16 * c + 4 * r + 3 < NX) || i < NX;
      const bool colok = (16 * t + 15 < NX) || j < NX;
      return (rowok && colok) ? v : 0.0;
    };
    auto tile_at = [](int c, int t) constexpr { return c * T - c * (c - 1) / 2 + (t - c); };   // index of the upper tile (c, t), c <= t
    if (W == 0 && st > lo) {   // Bv and Quu of the next grid point: their places (wave 1's H^T, this wave's last panel) have been read
      issue_dma_bv(st - 1);
      issue_bq(st - 1);
    }
    rw2_sync();   // both waves have read A -- and the strip -- for the last time
#pragma unroll
    for (int c = 0; c < T; ++c)
#pragma unroll
      for (int t = c; t < T; ++t) {
        if (!own(t)) continue;
        double* x_ = sA + tile_at(c, t) * SCR_TILE;
#pragma unroll
        for (int r = 0; r < 4; ++r) x_[(q + 4 * r) * SCR_LD + li] = masked(c, t, r, f[c][t][r]);
      }
    RW2_PROF(12);
    rw2_sync();   // all of F is in LDS, and the next strip
    // z = s+ - P+ Fx of the NEXT grid point rides along (one copy in LDS, each wave the entries of its column tiles): the elements of
    // P+ pass through the vector registers once, here -- formed at the stage top from the resident P+ it made the allocator keep all
    // 200 registers of it in the architectural half
    double zpart[T];
#pragma unroll
    for (int mt = 0; mt < T; ++mt) zpart[mt] = 0.0;
#pragma unroll
    for (int kt = 0; kt < T; ++kt)
#pragma unroll
      for (int mt = 0; mt < T; ++mt) {
        const double* x_ = sA + ((kt <= mt) ? tile_at(kt, mt) : tile_at(mt, kt)) * SCR_TILE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // upper tiles as computed (this wave's own: straight from its accumulators), lower ones transposed, diagonal ones mirrored
          // from their upper triangle (P exactly symmetric)
          double pv;
          if (kt < mt) {
            pv = own(mt) ? masked(kt, mt, r, f[kt][mt][r]) : x_[(q + 4 * r) * SCR_LD + li];
          } else if (kt > mt) {
            pv = x_[li * SCR_LD + q + 4 * r];
          } else {
            const double dv = own(mt) ? masked(kt, mt, r, f[kt][mt][r]) : x_[(q + 4 * r) * SCR_LD + li];
            const double tv = x_[li * SCR_LD + q + 4 * r];
            pv = (q + 4 * r <= li) ? dv : tv;
          }
          pp[kt][mt][r] = pv;
          if (own(mt) && 4 * kt + r < KG) {   // (z of this wave's column tiles)   // (Fx re-read where it is used: eighteen registers less while P+ passes through the vector file)
            const double fv = sStripN[4 * (4 * kt + r) + q];   // Fx of the next grid point (its strip was requested at the stage top)
            zpart[mt] = __builtin_fma(pv, (16 * kt + 4 * r + 3 < NX || 16 * kt + 4 * r + q < NX) ? fv : 0.0, zpart[mt]);   // (st == lo: a stale strip, the sum unused)
          }
        }
      }
    if (st > lo) {
#pragma unroll
      for (int mt = 0; mt < T; ++mt) {
        if (!own(mt)) continue;
        const double part = qsum(zpart[mt]);
        const int j = 16 * mt + li;
        const double sv = sS[(j < NX) ? j : 0];
        if (q == 0 && j < NX) sZ[j] = sv - part;
      }
    }
    RW2_PROF(14);
    rw2_sync();   // ... and has been read: the place belongs to A again
    RW2_PROF(15);
    if (st > lo) {
      if (W == 1) {
        issue_dma_qxu(st - 1);
        issue_dma_A(st - 1);
      }
    }
    asm volatile("" ::: "memory");
    RW2_PROF(13);
  }

  // ---- P of the last grid point of the segment ----
  {
    double* pw = a.ric + rinst + (size_t)lo * RL.stride + RL.off[RTOC_RIC_P];
#pragma unroll
    for (int kt = 0; kt < T; ++kt)
#pragma unroll
      for (int mt = 0; mt < T; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * kt + 4 * r + q, j = 16 * mt + li;
          if (own(kt) && i < NX && j < NX) pw[j + i * NX] = pp[kt][mt][r];
        }
  }
  if (stat) atomicOr(&a.status[b], stat);
}

template <int NV, int NU, int NS>
__global__ __launch_bounds__(128, 1) void riccati_backward_rw2_kernel(BwdArgs a) {
  // (wave-uniform branch: each wave runs the body compiled for its share; the barriers inside pair up one to one)
  if (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0) rw2_body<NV, NU, NS, 0>(a);
  else rw2_body<NV, NU, NS, 1>(a);
}

}  // namespace rtoc

// riccati_backward_rw.hpp -- register-resident backward Riccati recursion of the iCub-size shapes (nx = 64 / 70): ONE wavefront
// per OCP instance with the whole 512-entry register file of its SIMD (VGPR + AGPR, one wave per SIMD, four instances per CU).
//
// Same recursion and HBM record contract as riccati_backward.hpp (reference: src/riccati/riccati_recursion.cpp:32-80,
// riccati_factorizer.cpp:44-56,178-197, backward_riccati_recursion_factorizer.cpp:31-91); the machine mapping is the one of
// riccati_backward_rv.hpp carried to T = 4 / 5 state tiles, where the stacked operand [P+; PB^T] no longer fills its tiles and
// no state column is free for the riders:
//
//   * P+ lives in T x T accumulator tiles in the f64 MFMA C layout (lane (li, q), register r <-> row q + 4r, column li);
//     for a symmetric matrix that layout IS the A fragment (and the B fragment) of the products P+ [.]: no LDS copy of P+.
//   * Fxx is taken in its STRUCTURED form (src/dynamics/state_equation.cpp:52-55,80-82, checked on the device by
//     fxx_structure_kernel before the kernel is chosen): rows [NP, NV) are a e_k^T | c e_k^T.  Only the k groups that hold a
//     dense row -- the NP corner rows and the NV velocity rows -- are staged in LDS (21.5 of 32 KB at nv = 32: that is what
//     lets FOUR instances share a CU) and multiplied on the matrix cores; the structured rows are scaled copies of P+ / W tiles
//     in place, TS tiles and LS lanes (rows) away (NV = 16 TS + LS; nv = 32: LS = 0).
//   * PB = P+[:, v] Bv stays in accumulators: as B fragment of G = Quu + Bv^T PB[v, :] and of H = A^T PB, and its idle column NU
//     carries z = s+ - P+ Fx through both: column NU of G is Bv^T z_v (-> lu'), column NU of H is A^T z (-> s).
//   * H is produced rows = state (the structured rows of A^T [.] are row copies there) and transposed tile by tile through a
//     16 x 17 LDS scratch, Qxu^T added on the way (li along the contiguous index of Qxu): H^T is the B fragment of Z^T = Y H^T.
//   * LLT(G), Y = L^-1 by wave_llt_inv_blocked (16 + (NU - 16) columns, trailing update on the matrix cores);
//     K = -Y^T Z^T tile by tile to HBM; F = Qxx - Z Z^T + A^T (P+ A) column tile by column tile; P = sym(F).
//   * z, A^T z, lu', t = Y lu', k = -Y^T t and s = A^T z - lx + Z t are vectors in LDS / per-lane partial sums with a q-reduction
//     (no free column for riders at nx = 64).
//   * Record traffic: the dense k groups of A and the strip Fx | lx | lu by LDS-DMA; Bv, Quu, Qxu^T, Qxx by per-lane loads
//     straight into operand / accumulator registers.
//
// Scope (rw_applies, rtoc_capi.hip): grids without switching-time optimisation; switching-constraint grid points are single
// launches of the tile-split kernel (its one-stage mode), P+ / s+ handed over through the Riccati records -- the host cuts the
// horizon into segments [seg_hi .. seg_lo]; structured Fxx; RTOC_OPT_WRITEBACK_KKT = 0.  tests/rw_lane_model.py states the lane
// algebra in numpy against the oracle (tests/test_rw_lane_model.py).
#pragma once
#include "riccati_backward_rv.hpp"

namespace rtoc {

template <int NV, int NU>
struct RwCfg {
  static constexpr int NX = 2 * NV, NP = NV - NU;
  static constexpr int T = (NX + 15) / 16, TU = (NU + 15) / 16;
  static constexpr int KG = (NX + 3) / 4, KGU = (NU + 3) / 4;
  static constexpr int G0 = NV / 4;                 // first aligned k group that meets the velocity rows [NV, NX)
  static constexpr int G1 = (NP + 3) / 4;           // k groups [0, G1) meet the corner rows [0, NP)
  static constexpr int NDG = G1 + KG - G0;          // k groups with a dense row of A (staged in LDS)
  static constexpr int NUC = NU - 16 * (TU - 1);    // lane of the rider column NU in the last control tile
  static constexpr int TS = NV / 16, LS = NV % 16, QS = LS % 4, RS = LS / 4;
  static constexpr int LDA = lds_ld(4 * NDG), HL = LDA / 2;
  // T = 4 only: at T = 5 (nx = 70) P+ and F alone are 40 tiles = 320 registers and the compiler spills 540 of them -- measured 6.7 ms
  // per 1024 against the tile-split kernel's 4.8; the code below is written for general T (tests/rw_lane_model.py checks the lane
  // algebra of both shapes); T = 5 runs the two-wave form, riccati_backward_rw2.hpp
  static constexpr bool SHAPE_OK = (TU == 2) && (NU > 16) && (NU < 32) && (NP > 0) && (G1 < G0) && (NX % 2 == 0) && (NUC > 0) && (NUC < 16);
  static constexpr bool OK = (T == 4) && SHAPE_OK;    // one wave per instance (this file)
  static constexpr bool OK2 = (T == 5) && SHAPE_OK;   // two waves per instance (riccati_backward_rw2.hpp)
  static constexpr int cg(int g) { return g < G1 ? g : g - G0 + G1; }   // compact index of the dense k group g
  static constexpr int pad8(int n) { return (n + 7) & ~7; }
  static constexpr int SCR_LD = 17, SCR_TILE = pad8(16 * SCR_LD);
};

template <int NV, int NU, int NS>
struct RwLds {
  using C = RwCfg<NV, NU>;
  static constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  static constexpr int VOFF_LX = SL.kkt.off[RTOC_KKT_LX] - SL.kkt.off[RTOC_KKT_FX], VOFF_LU = SL.kkt.off[RTOC_KKT_LU] - SL.kkt.off[RTOC_KKT_FX];
  static constexpr int CV = (VOFF_LU + NU + 1) / 2, PS = (CV + 63) / 64;   // 16-byte chunks / DMA pieces of the strip Fx | lx | lu
  static constexpr int OFF_A = 0;
  static constexpr int OFF_ST = C::NX * C::LDA;
  static constexpr int OFF_S = OFF_ST + C::pad8(2 * CV);         // s+ (the last strip piece is partial: the strip takes 2 CV doubles)
  // ---- the landing zone of the Qxx panels: everything from here to the grid table is dead between the policy products and the
  //      transposes at the stage end ----
  static constexpr int OFF_G = OFF_S + C::pad8(C::NX);           // Quu (by DMA) -> G, factorised in place (L)
  static constexpr int OFF_Y = OFF_G + C::pad8(NU * NU);
  static constexpr int OFF_LINV = OFF_Y + C::pad8(NU * NU);
  static constexpr int OFF_SCR = OFF_LINV + C::pad8(NU);         // two transpose tiles; the Cholesky's 256-double scratch
  static constexpr int OFF_Z = OFF_SCR + 2 * C::SCR_TILE;
  static constexpr int OFF_W0 = OFF_Z + C::pad8(C::NX);
  static constexpr int OFF_LUP = OFF_W0 + C::pad8(C::NX);
  static constexpr int OFF_T = OFF_LUP + C::pad8(NU);
  static constexpr int OFF_GRID = OFF_T + C::pad8(NU);           // grid-point kinds of the horizon (ints)
  static constexpr int LDQ = lds_ld(C::NX), PANEL = 16 * LDQ;    // one column tile of Qxx (16 columns, padded: conflict-free both ways)
  static constexpr int OFF_Q = OFF_G;
  static_assert(OFF_Q + 2 * PANEL <= OFF_GRID, "two Qxx panels in the dead zone");
  static constexpr int DOUBLES = OFF_GRID + RV_MAX_STAGES / 2;
  static constexpr int BYTES = DOUBLES * 8;
  static_assert(2 * C::SCR_TILE >= 256, "the blocked Cholesky's scratch");
};

template <int NV, int NU, int NS>
__global__ __launch_bounds__(64, 1) void riccati_backward_rw_kernel(BwdArgs a) {
  using C = RwCfg<NV, NU>;
  using M = RwLds<NV, NU, NS>;
  static_assert(C::OK, "shape outside the register-wide kernel's tiling");
  constexpr int NX = C::NX, NP_ = C::NP, T = C::T, TU = C::TU, KG = C::KG, KGU = C::KGU, G0 = C::G0, G1 = C::G1, NUC = C::NUC;
  constexpr int TS = C::TS, LS = C::LS, QS = C::QS, RS = C::RS, LDA = C::LDA, HL = C::HL, SCR_LD = C::SCR_LD, SCR_TILE = C::SCR_TILE;
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt, RL = SL.ric;
  static_assert(M::VOFF_LX > 0 && M::VOFF_LU > M::VOFF_LX && M::VOFF_LX % 2 == 0 && M::VOFF_LU % 2 == 0, "Fx, lx, lu lie behind one another in the record");
  static_assert(KL.off[RTOC_KKT_FXX] % 2 == 0 && KL.off[RTOC_KKT_FX] % 2 == 0 && KL.off[RTOC_KKT_QUU] % 2 == 0 && KL.off[RTOC_KKT_QXX] % 2 == 0 && KL.stride % 2 == 0 && NX % 2 == 0, "16-byte chunks");
  constexpr int ST_FX = M::OFF_ST, ST_LX = ST_FX + M::VOFF_LX, ST_LU = ST_FX + M::VOFF_LU;
  constexpr int NPC_A = (NX + 64 / HL - 1) / (64 / HL);   // DMA instructions of A per grid point
  static_assert(NPC_A < 64, "counted wait");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const sA = smem + M::OFF_A;
  double* const sG = smem + M::OFF_G;
  double* const sY = smem + M::OFF_Y;
  double* const scr = smem + M::OFF_SCR;
  double* const sS = smem + M::OFF_S;
  double* const sZ = smem + M::OFF_Z;
  double* const sW0 = smem + M::OFF_W0;
  double* const sLup = smem + M::OFF_LUP;
  double* const sT = smem + M::OFF_T;
  int* const sGrid = reinterpret_cast<int*>(smem + M::OFF_GRID);

  const int b = a.first + (int)blockIdx.x;
  if (b >= a.batch) return;
  const int lane0 = threadIdx.x;
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
  auto issue_dma_A = [&](int stage) __attribute__((always_inline)) {
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride + KL.off[RTOC_KKT_FXX];
    constexpr int CPP = 64 / HL, NPC = (NX + CPP - 1) / CPP;
    const unsigned col0 = (unsigned)lane / HL, ch0 = (unsigned)lane - col0 * HL;
    const unsigned rl = 2 * ch0;                                                        // compact row of the chunk
    const unsigned srow = (rl < 4 * G1) ? rl : ((rl < 4 * C::NDG) ? rl + 4 * (G0 - G1) : 0);   // (the padding chunk loads anything)
    const unsigned voff = col0 * NX + srow;
    if (col0 < CPP) {
#pragma unroll
      for (int p = 0; p < NPC; ++p) {
        if ((p + 1) * CPP <= NX || p * CPP + (int)col0 < NX)
          __builtin_amdgcn_global_load_lds(kp + p * CPP * NX + voff, (lds_ptr_t)(sA + p * CPP * LDA), 16, 0, 0);
      }
    }
  };
  auto issue_dma_strip = [&](int stage) __attribute__((always_inline)) {
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride + KL.off[RTOC_KKT_FX];
#pragma unroll
    for (int p = 0; p < M::PS; ++p) {
      const int n = lane + 64 * p;
      if (64 * (p + 1) <= M::CV || n < M::CV) __builtin_amdgcn_global_load_lds(kp + 2 * n, (lds_ptr_t)(smem + M::OFF_ST + 128 * p), 16, 0, 0);
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
    if (term) {
      double* rr = a.ric + rinst + (size_t)N * RL.stride;
      const d2* s2 = reinterpret_cast<const d2*>(psrc);
      d2* t2 = reinterpret_cast<d2*>(rr + RL.off[RTOC_RIC_P]);
      for (int e = lane; e < NX * NX / 2; e += 64) t2[e] = s2[e];
      for (int e = lane; e < NX; e += 64) rr[RL.off[RTOC_RIC_S] + e] = -ssrc[e];
    }
  }
  // ---- operands that come straight from HBM, requested one stage ahead: Bv fragments (B operand of PB, A operand of G), Quu in the C
  //      layout.  RAW: lanes beyond the matrices load a clamped address and are masked where the value is USED ----
  double bvr[KG - G0][TU];
  auto issue_bq = [&](int stage) __attribute__((always_inline)) {
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride;
    const double* bp = kp + KL.off[RTOC_KKT_FVU];
#pragma unroll
    for (int g = G0; g < KG; ++g)
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) {
        const int k = 4 * g + q - NV, u = 16 * tu + li;
        const bool ok = k >= 0 && k < NV && u < NU;
        bvr[g - G0][tu] = bp[ok ? k + u * NV : (g - G0) * TU + tu];
      }
    // Quu -> the LDS place of G by DMA (flat: G's leading dimension is NU); the G product adds itself onto it
    constexpr int CQ = (NU * NU + 1) / 2, PQ = (CQ + 63) / 64;
    const double* gp = kp + KL.off[RTOC_KKT_QUU];
#pragma unroll
    for (int p = 0; p < PQ; ++p) {
      const int n = lane + 64 * p;
      if (64 * (p + 1) <= CQ || n < CQ) __builtin_amdgcn_global_load_lds(gp + 2 * n, (lds_ptr_t)(smem + M::OFF_G + 128 * p), 16, 0, 0);
    }
  };
  // Qxx comes in by DMA too, one column tile (16 columns, all rows: 8 KB, contiguous in the record) at a time into one of two padded
  // panels in the zone that is dead between the policy products and the transposes; F accumulates from zero and takes its start
  // value panel by panel at the end of the column loop's iterations (the same sum in another order; the symmetrisation of
  // brrf.cpp:85 folded in): panel p holds the transposed elements of the tiles (p, t >= p) -- read with li along the contiguous
  // index -- and the direct elements of the tiles (c < p, p).  No registers wait for these 32 KB.
  constexpr int LDQ = M::LDQ;
  constexpr bool EARLY = (T <= 4);   // Qxu^T requested at the stage top (64 registers for ~12k cycles) or just ahead of the H product
  auto NQ_NEXT = [](int t) constexpr { return (16 * (t + 2) <= NX) ? 16 : NX - 16 * (t + 1); };   // DMA instructions of panel t + 1
  d4 f[T][T];
  auto issue_qxx_panel = [&](const double* kr_, int p) __attribute__((always_inline)) {
    const double* qb_ = kr_ + KL.off[RTOC_KKT_QXX] + (size_t)16 * p * NX;
    double* dst = smem + M::OFF_Q + (p & 1) * M::PANEL;
#pragma unroll
    for (int col = 0; col < 16; ++col) {
      if (16 * p + col >= NX) continue;
      if (lane < NX / 2) __builtin_amdgcn_global_load_lds(qb_ + col * NX + 2 * lane, (lds_ptr_t)(dst + col * LDQ), 16, 0, 0);
    }
  };
  auto qxx_seed_panel = [&](int p) __attribute__((always_inline)) {
    const double* pan = smem + M::OFF_Q + (p & 1) * M::PANEL;
#pragma unroll
    for (int t = p; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // tile (p, t): element (16p + 4r + q, 16t + li) <- Qxx[16t + li][16p + 4r + q], column 4r + q of the panel
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
        const int j = 16 * p + li;
        const bool ok = (16 * p + 15 < NX) || j < NX;
        const double v = pan[16 * c + 4 * r + q + (ok ? li : 0) * LDQ];
        f[c][p][r] = __builtin_fma(0.5, ok ? v : 0.0, f[c][p][r]);
      }
  };
  issue_dma_strip(hi);
  issue_bq(hi);
  issue_dma_A(hi);
  for (int e = lane; e < a.nstages; e += 64) sGrid[e] = a.grid[e].type | (a.grid[e].dims << 8);   // (the host keeps nstages <= RV_MAX_STAGES)
  rv_lds_sync();

  for (int st = hi; st >= lo; --st) {
    lane = lane0;
    asm volatile("" : "+v"(lane));   // opaque per stage: keeps LICM from pinning the per-lane addresses of every unrolled loop
    lane &= 63;
    li = lane & 15;
    q = lane >> 4;
    const int gword = __builtin_amdgcn_readfirstlane(sGrid[st]);
    const bool impact = (gword & 0xff) == RTOC_GRID_IMPACT;
    const double* kr = a.kkt + kinst + (size_t)st * KL.stride;
    double* rr = a.ric + rinst + (size_t)st * RL.stride;

    RV_PROF(0);
    // the strip of this grid point and its Bv / Quu registers have landed; the DMA pieces of A -- the youngest vector-memory
    // operations of the previous stage, NPC_A instructions -- may still be in flight (A is first read by the H product)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC_A) : "memory");
    __builtin_amdgcn_wave_barrier();
    RV_PROF(1);
    // ================= 1. z = s+ - P+ Fx (brrf.cpp:86): per-lane partial sums over the rows a lane holds, q-reduction =========
    {
      double fxr[KG];
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const double v = smem[ST_FX + 4 * g + q];
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
    rv_lds_sync();
    auto zrow = [&](int g) __attribute__((always_inline)) -> double {   // z in the row layout: z[4g + q] (re-read where it rides: no registers held)
      const double v = sZ[4 * g + q];
      return (4 * g + 3 < NX || 4 * g + q < NX) ? v : 0.0;
    };

    RV_PROF(2);
    // Qxu^T in the layout of H^T (row u = 16 tu + 4r + q, column x = 16c + li), RAW
    d4 hq[TU][T];
    auto issue_hq = [&]() __attribute__((always_inline)) {
      const double* hp = kr + KL.off[RTOC_KKT_QXU];
#pragma unroll
      for (int tu = 0; tu < TU; ++tu)
#pragma unroll
        for (int c = 0; c < T; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = 16 * tu + 4 * r + q, x = 16 * c + li;
            hq[tu][c][r] = hp[((x < NX) ? x : NX - 1) + ((u < NU) ? u : NU - 1) * NX];
          }
    };
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
          bvf[g - G0][tu] = (k >= 0 && k < NV && u < NU) ? bvr[g - G0][tu] : 0.0;
        }
#pragma unroll
      for (int g = G0; g < KG; ++g)
#pragma unroll
        for (int tu = 0; tu < TU; ++tu)
#pragma unroll
          for (int c = 0; c < T; ++c) acc[c][tu] = mfma16(pp[g / 4][c][g % 4], bvf[g - G0][tu], acc[c][tu]);
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
            if (tu == TU - 1 && u0 < NU && li == NUC) sLup[u0] = smem[ST_LU + u0] - gacc[tr][tu][r];   // lu' = lu - Bv^T z_v
          }
      rv_lds_sync();
    }
    RV_PROF(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // A has landed
    __builtin_amdgcn_wave_barrier();
    // z into the idle column NU of PB: column NU of H = A^T PB is then A^T z (on an impact grid point PB is that column alone)
#pragma unroll
    for (int g = 0; g < KG; ++g) acc[g / 4][TU - 1][g % 4] = (li == NUC) ? zrow(g) : acc[g / 4][TU - 1][g % 4];

    // structured rows of A^T [.] for one column of C tiles (rows = state rows): dst[k] += ca src[k], dst[NV + k] += cc src[k],
    // k in [NP, NV); NV + k lies TS tiles, RS registers and QS q-groups below k (tests/rw_lane_model.py: struct_rows_add)
    const double ca = sA[NP_ + NP_ * LDA], cc = sA[NP_ + (NV + NP_) * LDA];   // A[NP][NP], A[NP][NV + NP] (row NP is a corner-group row: staged)
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

    d4 zt[TU][T];   // Z^T = Y H^T
    if (!impact) {
      issue_hq();
      // ================= 4. H^T = Qxu^T + PB^T A DIRECTLY in the layout Z^T = Y H^T reads (rows u, columns x): PB's C tiles are the
      //                      A operand as they stand, A's fragments the B operand; the structured rows k of A (ca e_k | cc e_NV+k)
      //                      are synthesised fragments that meet two or three column tiles.  Row NUC of control tile TU - 1 is the
      //                      rider (A^T z)^T.  No transposes, no rotations (tests/rw_lane_model.py: direct_ht) =================
      d4 hT[TU][T];
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) {
        asm volatile("" ::: "memory");   // (A's fragments are re-read per control tile rather than held across them)
#pragma unroll
        for (int c = 0; c < T; ++c) hT[tu][c] = zero4();
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
        if (tu == TU - 1) {   // the rider: w0 = A^T z
#pragma unroll
          for (int c = 0; c < T; ++c) {
            const int x = 16 * c + li;
            if (q == NUC % 4 && x < NX) sW0[x] = hT[tu][c][NUC / 4];
          }
        }
#pragma unroll
        for (int c = 0; c < T; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = 16 * tu + 4 * r + q, x = 16 * c + li;
            hT[tu][c][r] = (u < NU && x < NX) ? hT[tu][c][r] + hq[tu][c][r] : 0.0;
          }
      }
      RV_PROF(4);
      // ================= 3. LLT(G) (riccati_factorizer.cpp:49), Y = L^-1; t = Y lu', k = -Y^T t =================
      if (wave_llt_inv_blocked<NU, NU>(sG, sG, smem + M::OFF_LINV, sY, scr, lane)) stat |= RTOC_STAT_QUU_NOT_SPD;
      rv_lds_sync();
      RV_PROF(5);
      {
        const int u = (lane < NU) ? lane : 0;
        double tv = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) tv = __builtin_fma(sY[u + j * NU], sLup[j], tv);
        if (lane < NU) sT[lane] = tv;
        rv_lds_sync();
        double kv = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) kv = __builtin_fma(sY[j + u * NU], sT[j], kv);
        if (lane < NU) rr[RL.off[RTOC_RIC_KV] + lane] = -kv;
        if (is_bad(kv)) stat |= RTOC_STAT_NAN;
      }
      RV_PROF(6);
      // ================= 6. Z^T = Y H^T (Y lower triangular: the tile above the diagonal is skipped) =================
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) {
#pragma unroll
        for (int c = 0; c < T; ++c) zt[tu][c] = zero4();
#pragma unroll
        for (int gj = 0; gj < KGU; ++gj) {
          if (gj >= 4 * (tu + 1)) continue;
          const int m = 16 * tu + li, k = 4 * gj + q;
          const bool ok = m < NU && k < NU;
          const double v = sY[(ok ? m : 0) + (ok ? k : 0) * NU];
          const double av = ok ? v : 0.0;
#pragma unroll
          for (int c = 0; c < T; ++c) zt[tu][c] = mfma16(av, hT[gj / 4][c][gj % 4], zt[tu][c]);
        }
      }
      RV_PROF(7);
      // ================= 7. K = -Y^T Z^T (riccati_factorizer.cpp:55), tile by tile -> HBM (K row-major) =================
      double chk = 0.0;
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) {
        double yv[KGU];
#pragma unroll
        for (int gj = 0; gj < KGU; ++gj) {
          const int m = 16 * tu + li, k = 4 * gj + q;
          const bool ok = m < NU && k < NU;
          const double v = sY[(ok ? k : 0) + (ok ? m : 0) * NU];
          yv[gj] = ok ? -v : 0.0;
        }
#pragma unroll
        for (int c = 0; c < T; ++c) {
          d4 kk = zero4();
#pragma unroll
          for (int gj = 0; gj < KGU; ++gj) {
            if (gj < 4 * tu) continue;
            kk = mfma16(yv[gj], zt[gj / 4][c][gj % 4], kk);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = 16 * tu + 4 * r + q, x = 16 * c + li;
            if (u < NU && x < NX) rr[RL.off[RTOC_RIC_K] + u * NX + x] = kk[r];
            chk = __builtin_fma(kk[r], 0.0, chk);
          }
        }
      }
      if (is_bad(chk)) stat |= RTOC_STAT_NAN;
    } else {
      // impact grid point (riccati_factorizer.cpp:178-197): no controls; A^T z by the rider column alone
      d4 hx[T];   // row NUC: (A^T z)^T
#pragma unroll
      for (int c = 0; c < T; ++c) hx[c] = zero4();
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const double aop = acc[g / 4][TU - 1][g % 4];
#pragma unroll
        for (int c = 0; c < T; ++c) {
          const bool hit = struct_hit(g, c);
          if (!group_dense(g) && !hit) continue;
          double bop = group_dense(g) ? afrag(g, c) : 0.0;
          if (hit) bop += struct_frag(g, c);
          hx[c] = mfma16(aop, bop, hx[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < T; ++c) {
        const int x = 16 * c + li;
        if (q == NUC % 4 && x < NX) sW0[x] = hx[c][NUC / 4];
      }
#pragma unroll
      for (int tu = 0; tu < TU; ++tu)
#pragma unroll
        for (int c = 0; c < T; ++c) zt[tu][c] = zero4();
    }

    RV_PROF(8);
    // ================= 8. F starts from Qxx (upper tiles; off-diagonal ones symmetrised: brrf.cpp:85 folded into the start
    //                      value), F -= Z Z^T (brrf.cpp:82-84) =================
    // F accumulates from ZERO (-Z Z^T, then A^T W column by column); G, Y and the scratch are dead: the first two panels of Qxx
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    issue_qxx_panel(kr, 0);   // (panel 1 covers A^T z and t, which the update of s below still reads: behind it)
#pragma unroll
    for (int c = 0; c < T; ++c)
#pragma unroll
      for (int t = c; t < T; ++t) f[c][t] = zero4();
    if (!impact) {
#pragma unroll
      for (int gu = 0; gu < KGU; ++gu)
#pragma unroll
        for (int c = 0; c < T; ++c)
#pragma unroll
          for (int t = c; t < T; ++t) f[c][t] = mfma16(-zt[gu / 4][c][gu % 4], zt[gu / 4][t][gu % 4], f[c][t]);
    }
    RV_PROF(9);
    // ---- s = A^T z - lx - H k = w0 - lx + Z t (brrf.cpp:86-90): column layout, per-lane partial sums + q-reduction; -> LDS (the s+
    //      of the next grid point) and HBM, with the (zero) switching-time fields of the record ----
    rv_lds_sync();
    {
      double tr_[KGU];
#pragma unroll
      for (int gu = 0; gu < KGU; ++gu) {
        const double v = sT[4 * gu + q];
        tr_[gu] = (!impact && 4 * gu + q < NU) ? v : 0.0;
      }
#pragma unroll
      for (int c = 0; c < T; ++c) {
        double part = 0.0;
#pragma unroll
        for (int gu = 0; gu < KGU; ++gu) part = __builtin_fma(zt[gu / 4][c][gu % 4], tr_[gu], part);
        part = qsum(part);
        const int j = 16 * c + li;
        const int jc = (j < NX) ? j : 0;
        const double sn = sW0[jc] - smem[ST_LX + jc] + part;
        if (q == 0 && j < NX) {
          sS[j] = sn;
          rr[RL.off[RTOC_RIC_S] + j] = sn;
        }
        if (q == 1 && j < NX) rr[RL.off[RTOC_RIC_PSI] + j] = 0.0;
        if (q == 2 && j < NX) rr[RL.off[RTOC_RIC_PHI] + j] = 0.0;
      }
      if (lane < 5) rr[RL.off[RTOC_RIC_SCAL] + lane] = 0.0;
    }
    // the strip (Fx, lx, lu) has been read for the last time: the one of the next grid point
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    issue_qxx_panel(kr, 1);
    if (st > lo) issue_dma_strip(st - 1);
    asm volatile("" ::: "memory");

    RV_PROF(10);
    // ================= 9. column tile by column tile: W[:, t] = P+ A[:, t], F[c][t] += A^T[c] W[:, t] =================
#pragma unroll
    for (int t = 0; t < T; ++t) {
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
      // ---- panel t of Qxx has landed: every vector-memory operation older than the DMA of panel t + 1 is waited for (counted
      //      conservatively: only that younger panel's instructions may still be in flight); its part of the start value of F ----
      if (t + 1 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ_NEXT(t)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      qxx_seed_panel(t);
      // ---- P of grid point st + 1 -> HBM, a T-th per column tile, from the registers that hold it as P+ until the last of these
      //      products (element (i, j) through its mirror (j, i): li along the contiguous index) ----
      if (st < hi) {
        double* pw = a.ric + rinst + (size_t)(st + 1) * RL.stride + RL.off[RTOC_RIC_P];
#pragma unroll
        for (int mt = 0; mt < T; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * t + 4 * r + q, j = 16 * mt + li;
            if (i < NX && j < NX) pw[j + i * NX] = pp[t][mt][r];
          }
      }
      if (t + 2 < T) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the panel's reads are done: its buffer takes panel t + 2
        __builtin_amdgcn_wave_barrier();
        issue_qxx_panel(kr, t + 2);
      }
    }
    RV_PROF(11);
    // ---- A has been read for the last time: the dense rows of the next grid point's ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (st > lo) {
      issue_bq(st - 1);
      issue_dma_A(st - 1);   // (the LAST vector-memory operations of the stage: the counted wait at the next stage top)
    }
    asm volatile("" ::: "memory");

    RV_PROF(12);
    // ================= 10. P <- sym(F): upper tiles as computed, diagonal tiles mirrored, lower tiles transposed =================
    {
      auto masked = [&](int c, int t, int r, double v) -> double {
        const int i = 16 * c + 4 * r + q, j = 16 * t + li;
        if (16 * c + 4 * r >= NX) return 0.0;
        const bool rowok = (16 * c + 4 * r + 3 < NX) || i < NX;
        const bool colok = (16 * t + 15 < NX) || j < NX;
        return (rowok && colok) ? v : 0.0;
      };
      constexpr int NTR = T * (T + 1) / 2;
#pragma unroll
      for (int n0 = 0; n0 < NTR; n0 += 2) {
#pragma unroll
        for (int n = n0; n < n0 + 2 && n < NTR; ++n) {
          const int c = rv_tile_row(T, n), t = rv_tile_col(T, n);
          double* s_ = scr + (n - n0) * SCR_TILE;
#pragma unroll
          for (int r = 0; r < 4; ++r) s_[(q + 4 * r) * SCR_LD + li] = masked(c, t, r, f[c][t][r]);
        }
        rv_lds_sync();
#pragma unroll
        for (int n = n0; n < n0 + 2 && n < NTR; ++n) {
          const int c = rv_tile_row(T, n), t = rv_tile_col(T, n);
          const double* s_ = scr + (n - n0) * SCR_TILE;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double tr = s_[li * SCR_LD + q + 4 * r];
            if (t > c) {
              pp[c][t][r] = masked(c, t, r, f[c][t][r]);
              pp[t][c][r] = tr;                                  // zero outside the matrix: the tile was masked on its way in
            } else {
              pp[c][c][r] = (q + 4 * r <= li) ? masked(c, c, r, f[c][c][r]) : tr;   // upper triangle mirrored (P exactly symmetric)
            }
          }
        }
        rv_lds_sync();
      }
    }
    RV_PROF(13);
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
          if (i < NX && j < NX) pw[j + i * NX] = pp[kt][mt][r];
        }
  }
  if (stat) atomicOr(&a.status[b], stat);
}

}  // namespace rtoc

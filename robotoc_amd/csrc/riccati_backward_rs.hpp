// riccati_backward_rs.hpp -- role-split variant of the batched backward Riccati recursion.
//
// Same algebra, LDS carve and HBM record contract as riccati_backward.hpp (which documents the
// reference lines each step follows), different wave mapping: TWO wavefronts per OCP instance,
//   wave 0 ("matrix wave") issues every f64 MFMA product of the stage,
//   wave 1 ("vector wave") runs the dependent VALU chains -- z = s+ - P+ Fx, the in-wave
//          Cholesky of G, the triangular solves for K and k, w = A^T z, the policy copy-out --
// with block barriers only where data changes hands.  The instance still needs ~39 KB of LDS,
// so a CU holds 4 instances = 8 waves = 2 per SIMD: the matrix wave of one instance shares a
// SIMD with the vector wave of another and the MFMA and VALU pipes overlap in hardware, which
// a single in-order wave cannot do for itself (the serial Cholesky + solves cost about as many
// cycles as the two big MFMA products).
//
// Only for state dimensions with NX + 1 <= 64 (one vector wave covers a state vector): ANYmal,
// iiwa14.  Larger robots use the tile-split kernel of riccati_backward.hpp.
#pragma once
#include <type_traits>

#include "riccati_backward.hpp"

namespace rtoc {

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave complete in order; this only has to stop the compiler from
  // moving LDS accesses across the point and to drain the outstanding ones.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// One role's instruction stream.  The two roles run the SAME sequence of block barriers; keeping
// them in two separate loop nests (instead of if/else inside one loop) keeps the register
// live ranges of one role (MFMA accumulators / prefetch registers) out of the other role's code.
// One-way hand-offs matrix wave -> vector wave through an LDS word (monotonic sequence number).
// The matrix wave never consumes anything the vector wave produces before barrier B4, so the two
// mid-stage block barriers would only couple the two instruction streams; a flag lets the matrix
// wave run PB -> G -> PAa -> H -> F back to back while the vector wave trails it.  LDS is a single
// in-order unit per CU: once the flag store is visible, the wave's earlier LDS stores are too.
// The flag words are read and written with explicit DS instructions.  Through a `volatile int*` the compiler does not see the LDS
// address space and emits flat_load / flat_store with system coherence bits, each followed by `s_waitcnt vmcnt(0) lgkmcnt(0)`:
// every poll of a flag then DRAINS the wave's outstanding HBM traffic -- the record prefetch of the vector wave, the Qxx loads
// into the F accumulators of the matrix wave, the policy / P stores -- several times per stage.  (A generic pointer into LDS is
// the aperture base in the high word and the LDS byte offset in the low one.)
__device__ __forceinline__ int lds_flag_read(const volatile int* p) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(unsigned long long)p) : "memory");
  return v;
}
__device__ __forceinline__ void lds_flag_write(volatile int* p, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(unsigned long long)p), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_signal(volatile int* flag, int seq, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) lds_flag_write(flag, seq);
}
__device__ __forceinline__ void lds_wait(volatile int* flag, int seq) {
  while (__builtin_amdgcn_readfirstlane(lds_flag_read(flag)) < seq) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

// Barrier between the two waves of one instance.  NI == 1: the workgroup IS the instance, so the
// hardware barrier does it.  NI > 1 (several instances per workgroup): an LDS arrival counter --
// both waves run the same barrier sequence, so a monotonic count compared with 2 x (barriers
// passed) needs no reset.  Only LDS traffic is drained (like __syncthreads()), prefetch loads from
// HBM stay in flight.
template <int NI>
__device__ __forceinline__ void rs_sync(volatile int* cnt, int& epoch, int lane) {
  if constexpr (NI == 1) {
    __syncthreads();
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    epoch += 2;
    if (lane == 0)
      __hip_atomic_fetch_add(const_cast<int*>(cnt), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__builtin_amdgcn_readfirstlane(lds_flag_read(cnt)) < epoch) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  }
}
#define RTOC_BLOCK_SYNC() rs_sync<NI>(sFlag + 1, epoch, lane)

// SA ("structured A"): the caller guarantees (rtoc_check_fxx_structure) that the top half of every Fxx has the shape
// linearizeStateEquation / correctLinearizeStateEquation leave (src/dynamics/state_equation.cpp:52-55,80-82):
//   rows [NP, NV):  a e_i^T | c e_i^T      (Fqq = a I, Fqv = c I: a = 1, c = dt; c = 0 on impact grids)
//   rows [0, NP):   dense only inside the two NP x NP corners (the SE3 blocks of a floating base)
// so that rows [NP, NV) of A -- S = A - A~ -- contribute P+ S = [a P+[:, NP:NV] | c P+[:, NP:NV]] (scaled COPIES of
// columns of P+, not products) and S^T W = rows of W scaled; only the k-steps over the rows R = [0, NP) u [NV, NX)
// of A stay MFMA work: 7 of 9 k-steps in P+ A (column tiles without the Fx / fx riders) and in A^T (P+ A).
template <int NV, int NU, int NS, bool MW, int NI, bool SA = false>
__device__ __forceinline__ void riccati_backward_rs_body(const BwdArgs& a, const int slot) {
  static_assert(!SA || (NV % 4 == 2 && NV - NU >= 0), "structured-A path: row shift NV must be 2 mod 4 (quadrupeds, nv = 18)");
  constexpr int NP_ = NV - NU;  // passive (floating-base) coordinates: corner size of the state-equation blocks
  using C = BwdCfg<NV, NU, NS, 2>;
  constexpr int NX = C::NX, NT = 128, LDP = C::LDP, TNX = C::TNX, TMA = C::TMA, TNU = C::TNU;
  constexpr int CNT = TNX;  // the matrix wave owns every 16-tile
  static_assert(NX + 1 <= 64, "role-split kernel needs one vector wave per state vector");
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt, RL = SL.ric;
  extern __shared__ __attribute__((aligned(16))) double smem_all[];
  double* const smem = smem_all + slot * C::LDS_DOUBLES;
  double* const sP = smem + C::OFF_P;
  double* const sA = smem + C::OFF_A;
  double* const sPB = smem + C::OFF_PB;
  double* const sH = smem + C::OFF_H;
  double* const sKt = smem + C::OFF_KT;
  double* const sGK = smem + C::OFF_GK;
  double* const sBv = smem + C::OFF_BV;
  double* const sG = smem + C::OFF_G;
  double* const sL = smem + C::OFF_L;
  volatile int* const sFlag = reinterpret_cast<volatile int*>(smem + C::V_FLAG);

  const int tid0 = (MW ? 0 : 64) + (threadIdx.x & 63);  // thread index within the instance
  const int b = a.first + blockIdx.x * NI + slot;
  int epoch = 0;
  int tid = tid0, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  const int N = a.nstages - 1;
  const size_t kinst = (size_t)b * a.nstages * KL.stride;
  const size_t rinst = (size_t)b * a.nstages * RL.stride;
  unsigned stat = 0;

  // ---- terminal stage: P_N = Qxx_N, s_N = -lx_N (riccati_recursion.cpp:37-38) ----
  {
    const double* kr = a.kkt + kinst + (size_t)N * KL.stride;
    double* rr = a.ric + rinst + (size_t)N * RL.stride;
    copy_g2s_mat<NT, NX, NX, LDP>(sP, kr + KL.off[RTOC_KKT_QXX], tid);
    if (tid < NX) {
      const double v = -kr[KL.off[RTOC_KKT_LX] + tid];
      smem[C::V_SN + tid] = v;
      smem[C::V_PSIN + tid] = 0.0;
      smem[C::V_PHIN + tid] = 0.0;
      rr[RL.off[RTOC_RIC_S] + tid] = v;
    }
    if (tid < 8) smem[C::V_SCN + tid] = 0.0;
    RTOC_BLOCK_SYNC();
    copy_s2g_mat<NT, NX, NX, LDP>(rr + RL.off[RTOC_RIC_P], sP, tid);
  }

  // prefetch registers (next stage's record, loaded one stage ahead): held by the vector wave only
  constexpr int N2B = (NV * NU + 1) / 2, N2G = (NU * NU + 1) / 2;
  PreBuf<MatMap<64, NX>::passes(NX)> preA;
  PreBuf<MatMap<64, NX>::passes(NU)> preH;
  PreBuf<PreCnt<64, N2B>::value> preB;
  PreBuf<PreCnt<64, N2G>::value> preG;
  double preFx = 0.0, preLx = 0.0, preLu = 0.0;
  auto issue_loads = [&](int stage) {
    if constexpr (!MW) {
      const int v_ = tid - 64;
      const double* kp = a.kkt + kinst + (size_t)stage * KL.stride;
      // every load on every grid point, unconditionally (riccati_backward.hpp: issue_loads): no branch between the loads
      pre_load_mat<64, NX, NX>(preA, kp + KL.off[RTOC_KKT_FXX], v_);
      pre_load_mat<64, NX, NU>(preH, kp + KL.off[RTOC_KKT_QXU], v_);
      pre_load<64, N2B>(preB, kp + KL.off[RTOC_KKT_FVU], v_);
      pre_load<64, N2G>(preG, kp + KL.off[RTOC_KKT_QUU], v_);
      preFx = kp[KL.off[RTOC_KKT_FX] + (v_ < NX ? v_ : 0)];
      preLx = kp[KL.off[RTOC_KKT_LX] + (v_ < NX ? v_ : 0)];
      preLu = kp[KL.off[RTOC_KKT_LU] + (v_ < NU ? v_ : 0)];
    }
  };
  if (N >= 1) issue_loads(N - 1);

  // the vector wave's dependent chains (Cholesky, policy) are the critical path of a stage and can
  // only issue in the gaps of the matrix wave's MFMA stream on the shared SIMD: take those gaps first
  if constexpr (!MW) __builtin_amdgcn_s_setprio(3);
  // grid descriptors one stage ahead: a scalar load at the stage top would put an HBM/L2 round trip
  // on the critical path of every stage
  // (a scalar load shares its wait counter with the LDS traffic: the first LDS wait after it would
  // absorb the whole latency; a vector load of the eight ints of the descriptor -- lane l holds int l --
  // has its own counter and is read back with v_readlane at the next stage top).
  static_assert(offsetof(rtoc_grid, type) == 0 && offsetof(rtoc_grid, sto) == 4 && offsetof(rtoc_grid, sto_next) == 8 &&
                    offsetof(rtoc_grid, dims) == 20 && sizeof(rtoc_grid) >= 32,
                "lane <-> field map of the grid prefetch");
  int gv_ahead = reinterpret_cast<const int*>(a.grid + (N >= 1 ? N - 1 : 0))[tid0 & 7];
  int type_behind = a.grid[N].type;
  for (int st = N - 1; st >= 0; --st) {
    // opaque per-stage thread index: see riccati_backward.hpp (keeps LICM from pinning VGPRs)
    tid = tid0;
    asm volatile("" : "+v"(tid));
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    li = lane & 15;
    q = lane >> 4;
    constexpr bool mw = MW;       // matrix wave
    const int vt = tid - 64;      // vector-wave thread index (negative on the matrix wave)
    rtoc_grid g, gn;  // the fields of grid[st], grid[st + 1] this kernel reads
    g.type = __builtin_amdgcn_readlane(gv_ahead, 0);
    g.sto = __builtin_amdgcn_readlane(gv_ahead, 1);
    g.sto_next = __builtin_amdgcn_readlane(gv_ahead, 2);
    g.dims = __builtin_amdgcn_readlane(gv_ahead, 5);
    gn.type = type_behind;
    type_behind = g.type;
    gv_ahead = reinterpret_cast<const int*>(a.grid + (st > 0 ? st - 1 : 0))[lane & 7];  // grid[st - 1]
    const bool impact = (g.type == RTOC_GRID_IMPACT);
    const bool next_lift = (gn.type == RTOC_GRID_LIFT);
    const int ns = impact ? 0 : g.dims;
    const bool sto = g.sto != 0, sto_next = g.sto_next != 0;
    const double* kr = a.kkt + kinst + (size_t)st * KL.stride;
    double* rr = a.ric + rinst + (size_t)st * RL.stride;

    RTOC_PROF(0);
    RTOC_PROFV(16);
#define RTOC_GRID_PREV_STO (__builtin_amdgcn_readlane(gv_ahead, 1) != 0)
#define RTOC_PT_TOP_SYNC() do { if (do_pt) RTOC_BLOCK_SYNC(); } while (0)  // the roll needs no barrier here
#include "riccati_pt_block.inc"
#undef RTOC_PT_TOP_SYNC
#undef RTOC_GRID_PREV_STO
    RTOC_PROF(1);
    RTOC_PROFV(17);
    // ---- stage data: prefetched registers -> LDS (vector wave) ----
    // Hand-off vector -> matrix by sequence flags instead of a barrier: Bv and Quu first (all the
    // matrix wave needs for PB and G), then A, Qxu and the vectors while those products run.
    if constexpr (!MW) {
      if (!impact) {
        pre_store_flat<64, N2B>(sBv, preB, vt);
        pre_store_flat<64, N2G>(sG, preG, vt);
      }
      lds_signal(sFlag + 2, 3 * (N - st) - 2, lane);
      pre_store_mat<64, NX, NX, LDP>(sA, preA, vt);
      if (!impact) pre_store_mat<64, NX, NU, LDP>(sH, preH, vt);
      if (vt < NX) {
        smem[C::V_FX + vt] = preFx;
        smem[C::V_LX + vt] = preLx;
        if (sto) {
          smem[C::V_FFX + vt] = kr[KL.off[RTOC_KKT_FFX] + vt];
          smem[C::V_HX + vt] = kr[KL.off[RTOC_KKT_HX] + vt];
        }
      }
      if (!impact && vt < NU) {
        smem[C::V_LU + vt] = preLu;
        if (sto) smem[C::V_HU + vt] = kr[KL.off[RTOC_KKT_HU] + vt];
      }
    }
    if constexpr (!MW) {
      if (sto && vt < 8) smem[C::V_KSC + vt] = kr[KL.off[RTOC_KKT_SCAL] + vt];
      lds_signal(sFlag + 2, 3 * (N - st) - 1, lane);
#ifndef RTOC_RS_LATE_PREFETCH
      // next stage's record: HBM -> registers as soon as this stage's copy has left them for LDS, a whole stage ahead
      // of its use (issued at the end of the stage -- "after the register-hungry solve" of earlier versions -- its HBM
      // latency showed up as a ~2k-cycle wait of the matrix wave for Bv / Quu at the next stage top: 2.77 -> 2.67 ms;
      // costs two more spilled VGPRs, profiles/r02_kernel_resource_usage.txt)
      if (st > 0) issue_loads(st - 1);
#endif
    }
    if constexpr (MW) lds_wait(sFlag + 2, 3 * (N - st) - 2);

    RTOC_PROF(2);
    RTOC_PROFV(18);
    // Qxx of THIS stage straight from HBM into the accumulator registers of the F product, in the
    // MFMA C layout (row = q+4r, col = lane&15).  Issued two intervals before its first use, so the
    // latency hides behind PB / G / PAa; the accumulators are dead until then, so this prefetch
    // costs no extra registers and no LDS staging.
    d4 f[MW ? CNT : 1][MW ? TNX : 1];
    if constexpr (MW) {
      const double* qx_ = kr + KL.off[RTOC_KKT_QXX] + q + li * NX;   // Qxx[i][j]
      const double* qxt_ = kr + KL.off[RTOC_KKT_QXX] + li + q * NX;  // Qxx[j][i]
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int t = 0; t < TNX; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // Only the 16-tiles on and above the diagonal are computed: A^T P+ A and K^T G K are
            // symmetric, so tile (t,c) is the mirror of tile (c,t).  Off-diagonal tiles start from
            // the symmetrised Hessian block (Qxx[c][t] + Qxx[t][c]^T)/2, which makes the mirrored
            // result equal to the reference's P = (F + F^T)/2 up to the rounding of the products.
            if (t < c) continue;
            const int i = c * 16 + drow(q, r), j = t * 16 + li;
            const bool ok = (i < NX && j < NX);
            const double v = ok ? qx_[c * 16 + 4 * r + t * 16 * NX] : 0.0;
            if (t == c) {
              f[c][t][r] = v;
            } else {
              const double vt_ = ok ? qxt_[t * 16 + (c * 16 + 4 * r) * NX] : 0.0;
              f[c][t][r] = 0.5 * (v + vt_);
            }
          }
    }
    // ================= interval 1: [matrix] PB, G (+ Bv^T s+_v as a free-rider column)  || [vector] -- ========
    // lu' = lu - Bv^T z_v with z = s+ - P+ Fx: the P+ Fx part arrives from the matrix wave (it rides as an extra
    // column of the P+ A product), and so does Bv^T s+_v now -- as column NU of the G product, whose 16-tile has
    // 16 - NU spare columns (psi_u = hu + Bv^T y_v and phi_u = Bv^T Phi+_v of the STO stages, brrf.cpp:48-66, ride as
    // columns NU+1, NU+2).  The vector wave used to sum them here, 3k contended cycles in FRONT of the Cholesky,
    // which is the critical path of the stage from the moment G is ready.
    static_assert(NU + 3 <= 16, "free-rider columns of the G product");
    if constexpr (!MW) {
    } else {
     if (!impact) {
      // ---- PB = P+[:,v] Bv, and chained from its accumulators G = Quu + Bv^T PB[v,:] ----
      {
        static_assert(TNU == 1, "one control tile");
        d4 acc[CNT];
#pragma unroll
        for (int c = 0; c < CNT; ++c) acc[c] = zero4();
        // P+ is exactly symmetric: its A-operand fragments are read TRANSPOSED (k contiguous, row li strided by LDP) -- with
        // LDP = 2 mod 4 the sixteen rows of a half-wave land 12 words apart (an odd multiple of 4 words: all 64 banks once), whereas
        // li contiguous / k strided puts the two k-columns of a half-wave 12 words apart on top of each other (2-way conflicts)
        const double* pa_ = sP + (NV + q) + li * LDP;
        const double* pb_ = sBv + q + li * NV;
        // The C layout of PB (row i = q + 4r + 16c, column u = lane & 15) is the B-operand layout (k = q, n = u) of
        // the G product, so Bv^T PB[v,:] runs straight from the accumulators -- no LDS round trip between the two
        // products.  The rows of PB[v,:] start at NV, which need not be a multiple of 4: the k-steps walk the
        // aligned row groups i = 4 (NV/4 + g) + q and the A operand Bv^T[u][k = i - NV] is zero where k < 0.
        constexpr int G0 = NV / 4, KSG = (NX - 4 * G0 + 3) / 4;
        double ga[KSG];  // A operand of the G product: Bv[k][u = li], k = 4 (G0 + g) + q - NV (issued ahead of the PB stream)
#pragma unroll
        for (int g = 0; g < KSG; ++g) {
          const int k = 4 * (G0 + g) + q - NV;
          const bool ok = k >= 0 && k < NV && li < NU;
          const double v = sBv[(ok ? k : 0) + (li < NU ? li : 0) * NV];
          ga[g] = ok ? v : 0.0;
        }
        // rider columns of G (li == NU: s+_v, NU+1: Psi+_v (sto), NU+2: Phi+_v (sto && sto_next)): B operand from
        // the vectors, row i of the state
        const int rid = li - NU;
        const bool okr = rid == 0 || (sto && (rid == 1 || (rid == 2 && sto_next)));
        const double* pr_ = smem + (rid == 1 ? C::V_PSIN : (rid == 2 ? C::V_PHIN : C::V_SN)) + q;
        double gr[KSG];
#pragma unroll
        for (int g = 0; g < KSG; ++g) {
          const int i = 4 * (G0 + g) + q;
          const double v = pr_[4 * (G0 + g)];  // rows up to 4 ceil(NX / 4) - 1: inside the 8-padded vector slot
          gr[g] = (okr && i < NX) ? v : 0.0;
        }
#pragma unroll
        for (int ks = 0; ks < (NV + 3) / 4; ++ks) {
          const bool kok = (ks * 4 + 3 < NV) || (ks * 4 + q < NV);
          const double vb = pb_[ks * 4];
          const double bv = (kok && li < NU) ? vb : 0.0;
#pragma unroll
          for (int c = 0; c < CNT; ++c) {
            const double v = pa_[c * 16 * LDP + ks * 4];
            const double av = (kok && (c * 16 + li < NX)) ? v : 0.0;
            acc[c] = mfma16(av, bv, acc[c]);
          }
        }
        d4 gacc = zero4();
#pragma unroll
        for (int g = 0; g < KSG; ++g) {
          const int gg = G0 + g;
          const double pbv = acc[gg / 4][gg % 4];          // PB[4 gg + q][li] (zero in the columns li >= NU)
          gacc = mfma16(ga[g], rid >= 0 ? gr[g] : pbv, gacc);
        }
        // PB -> LDS (A-operand rows of the [P+; PB^T] A product) while the G chain drains
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = c * 16 + drow(q, r), u = li;
            if (i < NX && u < NU) sPB[i + u * LDP] = acc[c][r];
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u0 = drow(q, r), u1 = li;
          if (u0 < NU && u1 < NU) sG[u0 + u1 * NU] += gacc[r];
          if (u0 < NU && u1 >= NU && u1 < NU + 3)
            smem[(u1 == NU ? C::V_BTS : (u1 == NU + 1 ? C::V_BTPSI : C::V_BTPHI)) + u0] = gacc[r];
        }
      }
     } else {
      for (int e = lane; e < NU * LDP; e += 64) sPB[e] = 0.0;
     }
      lds_signal(sFlag, 3 * (N - st) - 2, lane);  // G ready
    }
    RTOC_PROFV(20);
#ifdef RTOC_RS_EARLY_PCOPY
    if constexpr (!MW) {
      // P of the previous stage (still intact in sP until the end of this stage) -> HBM, in the
      // shadow of the matrix wave's PB / G products
      if (st < N - 1)
        copy_s2g_mat<64, NX, NX, LDP>(a.ric + rinst + (size_t)(st + 1) * RL.stride + RL.off[RTOC_RIC_P], sP, vt);
    }
#endif
    if constexpr (!MW) lds_wait(sFlag, 3 * (N - st) - 2);
    RTOC_PROFV(21);

    RTOC_PROF(3);
    // ================= interval 2: [matrix] PAa, H     || [vector] LLT(G), w = A^T z =========
    d4 pa[MW ? TMA : 1][MW ? CNT : 1];
    static_assert(NU <= 16, "role-split kernel keeps L^-1 in one MFMA tile");
    if constexpr (MW) {
#pragma unroll
      for (int tm = 0; tm < TMA; ++tm)
#pragma unroll
        for (int c = 0; c < CNT; ++c) pa[tm][c] = zero4();
      const double* pb_ = sA + q + li * LDP;
      // row k = 4 ks + q of A takes part in the structured k-steps iff it is one of the dense rows R
      auto in_R = [&](int ks) { return 4 * ks + q < NP_ || 4 * ks + q >= NV; };
      if constexpr (SA) {
        // The two scale factors below are read from A in LDS: A must be THERE.  On an impact grid this wave gets here
        // straight from the Bv / Quu flag (no PB / G products in between) while the vector wave is still unpacking A --
        // and c differs between impact grids (0) and their neighbours (dt).  The stale read never showed with the
        // vector wave's loads drained in front of the flag (its vmcnt(0) at the stage top), and at 7 of 100k
        // instance-sweeps without (tools/determinism_probe.py).
        lds_wait(sFlag + 2, 3 * (N - st) - 1);  // A, Qxu, Fx in LDS
        // [P+; PB^T] S: column j of the result is a scaled COPY of column src(j) of [P+; PB^T]
        //   j in [NP, NV): a * col j        j in [NV + NP, NX): c * col (j - NV)
        // for the column tiles that are multiplied over R only (all but the last, which carries the Fx / fx riders
        // and stays dense).  a = A[NP][NP], c = A[NP][NV + NP] (checked uniform by rtoc_check_fxx_structure).
        const double ca = sA[NP_ + NP_ * LDP], cc = sA[NP_ + (NV + NP_) * LDP];
#pragma unroll
        for (int c = 0; c < CNT - 1; ++c) {
          const int j = c * 16 + li;
          const bool isa = j >= NP_ && j < NV, isc = j >= NV + NP_ && j < NX;
          const int src = isa ? j : (isc ? j - NV : 0);
          const double coef = isa ? ca : (isc ? cc : 0.0);
#pragma unroll
          for (int tm = 0; tm < TMA; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = tm * 16 + drow(q, r);
              double v;
              if (tm * 16 + 4 * r + 3 < NX)
                v = sP[i + src * LDP];
              else if (tm * 16 + 4 * r >= NX)
                v = (i < NX + NU) ? sPB[src + (i - NX) * LDP] : 0.0;
              else
                v = (i < NX) ? sP[i + src * LDP] : ((i < NX + NU) ? sPB[src + (i - NX) * LDP] : 0.0);
              pa[tm][c][r] = coef * v;
            }
        }
      }
      // last column tile: columns j < NX are A, column NX is Fx (not in STO stages, which keep the
      // VALU path for their extra vectors) -- P+ Fx and PB^T Fx come out of the same MFMAs
      static_assert(TNX * 16 >= NX + 2, "no spare columns for Fx / fx in the last tile");
      constexpr int JL = (CNT - 1) * 16;
      // column NX: Fx (-> P+ Fx, PB^T Fx); column NX+1 on STO stages: fx (-> P+ fx, PB^T fx)
      const int colsel = (JL + li == NX) ? 1 : ((JL + li == NX + 1 && sto) ? 2 : 0);
      const bool fxcol = colsel != 0;
      const double* pbl_ = (JL + li < NX) ? (sA + q + (JL + li) * LDP)
                                          : (smem + (JL + li == NX ? C::V_FX : C::V_FFX) + q);
      const bool okl = (JL + li < NX) || fxcol;
      // Row tiles in two passes: first the tiles that hold PB^T rows (H = Qxu^T-part, and PB^T Fx),
      // so the vector wave can start the policy products while the P+ A rows are still being
      // multiplied; then the pure P+ rows.
      constexpr int TMH = NX / 16;  // first row tile with a PB^T row
      constexpr int KSA = (NX + 3) / 4;
      auto run_pass = [&](auto tm0c, auto tm1c) {
        constexpr int TM0 = decltype(tm0c)::value, TM1 = decltype(tm1c)::value;
        if constexpr (TM1 > TM0) {
          // Operands of k-step ks, software-pipelined one step ahead in two halves: the LDS reads of step
          // ks+1 are ISSUED before the MFMAs of step ks (scheduling barrier in between -- left alone, the
          // scheduler sinks them behind the MFMAs and every step then waits out an LDS round trip), and
          // only masked / selected into operands after those MFMAs, when they have landed.
          struct Raw {
            double a[TMA], amix, b[CNT];
          };
          auto load_raw = [&](int ks, Raw& r) {
            r.amix = 0.0;
#pragma unroll
            for (int tm = TM0; tm < TM1; ++tm) {
              if (tm * 16 + 15 < NX) {
                r.a[tm] = sP[q + ks * 4 + (li + tm * 16) * LDP];   // P+[k][i] = P+[i][k], conflict-free (see the PB product)
              } else if (tm * 16 >= NX) {
                r.a[tm] = sPB[q + li * LDP + (tm * 16 - NX) * LDP + ks * 4];
              } else {  // the tile that straddles P+ rows and PB^T rows
                r.a[tm] = sP[q + ks * 4 + (li + tm * 16) * LDP];   // P+[k][i] = P+[i][k], conflict-free (see the PB product)
                r.amix = sPB[q + li * LDP + (tm * 16 - NX) * LDP + ks * 4];
              }
            }
#pragma unroll
            for (int c = 0; c < CNT; ++c) r.b[c] = (c < CNT - 1) ? pb_[ks * 4 + c * 16 * LDP] : pbl_[ks * 4];
          };
          auto finish_ops = [&](int ks, const Raw& r, double (&av)[TMA], double (&bv)[CNT]) {
            const bool kok = (ks * 4 + 3 < NX) || (ks * 4 + q < NX);
#pragma unroll
            for (int tm = TM0; tm < TM1; ++tm) {
              const int i = tm * 16 + li;
              double v = r.a[tm];
              if (tm * 16 + 15 >= NX && tm * 16 < NX) v = (i < NX) ? v : r.amix;
              if (tm * 16 + 15 >= NX && tm * 16 + 15 >= NX + NU) v = (i < NX + NU) ? v : 0.0;
              av[tm] = kok ? v : 0.0;
            }
            // B columns are NOT masked: an output column depends on its own B column only, and the columns
            // beyond A | Fx | fx are never read back (finite or not, whatever LDS holds there is harmless).
            // Every VALU instruction between two f64 MFMAs costs its full issue time (shared port).
#pragma unroll
            for (int c = 0; c < CNT; ++c) bv[c] = kok ? r.b[c] : 0.0;
            if constexpr (SA) {  // structured column tiles: rows of A outside R are covered by the S copies
              const bool keep = in_R(ks);
#pragma unroll
              for (int c = 0; c < CNT - 1; ++c) bv[c] = keep ? bv[c] : 0.0;
            }
          };
          Raw raw[2];
          load_raw(0, raw[0]);
#pragma unroll
          for (int ks = 0; ks < KSA; ++ks) {
            if (ks + 1 < KSA) load_raw(ks + 1, raw[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            double av[TMA], bv[CNT];
            finish_ops(ks, raw[ks & 1], av, bv);
            // k-steps whose four rows all lie in [NP, NV) touch the last (dense) column tile only
            constexpr int KS_LO = (NP_ + 3) / 4, KS_HI = NV / 4;  // k-steps [KS_LO, KS_HI) are entirely outside R
            const bool skip_struct = SA && ks >= KS_LO && ks < KS_HI;
#pragma unroll
            for (int tm = TM0; tm < TM1; ++tm)
#pragma unroll
              for (int c = 0; c < CNT; ++c)
                if (!(skip_struct && c < CNT - 1)) pa[tm][c] = mfma16(av[tm], bv[c], pa[tm][c]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      // column NX of [P+; PB^T] [A | Fx | fx]: rows < NX are P+ Fx -> z = s+ - P+ Fx (brrf.cpp:86),
      // rows NX.. are PB^T Fx = Bv^T (P+ Fx)_v -> the missing part of lu' (parked in the T slot).
      // column NX+1 (STO): y = P+ fx + Psi+ (brrf.cpp:52), PB^T fx -> missing part of psi_u (W slot).
      auto write_fx_column = [&](auto tm0c, auto tm1c) {
        constexpr int TM0 = decltype(tm0c)::value, TM1 = decltype(tm1c)::value;
        if (fxcol) {
          const double sgn = colsel == 1 ? -1.0 : 1.0;
          const double* base = smem + (colsel == 1 ? C::V_SN : C::V_PSIN);
          double* dstx = smem + (colsel == 1 ? C::V_Z : C::V_Y);
          double* dstu = smem + (colsel == 1 ? C::V_TV : C::V_WV);
#pragma unroll
          for (int tm = TM0; tm < TM1; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = tm * 16 + drow(q, r);
              const double v = pa[tm][CNT - 1][r];
              if (tm * 16 + 4 * r + 3 < NX) {
                dstx[row] = base[row] + sgn * v;
              } else if (tm * 16 + 4 * r >= NX) {
                if (row < NX + NU) dstu[row - NX] = v;
              } else {
                if (row < NX)
                  dstx[row] = base[row] + sgn * v;
                else if (row < NX + NU)
                  dstu[row - NX] = v;
              }
            }
        }
      };
      using std::integral_constant;
      lds_wait(sFlag + 2, 3 * (N - st) - 1);  // A, Qxu, Fx in LDS
#ifdef RTOC_RS_YIELD_CHOL
      // experiment: leave the SIMD's issue port to the vector wave's Cholesky (a dependent VALU chain that a back-to-
      // back f64 MFMA stream stretches 2-3x) before starting the P+ A product
      lds_wait(sFlag + 3, N - st);
#endif
      run_pass(integral_constant<int, TMH>{}, integral_constant<int, TMA>{});
      if (!impact) {
#pragma unroll
        for (int tm = TMH; tm < TMA; ++tm)
#pragma unroll
          for (int c = 0; c < CNT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = tm * 16 + drow(q, r);
              const int j = c * 16 + li;
              if (tm * 16 + 4 * r + 3 >= NX) {
                if (row >= NX && row < NX + NU && j < NX) sH[j + (row - NX) * LDP] += pa[tm][c][r];
              }
            }
      }
      write_fx_column(integral_constant<int, TMH>{}, integral_constant<int, TMA>{});
      lds_signal(sFlag, 3 * (N - st) - 1, lane);  // H and PB^T Fx ready, PB no longer read
      run_pass(integral_constant<int, 0>{}, integral_constant<int, TMH>{});
      write_fx_column(integral_constant<int, 0>{}, integral_constant<int, TMH>{});
      lds_signal(sFlag, 3 * (N - st), lane);  // z ready
    } else {
      if (!impact) {
        // LLT(G) (riccati_factorizer.cpp:49) and, in the same sweep, Y = L^-1 (column-major in the
        // dead Bv buffer): the triangular solves of the policy become two MFMA products below
        if (wave_llt_inv<NU, NU>(sG, sL, smem + C::V_LINV, sBv, NU, lane)) stat |= RTOC_STAT_QUU_NOT_SPD;
      }
#ifdef RTOC_RS_YIELD_CHOL
      lds_signal(sFlag + 3, N - st, lane);  // factorisation done: the matrix wave may start its dense MFMA stream
#endif
      RTOC_PROFV(22);
    }
    RTOC_PROFV(23);
    if constexpr (!MW) lds_wait(sFlag, 3 * (N - st) - 1);
    RTOC_PROFV(24);
    if constexpr (!MW) {
      if (!impact && vt < NU) {
        smem[C::V_LU + vt] += smem[C::V_TV + vt] - smem[C::V_BTS + vt];  // lu' = lu - Bv^T s+_v + PB^T Fx
        if (sto) {
          smem[C::V_PSIU + vt] = (smem[C::V_BTPSI + vt] + smem[C::V_HU + vt]) + smem[C::V_WV + vt];  // + PB^T fx
          smem[C::V_PHIU + vt] = sto_next ? smem[C::V_BTPHI + vt] : 0.0;
        }
      }
      // hand-off vector -> matrix: the inverse factor Y and lu' (psi_u, phi_u) are in LDS
      lds_signal(sFlag + 2, 3 * (N - st), lane);
    }

    RTOC_PROF(4);
    RTOC_PROF(5);
    // ================= interval 3: [matrix] F = Qxx + AtP A  || [vector] K, k solve ==========
    constexpr int TKT = (NX + 3 + 15) / 16;  // column tiles of the stacked right operand [H^T | lu' | psi_u | phi_u]
    d4 zt[MW ? TKT : 1];                     // Z^T = Y [H^T | ...] = L^-1 H^T: lives until F -= Z Z^T below
    if constexpr (MW) {
      const double* pbf_ = sA + q + li * LDP;
      // k runs over the register groups (tm, r) of PAa that hold P rows: g = 4*tm + r < KSF
      constexpr int KSF = (NX + 3) / 4;
      // same two-half software pipeline as the P+ A product: issue the reads of step g+1, barrier, mask and
      // multiply step g
      auto load_b = [&](int gidx, double (&bv)[TNX]) {
#pragma unroll
        for (int t = 0; t < TNX; ++t) bv[t] = pbf_[gidx * 4 + t * 16 * LDP];
      };
      double bvf[2][TNX];
      if constexpr (!SA) {
      load_b(0, bvf[0]);
#pragma unroll
      for (int gidx = 0; gidx < KSF; ++gidx) {
        if (gidx + 1 < KSF) load_b(gidx + 1, bvf[(gidx + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bool kok = (gidx * 4 + 3 < NX) || (gidx * 4 + q < NX);
        double bm[TNX];  // columns >= NX unmasked: they only reach entries of F that are never stored
#pragma unroll
        for (int t = 0; t < TNX; ++t) bm[t] = kok ? bvf[gidx & 1][t] : 0.0;
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
          const double avv = kok ? pa[gidx / 4][c][gidx % 4] : 0.0;
#pragma unroll
          for (int t = c; t < TNX; ++t) f[c][t] = mfma16(avv, bm[t], f[c][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      } else {
        // Structured form: F += A^T W with W = P+ A in the accumulators `pa` (their C layout is the B-operand
        // layout) and A^T as the A operand -- the same LDS words as above, A[k = 4g + q][16c + li], now read as
        // (m = column of A, k = row of A).  Rows k of A outside R are masked (k-steps wholly outside are skipped) and
        // their part S^T W is added as scaled rows of W:
        //   F[i][:] += a W[i][:] (i in [NP, NV)),   F[i][:] += c W[i - NV][:] (i in [NV + NP, NX)).
        constexpr int KS_LO = (NP_ + 3) / 4, KS_HI = NV / 4;
        constexpr int FIRST = (KS_LO == 0) ? KS_HI : 0;
        load_b(FIRST, bvf[0]);
        int par = 0;
#pragma unroll
        for (int gidx = 0; gidx < KSF; ++gidx) {
          if (gidx >= KS_LO && gidx < KS_HI) continue;  // all four rows in [NP, NV)
          int nxt = gidx + 1;
          if (nxt >= KS_LO && nxt < KS_HI) nxt = KS_HI;
          if (nxt < KSF) load_b(nxt, bvf[par ^ 1]);
          __builtin_amdgcn_sched_barrier(0);
          const bool kok = ((gidx * 4 + 3 < NX) || (gidx * 4 + q < NX)) && (4 * gidx + q < NP_ || 4 * gidx + q >= NV);
          double am[TNX];
#pragma unroll
          for (int t = 0; t < TNX; ++t) am[t] = kok ? bvf[par][t] : 0.0;
#pragma unroll
          for (int t = 0; t < TNX; ++t) {
            const double bvv = kok ? pa[gidx / 4][t][gidx % 4] : 0.0;
#pragma unroll
            for (int c = 0; c <= t; ++c) f[c][t] = mfma16(am[c], bvv, f[c][t]);
          }
          __builtin_amdgcn_sched_barrier(0);
          par ^= 1;
        }
        const double ca = sA[NP_ + NP_ * LDP], cc = sA[NP_ + (NV + NP_) * LDP];
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // a-part: rows i = 16c + 4r + q in [NP, NV), same lane and register of W
            if (16 * c + 4 * r + 3 >= NP_ && 16 * c + 4 * r < NV) {
              const int i = 16 * c + 4 * r + q;
              const double coef = (i >= NP_ && i < NV) ? ca : 0.0;
#pragma unroll
              for (int t = c; t < TNX; ++t) f[c][t][r] += coef * pa[c][t][r];
            }
            // c-part: rows i in [NV + NP, NX) take row i - NV of W: NV = 2 (mod 4), so that row sits two q-groups
            // away (lane ^ 32) in register group r (q >= 2) or r - 1 (q < 2)
            if (16 * c + 4 * r + 3 >= NV + NP_ && 16 * c + 4 * r < NX) {
              const int i = 16 * c + 4 * r + q;
              const double coef = (i >= NV + NP_ && i < NX) ? cc : 0.0;
              constexpr int SH = (NV + 2) / 4;             // i - NV = 4 (4c + r - SH) + (q + 2)  for q < 2
              const int g_hi = 4 * c + r - (NV - 2) / 4;   //       = 4 (4c + r - (NV-2)/4) + (q - 2)  for q >= 2
              const int g_lo = 4 * c + r - SH;
#pragma unroll
              for (int t = c; t < TNX; ++t) {
                // sender lane holds q_s = q ^ 2: q_s < 2 serves a destination with q >= 2 (group g_hi), else g_lo
                const double hi_v = (g_hi >= 0) ? pa[(g_hi >= 0 ? g_hi : 0) / 4][t][(g_hi >= 0 ? g_hi : 0) % 4] : 0.0;
                const double lo_v = (g_lo >= 0) ? pa[(g_lo >= 0 ? g_lo : 0) / 4][t][(g_lo >= 0 ? g_lo : 0) % 4] : 0.0;
                const double send = (q < 2) ? hi_v : lo_v;
                const double got = __shfl_xor(send, 32, 64);
                f[c][t][r] += coef * got;
              }
            }
          }
      }
      // the policy products run here, on the wave that owns the MFMA stream of this SIMD: issued
      // from the vector wave they queued behind the F chain above anyway
      lds_wait(sFlag + 2, 3 * (N - st));
      if (!impact && ns == 0) {
        // K = -G^-1 H^T, k = -G^-1 lu', T = -G^-1 psi_u, W = -G^-1 phi_u
        // (riccati_factorizer.cpp:55-56, :125-130) for all right-hand sides at once, as the two
        // triangular solves written as products with Y = L^-1:
        //   Z^T = Y [H^T | lu' | psi_u | phi_u],   [K | k | T | W] = -Y^T Z^T.
        // The three vectors ride as columns NX, NX+1, NX+2 behind the columns of H^T.  The C layout
        // of Z^T (row i = q+4r, column = lane&15) is the B-operand layout of the second product, so
        // Z^T never leaves the registers.  (G^-1 = Y^T Y is never formed: that would square the
        // conditioning of the factor.)
        {
          constexpr int KSU = (NU + 3) / 4;
          d4 kk[TKT];
          // per-lane source of column x = 16c + li in the tiles that are not pure H columns
          const double* psrc[TKT];
          int ssrc[TKT];
          bool oksrc[TKT];
#pragma unroll
          for (int c = 0; c < TKT; ++c) {
            zt[c] = zero4();
            kk[c] = zero4();
            const int x = c * 16 + li;
            const int voff = (x == NX ? C::V_LU : (x == NX + 1 ? C::V_PSIU : C::V_PHIU));
            psrc[c] = (x < NX) ? (sH + x + q * LDP) : (smem + voff + q);
            ssrc[c] = (x < NX) ? 4 * LDP : 4;
            oksrc[c] = (x < NX) || x == NX || (sto && (x == NX + 1 || (x == NX + 2 && sto_next)));
          }
          const double* ph_ = sH + li + q * LDP;   // B[k = u][n = x] = H[x][u]
          const double* py1_ = sBv + li + q * NU;  // A[m = i][k = u] = Y[i][u]
          const double* py2_ = sBv + q + li * NU;  // A[m = u][k = i] = Y[i][u]
          double y1[KSU], y2[KSU];
#pragma unroll
          for (int ks = 0; ks < KSU; ++ks) {
            const bool kok = (ks * 4 + 3 < NU) || (ks * 4 + q < NU);
            const double v1 = py1_[ks * 4 * NU], v2 = py2_[ks * 4];
            y1[ks] = (kok && li < NU) ? v1 : 0.0;
            y2[ks] = (kok && li < NU) ? -v2 : 0.0;
          }
#pragma unroll
          for (int ks = 0; ks < KSU; ++ks) {
            const bool kok = (ks * 4 + 3 < NU) || (ks * 4 + q < NU);
#pragma unroll
            for (int c = 0; c < TKT; ++c) {
              double v;
              bool ok;
              if (c * 16 + 15 < NX) {
                v = ph_[c * 16 + ks * 4 * LDP];
                ok = kok;
              } else {
                v = psrc[c][ks * ssrc[c]];
                ok = kok && oksrc[c];
              }
              zt[c] = mfma16(y1[ks], ok ? v : 0.0, zt[c]);
            }
          }
#pragma unroll
          for (int ks = 0; ks < KSU; ++ks)
#pragma unroll
            for (int c = 0; c < TKT; ++c) kk[c] = mfma16(y2[ks], zt[c][ks], kk[c]);
          // K[u = q+4r][x = 16c+li] -> K^T buffer (x + u*LDP); columns NX.. -> k, T, W
          double chk = 0.0;  // any NaN / Inf in the policy poisons this sum (0 * Inf = NaN)
#pragma unroll
          for (int c = 0; c < TKT; ++c) {
            double* dst;
            int dstride;
            if (c * 16 + 15 < NX) {
              dst = sKt + c * 16 + li + q * LDP;
              dstride = 4 * LDP;
            } else {
              const int x = c * 16 + li;
              double* d = smem + C::V_FLAG + 4;  // dummy slot for the spare lanes
              d = (x == NX + 2) ? (smem + C::V_WV + q) : d;
              d = (x == NX + 1) ? (smem + C::V_TV + q) : d;
              d = (x == NX) ? (smem + C::V_KV + q) : d;
              d = (x < NX) ? (sKt + x + q * LDP) : d;
              dst = d;
              dstride = (x < NX) ? 4 * LDP : (x < NX + 3 ? 4 : 0);
            }
#pragma unroll
            for (int r = 0; r < KSU; ++r) {
              if (r * 4 + 3 < NU || r * 4 + q < NU) dst[r * dstride] = kk[c][r];
              chk = __builtin_fma(kk[c][r], 0.0, chk);
            }
          }
          if (is_bad(chk)) stat |= RTOC_STAT_NAN;
        }
      }
    } else {
      // w = A^T z - lx (brrf.cpp:87-88), off the matrix wave's critical path: z came with the H flag
      lds_wait(sFlag, 3 * (N - st));
      if (vt < NX) {
        // A^T z (brrf.cpp:87-88) and, on STO stages, A^T y and A^T Phi+ (brrf.cpp:53,60) in one pass
        // over column vt of A: 128-bit LDS reads, two accumulators per product
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        static_assert((LDP & 1) == 0 && (C::OFF_A & 1) == 0 && (C::V_Z & 1) == 0 && (C::V_Y & 1) == 0 &&
                          (C::V_PHIN & 1) == 0 && (NX & 1) == 0,
                      "128-bit LDS reads");
        const dbl2* pa2 = reinterpret_cast<const dbl2*>(sA + vt * LDP);
        const dbl2* pz2 = reinterpret_cast<const dbl2*>(smem + C::V_Z);
        const dbl2* py2 = reinterpret_cast<const dbl2*>(smem + C::V_Y);
        const dbl2* pf2 = reinterpret_cast<const dbl2*>(smem + C::V_PHIN);
        double a0 = 0.0, a1 = 0.0, y0 = 0.0, y1 = 0.0, f0 = 0.0, f1 = 0.0;
        if (!sto) {
          double a2 = 0.0, a3 = 0.0;
#pragma unroll
          for (int k2 = 0; k2 + 1 < NX / 2; k2 += 2) {
            const dbl2 av0 = pa2[k2], zv0 = pz2[k2], av1 = pa2[k2 + 1], zv1 = pz2[k2 + 1];
            a0 += av0.x * zv0.x;
            a1 += av0.y * zv0.y;
            a2 += av1.x * zv1.x;
            a3 += av1.y * zv1.y;
          }
          if ((NX / 2) & 1) {
            const dbl2 av0 = pa2[NX / 2 - 1], zv0 = pz2[NX / 2 - 1];
            a0 += av0.x * zv0.x;
            a1 += av0.y * zv0.y;
          }
          a0 += a2;
          a1 += a3;
        } else {
          const double ysel = impact ? 0.0 : 1.0;
#pragma unroll
          for (int k2 = 0; k2 < NX / 2; ++k2) {
            const dbl2 av = pa2[k2], zv = pz2[k2], fv = pf2[k2], yv = py2[k2];
            a0 += av.x * zv.x;
            a1 += av.y * zv.y;
            f0 += av.x * fv.x;
            f1 += av.y * fv.y;
            y0 += av.x * (ysel * yv.x);
            y1 += av.y * (ysel * yv.y);
          }
        }
        smem[C::V_SNEW + vt] = (a0 + a1) - smem[C::V_LX + vt];
        if (sto) {
          if (!impact) {
            smem[C::V_PSIX + vt] = (y0 + y1) + smem[C::V_HX + vt];
            smem[C::V_PHIX + vt] = sto_next ? (f0 + f1) : 0.0;
          } else {
            smem[C::V_PHIX + vt] = f0 + f1;
          }
        }
      }
#ifndef RTOC_RS_EARLY_PCOPY
      // P of the previous stage (intact in sP until the matrix wave writes the new one after B4) -> HBM.  Here, behind
      // the Cholesky / inverse factor and w: between the G flag and the Y hand-off the vector wave IS the critical
      // path of the stage (phase stamps: the matrix wave finished the F product ~3k cycles before Y arrived), and this
      // copy used to sit in front of the factorisation.
      if (st < N - 1)
        copy_s2g_mat<64, NX, NX, LDP>(a.ric + rinst + (size_t)(st + 1) * RL.stride + RL.off[RTOC_RIC_P], sP, vt);
#endif
    }
    RTOC_PROF(13);
    RTOC_PROFV(14);
    RTOC_BLOCK_SYNC();  // B4: K, k ready; F product done; A and H(after SC) free

    RTOC_PROF(6);
    RTOC_PROFV(26);
    if (NS > 0 && ns > 0) {
#include "riccati_sc_block.inc"
      RTOC_BLOCK_SYNC();
    }
    RTOC_PROF(7);
    if (a.writeback && !impact) {
      double* kw = a.kkt_rw + kinst + (size_t)st * KL.stride;
      copy_s2g_mat<NT, NX, NU, LDP>(kw + KL.off[RTOC_KKT_QXU], sH, tid);
      copy_s2g_flat<NT>(kw + KL.off[RTOC_KKT_QUU], sG, NU * NU, tid);
      if (tid < NU) kw[KL.off[RTOC_KKT_LU] + tid] = smem[C::V_LU + tid];
      RTOC_BLOCK_SYNC();
    }

    // ================= interval 4: [matrix] GK, F -= K^T GK, P   || [vector] policy -> HBM ====
    if constexpr (MW) {
      if (!impact && ns == 0) {
        // K^T G K = H G^-1 H^T = Z Z^T with Z^T = L^-1 H^T still in the registers of the policy products:
        // its C layout is both the A-operand layout (m = state index, k = u) and the B-operand layout
        // (k = u, n = state index) of this product, so F -= K^T G K (brrf.cpp:82-84) needs no operand
        // loads and no G K intermediate.  Columns >= NX of Z^T (lu', psi_u, phi_u) only reach entries
        // of F outside NX x NX, which are never stored.
        static_assert(TKT >= TNX, "Z^T covers every column tile of F");
#pragma unroll
        for (int ks = 0; ks < (NU + 3) / 4; ++ks)
#pragma unroll
          for (int c = 0; c < CNT; ++c)
#pragma unroll
            for (int t = c; t < TNX; ++t) f[c][t] = mfma16(-zt[c][ks], zt[t][ks], f[c][t]);
      } else if (!impact) {
        {
          d4 acc[TNU][CNT];
#pragma unroll
          for (int t = 0; t < TNU; ++t)
#pragma unroll
            for (int c = 0; c < CNT; ++c) acc[t][c] = zero4();
          const double* pa_ = sG + li + q * NU;
          const double* pb_ = sKt + li + q * LDP;
#pragma unroll
          for (int ks = 0; ks < (NU + 3) / 4; ++ks) {
            const bool kok = (ks * 4 + 3 < NU) || (ks * 4 + q < NU);
            double av[TNU], bv[CNT];
#pragma unroll
            for (int t = 0; t < TNU; ++t) {
              const double v = pa_[t * 16 + ks * 4 * NU];
              av[t] = (kok && (t * 16 + li < NU)) ? v : 0.0;
            }
#pragma unroll
            for (int c = 0; c < CNT; ++c) {
              const double v = pb_[c * 16 + ks * 4 * LDP];
              bv[c] = (kok && (c * 16 + li < NX)) ? v : 0.0;
            }
#pragma unroll
            for (int t = 0; t < TNU; ++t)
#pragma unroll
              for (int c = 0; c < CNT; ++c) acc[t][c] = mfma16(av[t], bv[c], acc[t][c]);
          }
#pragma unroll
          for (int t = 0; t < TNU; ++t)
#pragma unroll
            for (int c = 0; c < CNT; ++c)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int u = t * 16 + drow(q, r), j = c * 16 + li;
                if (u < NU && j < NX) {
                  sGK[u + j * NU] = acc[t][c][r];
                }
              }
        }
        wave_lds_sync();
        RTOC_PROF(8);
        {
          const double* pa_ = sKt + li + q * LDP;
          const double* pb_ = sGK + q + li * NU;
#pragma unroll
          for (int ks = 0; ks < (NU + 3) / 4; ++ks) {
            const bool kok = (ks * 4 + 3 < NU) || (ks * 4 + q < NU);
            double av[CNT], bv[TNX];
#pragma unroll
            for (int c = 0; c < CNT; ++c) {
              const double v = pa_[c * 16 + ks * 4 * LDP];
              av[c] = (kok && (c * 16 + li < NX)) ? -v : 0.0;
            }
#pragma unroll
            for (int t = 0; t < TNX; ++t) {
              const double v = pb_[ks * 4 + t * 16 * NU];
              bv[t] = (kok && (t * 16 + li < NX)) ? v : 0.0;
            }
#pragma unroll
            for (int c = 0; c < CNT; ++c)
#pragma unroll
              for (int t = c; t < TNX; ++t) f[c][t] = mfma16(av[c], bv[t], f[c][t]);
          }
        }
        if (NS > 0 && ns > 0) {
          // switching constraint: the reference subtracts 2 K^T Phiu^T M from Qxx and lets
          // P = (F + F^T)/2 symmetrise it; with only the upper tiles held the symmetric form
          // K^T D + D^T K, D = Phiu^T M, is accumulated instead (rare stage, two more passes).
          wave_lds_sync();
#pragma unroll 1
          for (int e = lane; e < NU * NX; e += 64) {
            const int u = e % NU, j = e / NU;
            double dtm = 0.0;
            for (int l = 0; l < ns; ++l)
              dtm += smem[C::S_PHIU + l + u * C::NSP] * smem[C::S_M + l + j * C::NSP];
            sGK[u + j * NU] = dtm;
          }
          wave_lds_sync();
          const double* pk_ = sKt + li + q * LDP;
          const double* pd_ = sGK + q + li * NU;
#pragma unroll 1
          for (int ks = 0; ks < (NU + 3) / 4; ++ks) {
            const bool kok = (ks * 4 + 3 < NU) || (ks * 4 + q < NU);
            double ak[TNX], ad[TNX];
#pragma unroll
            for (int c = 0; c < TNX; ++c) {
              const bool ok = kok && (c * 16 + li < NX);
              const double vk = pk_[c * 16 + ks * 4 * LDP];
              const double vd = pd_[ks * 4 + c * 16 * NU];
              ak[c] = ok ? -vk : 0.0;
              ad[c] = ok ? vd : 0.0;
            }
#pragma unroll
            for (int c = 0; c < CNT; ++c)
#pragma unroll
              for (int t = c; t < TNX; ++t) {
                f[c][t] = mfma16(ak[c], ad[t], f[c][t]);   // -(K^T D)[i][j]
                f[c][t] = mfma16(ad[c], ak[t], f[c][t]);   // -(D^T K)[i][j]
              }
          }
        }
      }
      RTOC_PROF(9);
      if (a.writeback) {
        double* kw = a.kkt_rw + kinst + (size_t)st * KL.stride;
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int t = c; t < TNX; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = c * 16 + drow(q, r), j = t * 16 + li;
              if (i < NX && j < NX) {
                kw[KL.off[RTOC_KKT_QXX] + i + j * NX] = f[c][t][r];
                if (t > c) kw[KL.off[RTOC_KKT_QXX] + j + i * NX] = f[c][t][r];
              }
            }
      }
      // ---- P = (F + F^T)/2 (brrf.cpp:85).  Only the upper tiles of F exist; they are written to both
      //      (i,j) and (j,i), and inside the diagonal tiles the upper triangle is mirrored likewise:
      //      P is exactly symmetric, and differs from the arithmetic mean only by the rounding-level
      //      asymmetry the MFMA summation order leaves in a diagonal tile (no LDS round trip). ----
      {
        double* pw_ = sP + q + li * LDP;  // (i,j) at i + j*LDP
        double* pm_ = sP + li + q * LDP;  // (j,i)
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int t = c; t < TNX; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = c * 16 + drow(q, r), j = t * 16 + li;
              if (i < NX && j < NX && (t > c || i <= j)) {
                pw_[c * 16 + 4 * r + t * 16 * LDP] = f[c][t][r];
                if (i != j) pm_[t * 16 + (c * 16 + 4 * r) * LDP] = f[c][t][r];
              }
            }
      }
    } else {
      // next stage's record: HBM -> registers of the vector wave.  Issued here, after the register-
      // hungry solve / constraint code, so that the prefetched values are not spilled.
      if (!impact && ns == 0) {
        // s -= H k = K^T lu' (K^T = -H G^-1); K^T came from the matrix wave before B4
        if (vt < NX) {
          double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
          for (int u = 0; u + 1 < NU; u += 2) {
            acc0 += sKt[vt + u * LDP] * smem[C::V_LU + u];
            acc1 += sKt[vt + (u + 1) * LDP] * smem[C::V_LU + u + 1];
          }
          if (NU & 1) acc0 += sKt[vt + (NU - 1) * LDP] * smem[C::V_LU + NU - 1];
          smem[C::V_SNEW + vt] -= acc0 + acc1;
        }
      }
#ifdef RTOC_RS_LATE_PREFETCH
      if (st > 0) issue_loads(st - 1);
#endif
      RTOC_PROFV(27);
      // vector wave: LQR policy of this stage -> HBM (K row-major == Kt column-major)
      if (!impact) {
        copy_s2g_mat<64, NX, NU, LDP>(rr + RL.off[RTOC_RIC_K], sKt, vt);
        if (vt < NU) {
          rr[RL.off[RTOC_RIC_KV] + vt] = smem[C::V_KV + vt];
          if (sto) {
            rr[RL.off[RTOC_RIC_T] + vt] = smem[C::V_TV + vt];
            rr[RL.off[RTOC_RIC_W] + vt] = smem[C::V_WV + vt];
            rr[RL.off[RTOC_RIC_PSIU] + vt] = smem[C::V_PSIU + vt];
            rr[RL.off[RTOC_RIC_PHIU] + vt] = smem[C::V_PHIU + vt];
          }
        }
      }
    }
    RTOC_PROFV(15);
    RTOC_BLOCK_SYNC();  // B5: P complete in sP

    RTOC_PROF(10);
    RTOC_PROFV(28);
    if (sto) {
#include "riccati_sto_block.inc"
      RTOC_BLOCK_SYNC();
    }

    RTOC_PROF(11);
    // ---- results -> HBM; roll the LDS "next" state ----
    // (P of this stage goes to HBM from the vector wave while it waits for the next stage's G)
    // Done by the vector wave, which produced s / Psi / Phi and is their first consumer in the next
    // stage: no barrier is needed at the next stage top (the matrix wave reads s+ only after the
    // F2 hand-off of that stage, which this wave signals after these writes).
    static_assert(NX <= 64, "one vector-wave lane per state entry");
    if constexpr (!MW) {
      if (vt < NX) {
        const double sv = smem[C::V_SNEW + vt];
        const double psi = sto ? smem[C::V_PSI + vt] : 0.0;
        const double phi = sto ? smem[C::V_PHI + vt] : 0.0;
        rr[RL.off[RTOC_RIC_S] + vt] = sv;
        rr[RL.off[RTOC_RIC_PSI] + vt] = psi;
        rr[RL.off[RTOC_RIC_PHI] + vt] = phi;
        if (sto && !impact) {
          rr[RL.off[RTOC_RIC_PSIX] + vt] = smem[C::V_PSIX + vt];
          rr[RL.off[RTOC_RIC_PHIX] + vt] = smem[C::V_PHIX + vt];
        }
        smem[C::V_SN + vt] = sv;
        smem[C::V_PSIN + vt] = psi;
        smem[C::V_PHIN + vt] = phi;
      }
      if (vt < 5) {
        const double v = sto ? smem[C::V_SC + vt] : 0.0;
        rr[RL.off[RTOC_RIC_SCAL] + vt] = v;
        smem[C::V_SCN + vt] = v;
      }
    }
    RTOC_PROF(12);
    RTOC_PROFV(29);
  }

  // ---- grid[0].sto: trailing phase transition writes sto_policy_[0] (riccati_recursion.cpp:75-79) ----
  RTOC_BLOCK_SYNC();
  if (N >= 1) copy_s2g_mat<NT, NX, NX, LDP>(a.ric + rinst + RL.off[RTOC_RIC_P], sP, tid0);
  {
    const rtoc_grid g0 = a.grid[0];
    if (g0.sto && g0.sto_next) {
      double* pr = a.ric + rinst;
      const double xi = smem[C::V_SCN + 0], chi = smem[C::V_SCN + 1], rho = smem[C::V_SCN + 2],
                   eta = smem[C::V_SCN + 3], iota = smem[C::V_SCN + 4];
      double sgm = xi - 2.0 * chi + rho;
      const double eps = 1.4901161193847656e-08;
      if ((sgm * a.max_dts0) < fabs(eta - iota) || sgm < eps)
        sgm = fabs(sgm) + fabs(eta - iota) / a.max_dts0;
      const double isg = 1.0 / sgm;
      if (tid < NX)
        pr[RL.off[RTOC_RIC_DTSDX] + tid] = -isg * (smem[C::V_PSIN + tid] - smem[C::V_PHIN + tid]);
      if (tid == 0) {
        pr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTSDTS] = isg * (xi - chi);
        pr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0] = -isg * (eta - iota);
      }
    }
  }
  if (stat) atomicOr(&a.status[b], stat);
}

template <int NV, int NU, int NS>
__global__ __launch_bounds__(128, 2) void riccati_backward_rs_kernel(BwdArgs a) {
  if (a.first + (int)blockIdx.x >= a.batch) return;
  extern __shared__ __attribute__((aligned(16))) double smem_all[];
  using C = BwdCfg<NV, NU, NS, 2>;
  if (threadIdx.x < 4) reinterpret_cast<int*>(smem_all + C::V_FLAG)[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < 64)
    riccati_backward_rs_body<NV, NU, NS, true, 1>(a, 0);
  else
    riccati_backward_rs_body<NV, NU, NS, false, 1>(a, 0);
}

// Four instances per 512-thread workgroup, one per SIMD: the matrix wave and the vector wave of
// an instance are the two waves that the dispatcher put on the SAME SIMD.  f64 MFMA on gfx950
// issues through the SIMD's VALU port for its whole 64 cycles (tools/probes/pipe_share_probe.hip:
// a VALU chain of another wave on that SIMD makes no progress while MFMAs stream), so a vector
// wave sharing a SIMD with a FOREIGN matrix wave only runs in that wave's operand-load gaps and
// its dependent chains (Cholesky, solves) stretch 3-4x; sharing with its OWN matrix wave, it runs
// exactly when that wave is waiting for it.  Pairing is read from HW_ID; if the dispatcher ever
// places the waves differently the static pairing (wave, wave+4) is used -- still correct.
template <int NV, int NU, int NS, bool SA = false>
__global__ __launch_bounds__(512) void riccati_backward_rs4_kernel(BwdArgs a) {
  using C = BwdCfg<NV, NU, NS, 2>;
  extern __shared__ __attribute__((aligned(16))) double smem_all[];
  int* const hdr = reinterpret_cast<int*>(smem_all + 4 * C::LDS_DOUBLES);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 4) {
    hdr[threadIdx.x] = 0;
    int* f = reinterpret_cast<int*>(smem_all + threadIdx.x * C::LDS_DOUBLES + C::V_FLAG);
    f[0] = 0;
    f[1] = 0;
    f[2] = 0;
    f[3] = 0;
  }
  __syncthreads();
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const int simd = (hw >> 4) & 3;
  int order = 0;
  if (lane == 0) order = atomicAdd(&hdr[simd], 1);
  order = __builtin_amdgcn_readfirstlane(order);
  __syncthreads();
  const bool paired = hdr[0] == 2 && hdr[1] == 2 && hdr[2] == 2 && hdr[3] == 2;
  const int slot = __builtin_amdgcn_readfirstlane(paired ? simd : (wave & 3));
  const int role = __builtin_amdgcn_readfirstlane(paired ? order : (wave >> 2));
  if (a.first + (int)blockIdx.x * 4 + slot >= a.batch) return;
  if (role == 0)
    riccati_backward_rs_body<NV, NU, NS, true, 4, SA>(a, slot);
  else
    riccati_backward_rs_body<NV, NU, NS, false, 4, SA>(a, slot);
}
#undef RTOC_BLOCK_SYNC

// Does every Fxx of the batch have the structure the SA kernels assume (see riccati_backward_rs_body)?  One wave per
// (instance, grid point); any violation sets *flag.  Reads the top NV rows of Fxx only (~1 GB per 4096 x 47 ANYmal
// records: 0.2 ms), run once per upload of the KKT records, not per sweep.
struct FxxCheckArgs {
  const double* kkt;
  const rtoc_grid* grid;
  int* flag;
  int nstages, batch, nv, np, fxx_off, stride;
};
static __global__ __launch_bounds__(64) void fxx_structure_kernel(FxxCheckArgs a) {
  const int item = blockIdx.x, nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const int nv = a.nv, nx = 2 * nv, np = a.np;
  const double* A = a.kkt + ((size_t)b * a.nstages + st) * a.stride + a.fxx_off;  // column-major nx x nx
  const double ca = A[np + (size_t)np * nx], cc = A[np + (size_t)(nv + np) * nx];
  bool bad = false;
  for (int e = threadIdx.x; e < nv * nx; e += 64) {
    const int i = e % nv, j = e / nv;  // rows [0, nv) of column j
    const double v = A[i + (size_t)j * nx];
    if (i < np) {
      const bool corner = j < np || (j >= nv && j < nv + np);
      if (!corner && v != 0.0) bad = true;
    } else {
      const double want = (j == i) ? ca : ((j == nv + i) ? cc : 0.0);
      if (v != want) bad = true;
    }
  }
  if (__any(bad) && threadIdx.x == 0) atomicOr(a.flag, 1);
}

}  // namespace rtoc

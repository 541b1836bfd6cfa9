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
__device__ __forceinline__ void lds_signal(volatile int* flag, int seq, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) *flag = seq;
}
__device__ __forceinline__ void lds_wait(volatile int* flag, int seq) {
  while (*flag < seq) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

template <int NV, int NU, int NS, bool MW>
__device__ __forceinline__ void riccati_backward_rs_body(const BwdArgs& a) {
  using C = BwdCfg<NV, NU, NS, 2>;
  constexpr int NX = C::NX, NT = 128, LDP = C::LDP, TNX = C::TNX, TMA = C::TMA, TNU = C::TNU;
  constexpr int CNT = TNX;  // the matrix wave owns every 16-tile
  static_assert(NX + 1 <= 64, "role-split kernel needs one vector wave per state vector");
  extern __shared__ __attribute__((aligned(16))) double smem[];
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

  const int tid0 = threadIdx.x;
  const int b = blockIdx.x;
  int tid = tid0, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  const int N = a.nstages - 1;
  const size_t kinst = (size_t)b * a.nstages * a.kl.stride;
  const size_t rinst = (size_t)b * a.nstages * a.rl.stride;
  const int* ko = a.kl.off;
  const int* ro = a.rl.off;
  unsigned stat = 0;

  // ---- terminal stage: P_N = Qxx_N, s_N = -lx_N (riccati_recursion.cpp:37-38) ----
  {
    const double* kr = a.kkt + kinst + (size_t)N * a.kl.stride;
    double* rr = a.ric + rinst + (size_t)N * a.rl.stride;
    copy_g2s_mat<NT, NX, NX, LDP>(sP, kr + ko[RTOC_KKT_QXX], tid);
    if (tid < NX) {
      const double v = -kr[ko[RTOC_KKT_LX] + tid];
      smem[C::V_SN + tid] = v;
      smem[C::V_PSIN + tid] = 0.0;
      smem[C::V_PHIN + tid] = 0.0;
      rr[ro[RTOC_RIC_S] + tid] = v;
    }
    if (tid < 8) smem[C::V_SCN + tid] = 0.0;
    if (tid == 0) sFlag[0] = 0;
    __syncthreads();
    copy_s2g_mat<NT, NX, NX, LDP>(rr + ro[RTOC_RIC_P], sP, tid);
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
      const double* kp = a.kkt + kinst + (size_t)stage * a.kl.stride;
      const bool imp = a.grid[stage].type == RTOC_GRID_IMPACT;
      pre_load_mat<64, NX, NX>(preA, kp + ko[RTOC_KKT_FXX], v_);
      if (!imp) {
        pre_load_mat<64, NX, NU>(preH, kp + ko[RTOC_KKT_QXU], v_);
        pre_load<64, N2B>(preB, kp + ko[RTOC_KKT_FVU], v_);
        pre_load<64, N2G>(preG, kp + ko[RTOC_KKT_QUU], v_);
      }
      if (v_ < NX) {
        preFx = kp[ko[RTOC_KKT_FX] + v_];
        preLx = kp[ko[RTOC_KKT_LX] + v_];
      }
      if (!imp && v_ < NU) preLu = kp[ko[RTOC_KKT_LU] + v_];
    }
  };
  if (N >= 1) issue_loads(N - 1);

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
    const rtoc_grid g = a.grid[st];
    const rtoc_grid gn = a.grid[st + 1];
    const bool impact = (g.type == RTOC_GRID_IMPACT);
    const bool next_lift = (gn.type == RTOC_GRID_LIFT);
    const int ns = impact ? 0 : g.dims;
    const bool sto = g.sto != 0, sto_next = g.sto_next != 0;
    const double* kr = a.kkt + kinst + (size_t)st * a.kl.stride;
    double* rr = a.ric + rinst + (size_t)st * a.rl.stride;

    RTOC_PROF(0);
#include "riccati_pt_block.inc"
    RTOC_PROF(1);
    // ---- stage data: prefetched registers -> LDS (vector wave) ----
    if constexpr (!MW) {
      pre_store_mat<64, NX, NX, LDP>(sA, preA, vt);
      if (!impact) {
        pre_store_flat<64, N2B>(sBv, preB, vt);
        pre_store_mat<64, NX, NU, LDP>(sH, preH, vt);
        pre_store_flat<64, N2G>(sG, preG, vt);
      }
      if (vt < NX) {
        smem[C::V_FX + vt] = preFx;
        smem[C::V_LX + vt] = preLx;
        if (sto) {
          smem[C::V_FFX + vt] = kr[ko[RTOC_KKT_FFX] + vt];
          smem[C::V_HX + vt] = kr[ko[RTOC_KKT_HX] + vt];
        }
      }
      if (!impact && vt < NU) {
        smem[C::V_LU + vt] = preLu;
        if (sto) smem[C::V_HU + vt] = kr[ko[RTOC_KKT_HU] + vt];
      }
    }
    if (sto && tid < 8) smem[C::V_KSC + tid] = kr[ko[RTOC_KKT_SCAL] + tid];
    __syncthreads();  // B1

    RTOC_PROF(2);
    // Qxx of THIS stage straight from HBM into the accumulator registers of the F product, in the
    // MFMA C layout (row = q+4r, col = lane&15).  Issued two intervals before its first use, so the
    // latency hides behind PB / G / PAa; the accumulators are dead until then, so this prefetch
    // costs no extra registers and no LDS staging.
    d4 f[MW ? CNT : 1][MW ? TNX : 1];
    if constexpr (MW) {
      const double* qx_ = kr + ko[RTOC_KKT_QXX] + q + li * NX;
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int t = 0; t < TNX; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = c * 16 + drow(q, r), j = t * 16 + li;
            f[c][t][r] = (i < NX && j < NX) ? qx_[c * 16 + 4 * r + t * 16 * NX] : 0.0;
          }
    }
    // ================= interval 1: [matrix] PB, G      || [vector] z, lu' =================
    if constexpr (!MW) {
      if (vt < NX) {
        double acc = 0.0, accy = 0.0;
#pragma unroll 9
        for (int k = 0; k < NX; ++k) {
          const double p = sP[vt + k * LDP];
          acc += p * smem[C::V_FX + k];
          if (sto) accy += p * smem[C::V_FFX + k];
        }
        smem[C::V_Z + vt] = smem[C::V_SN + vt] - acc;
        if (sto) smem[C::V_Y + vt] = accy + smem[C::V_PSIN + vt];
      }
      wave_lds_sync();
      if (!impact && vt < NU) {
        double acc = 0.0, ap = 0.0, aph = 0.0;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const double bv = sBv[k + vt * NV];
          acc += bv * smem[C::V_Z + NV + k];
          if (sto) {
            ap += bv * smem[C::V_Y + NV + k];
            if (sto_next) aph += bv * smem[C::V_PHIN + NV + k];
          }
        }
        smem[C::V_LU + vt] -= acc;
        if (sto) {
          smem[C::V_PSIU + vt] = ap + smem[C::V_HU + vt];
          smem[C::V_PHIU + vt] = sto_next ? aph : 0.0;
        }
      }
    } else {
     if (!impact) {
      // ---- PB = P+[:,v] Bv ----
      {
        d4 acc[CNT][TNU];
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int t = 0; t < TNU; ++t) acc[c][t] = zero4();
        const double* pa_ = sP + li + (NV + q) * LDP;
        const double* pb_ = sBv + q + li * NV;
#pragma unroll
        for (int ks = 0; ks < (NV + 3) / 4; ++ks) {
          const bool kok = (ks * 4 + 3 < NV) || (ks * 4 + q < NV);
          double bv[TNU];
#pragma unroll
          for (int t = 0; t < TNU; ++t) {
            const double v = pb_[ks * 4 + t * 16 * NV];
            bv[t] = (kok && (t * 16 + li < NU)) ? v : 0.0;
          }
#pragma unroll
          for (int c = 0; c < CNT; ++c) {
            const double v = pa_[c * 16 + ks * 4 * LDP];
            const double av = (kok && (c * 16 + li < NX)) ? v : 0.0;
#pragma unroll
            for (int t = 0; t < TNU; ++t) acc[c][t] = mfma16(av, bv[t], acc[c][t]);
          }
        }
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int t = 0; t < TNU; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = c * 16 + drow(q, r), u = t * 16 + li;
              if (i < NX && u < NU) sPB[i + u * LDP] = acc[c][t][r];
            }
      }
      wave_lds_sync();
      // ---- G = Quu + Bv^T PB[v,:] ----
      {
        d4 acc[TNU][TNU];
#pragma unroll
        for (int t0 = 0; t0 < TNU; ++t0)
#pragma unroll
          for (int t1 = 0; t1 < TNU; ++t1) acc[t0][t1] = zero4();
        const double* pa_ = sBv + q + li * NV;
        const double* pb_ = sPB + NV + q + li * LDP;
#pragma unroll
        for (int ks = 0; ks < (NV + 3) / 4; ++ks) {
          const bool kok = (ks * 4 + 3 < NV) || (ks * 4 + q < NV);
          double av[TNU], bv[TNU];
#pragma unroll
          for (int t = 0; t < TNU; ++t) {
            const bool ok = kok && (t * 16 + li < NU);
            const double va = pa_[ks * 4 + t * 16 * NV];
            const double vb = pb_[ks * 4 + t * 16 * LDP];
            av[t] = ok ? va : 0.0;
            bv[t] = ok ? vb : 0.0;
          }
#pragma unroll
          for (int t0 = 0; t0 < TNU; ++t0)
#pragma unroll
            for (int t1 = 0; t1 < TNU; ++t1) acc[t0][t1] = mfma16(av[t0], bv[t1], acc[t0][t1]);
        }
#pragma unroll
        for (int t0 = 0; t0 < TNU; ++t0)
#pragma unroll
          for (int t1 = 0; t1 < TNU; ++t1)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int u0 = t0 * 16 + drow(q, r), u1 = t1 * 16 + li;
              if (u0 < NU && u1 < NU) sG[u0 + u1 * NU] += acc[t0][t1][r];
            }
      }
     } else {
      for (int e = lane; e < NU * LDP; e += 64) sPB[e] = 0.0;
     }
      lds_signal(sFlag, 2 * (N - st) - 1, lane);  // G ready
    }
    if constexpr (!MW) lds_wait(sFlag, 2 * (N - st) - 1);

    RTOC_PROF(3);
    // ================= interval 2: [matrix] PAa, H     || [vector] LLT(G), w = A^T z =========
    d4 pa[MW ? TMA : 1][MW ? CNT : 1];
    if constexpr (MW) {
#pragma unroll
      for (int tm = 0; tm < TMA; ++tm)
#pragma unroll
        for (int c = 0; c < CNT; ++c) pa[tm][c] = zero4();
      const double* pb_ = sA + q + li * LDP;
      // operands of k-step ks: software-pipelined one step ahead (the scheduler otherwise hoists the
      // LDS loads of all nine steps at once and spills)
      auto load_ops = [&](int ks, double (&av)[TMA], double (&bv)[CNT]) {
        const bool kok = (ks * 4 + 3 < NX) || (ks * 4 + q < NX);
#pragma unroll
        for (int tm = 0; tm < TMA; ++tm) {
          const int i = tm * 16 + li;
          double v;
          if (tm * 16 + 15 < NX) {
            v = sP[li + q * LDP + tm * 16 + ks * 4 * LDP];
          } else if (tm * 16 >= NX) {
            v = sPB[q + li * LDP + (tm * 16 - NX) * LDP + ks * 4];
            if (tm * 16 + 15 >= NX + NU) v = (i < NX + NU) ? v : 0.0;
          } else {
            const double vp = sP[li + q * LDP + tm * 16 + ks * 4 * LDP];
            const double vb = sPB[q + li * LDP + (tm * 16 - NX) * LDP + ks * 4];
            v = (i < NX) ? vp : vb;
            if (tm * 16 + 15 >= NX + NU) v = (i < NX + NU) ? v : 0.0;
          }
          av[tm] = kok ? v : 0.0;
        }
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
          const double v = pb_[ks * 4 + c * 16 * LDP];
          bv[c] = (kok && (c * 16 + li < NX)) ? v : 0.0;
        }
      };
      constexpr int KSA = (NX + 3) / 4;
      double av[2][TMA], bv[2][CNT];
      load_ops(0, av[0], bv[0]);
#pragma unroll
      for (int ks = 0; ks < KSA; ++ks) {
        if (ks + 1 < KSA) load_ops(ks + 1, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
#pragma unroll
        for (int tm = 0; tm < TMA; ++tm)
#pragma unroll
          for (int c = 0; c < CNT; ++c) pa[tm][c] = mfma16(av[ks & 1][tm], bv[ks & 1][c], pa[tm][c]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!impact) {
#pragma unroll
        for (int tm = 0; tm < TMA; ++tm)
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
    } else {
      if (!impact) {
        if (wave_llt<NU, NU>(sG, sL, smem + C::V_LINV, NU, lane)) stat |= RTOC_STAT_QUU_NOT_SPD;
      }
      if (vt < NX) {
        double acc = 0.0, ap = 0.0, aph = 0.0;
#pragma unroll 9
        for (int k = 0; k < NX; ++k) {
          const double av = sA[k + vt * LDP];
          acc += av * smem[C::V_Z + k];
          if (sto) {
            if (!impact) ap += av * smem[C::V_Y + k];
            aph += av * smem[C::V_PHIN + k];
          }
        }
        smem[C::V_SNEW + vt] = acc - smem[C::V_LX + vt];
        if (sto) {
          if (!impact) {
            smem[C::V_PSIX + vt] = ap + smem[C::V_HX + vt];
            smem[C::V_PHIX + vt] = sto_next ? aph : 0.0;
          } else {
            smem[C::V_PHIX + vt] = aph;
          }
        }
      }
    }
    if constexpr (MW) lds_signal(sFlag, 2 * (N - st), lane);  // H ready, PB no longer read
    if constexpr (!MW) lds_wait(sFlag, 2 * (N - st));

    RTOC_PROF(4);
    RTOC_PROF(5);
    // ================= interval 3: [matrix] F = Qxx + AtP A  || [vector] K, k solve ==========
    if constexpr (MW) {
      const double* pbf_ = sA + q + li * LDP;
      // k runs over the register groups (tm, r) of PAa that hold P rows: g = 4*tm + r < KSF
      constexpr int KSF = (NX + 3) / 4;
      auto load_b = [&](int gidx, double (&bv)[TNX]) {
        const bool kok = (gidx * 4 + 3 < NX) || (gidx * 4 + q < NX);
#pragma unroll
        for (int t = 0; t < TNX; ++t) {
          const double v = pbf_[gidx * 4 + t * 16 * LDP];
          bv[t] = (kok && (t * 16 + li < NX)) ? v : 0.0;
        }
      };
      double bvf[2][TNX];
      load_b(0, bvf[0]);
#pragma unroll
      for (int gidx = 0; gidx < KSF; ++gidx) {
        if (gidx + 1 < KSF) load_b(gidx + 1, bvf[(gidx + 1) & 1]);
        const bool kok = (gidx * 4 + 3 < NX) || (gidx * 4 + q < NX);
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
          const double avv = kok ? pa[gidx / 4][c][gidx % 4] : 0.0;
#pragma unroll
          for (int t = 0; t < TNX; ++t) f[c][t] = mfma16(avv, bvf[gidx & 1][t], f[c][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if (!impact && ns == 0) {
        // K = -G^-1 H^T, k = -G^-1 lu (riccati_factorizer.cpp:55-56)
        if (vt <= NX) {
          double x[NU];
#pragma unroll
          for (int u = 0; u < NU; ++u) x[u] = (vt < NX) ? sH[vt + u * LDP] : smem[C::V_LU + u];
          llt_solve_reg<NU, NU>(sL, smem + C::V_LINV, x, NU);
          bool bad = false;
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            bad = bad || is_bad(x[u]);
            if (vt < NX)
              sKt[vt + u * LDP] = -x[u];
            else
              smem[C::V_KV + u] = -x[u];
          }
          if (bad) stat |= RTOC_STAT_NAN;
          if (sto && vt == NX) {
            double t[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) t[u] = smem[C::V_PSIU + u];
            llt_solve_reg<NU, NU>(sL, smem + C::V_LINV, t, NU);
#pragma unroll
            for (int u = 0; u < NU; ++u) smem[C::V_TV + u] = -t[u];
            if (sto_next) {
#pragma unroll
              for (int u = 0; u < NU; ++u) t[u] = smem[C::V_PHIU + u];
              llt_solve_reg<NU, NU>(sL, smem + C::V_LINV, t, NU);
#pragma unroll
              for (int u = 0; u < NU; ++u) smem[C::V_WV + u] = -t[u];
            } else {
#pragma unroll
              for (int u = 0; u < NU; ++u) smem[C::V_WV + u] = 0.0;
            }
          }
        }
        wave_lds_sync();
        // s -= H k = K^T lu' (K^T = -H G^-1)
        if (vt < NX) {
          double acc = 0.0;
#pragma unroll
          for (int u = 0; u < NU; ++u) acc += sKt[vt + u * LDP] * smem[C::V_LU + u];
          smem[C::V_SNEW + vt] -= acc;
        }
      }
    }
    __syncthreads();  // B4: K, k ready; F product done; A and H(after SC) free

    RTOC_PROF(6);
    if (NS > 0 && ns > 0) {
#include "riccati_sc_block.inc"
      __syncthreads();
    }
    RTOC_PROF(7);
    if (a.writeback && !impact) {
      double* kw = a.kkt_rw + kinst + (size_t)st * a.kl.stride;
      copy_s2g_mat<NT, NX, NU, LDP>(kw + ko[RTOC_KKT_QXU], sH, tid);
      copy_s2g_flat<NT>(kw + ko[RTOC_KKT_QUU], sG, NU * NU, tid);
      if (tid < NU) kw[ko[RTOC_KKT_LU] + tid] = smem[C::V_LU + tid];
      __syncthreads();
    }

    // ================= interval 4: [matrix] GK, F -= K^T GK, P   || [vector] policy -> HBM ====
    if constexpr (MW) {
      if (!impact) {
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
                  double v = acc[t][c][r];
                  if (NS > 0 && ns > 0) {
                    double dtm = 0.0;
                    for (int l = 0; l < ns; ++l)
                      dtm += smem[C::S_PHIU + l + u * C::NSP] * smem[C::S_M + l + j * C::NSP];
                    v += 2.0 * dtm;
                  }
                  sGK[u + j * NU] = v;
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
              for (int t = 0; t < TNX; ++t) f[c][t] = mfma16(av[c], bv[t], f[c][t]);
          }
        }
      }
      RTOC_PROF(9);
      if (a.writeback) {
        double* kw = a.kkt_rw + kinst + (size_t)st * a.kl.stride;
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int t = 0; t < TNX; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = c * 16 + drow(q, r), j = t * 16 + li;
              if (i < NX && j < NX) kw[ko[RTOC_KKT_QXX] + i + j * NX] = f[c][t][r];
            }
      }
      // ---- P = (F + F^T)/2 in the MFMA register layout (brrf.cpp:85) ----
      {
        double* pw_ = sP + q + li * LDP;
        const double* pr_ = sP + li + q * LDP;
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int t = 0; t < TNX; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = c * 16 + drow(q, r), j = t * 16 + li;
              if (i < NX && j < NX) pw_[c * 16 + 4 * r + t * 16 * LDP] = f[c][t][r];
            }
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int t = 0; t < TNX; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double v = pr_[t * 16 + (c * 16 + 4 * r) * LDP];
              f[c][t][r] = 0.5 * (f[c][t][r] + v);
            }
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int t = 0; t < TNX; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = c * 16 + drow(q, r), j = t * 16 + li;
              if (i < NX && j < NX) pw_[c * 16 + 4 * r + t * 16 * LDP] = f[c][t][r];
            }
      }
    } else {
      // next stage's record: HBM -> registers of the vector wave.  Issued here, after the register-
      // hungry solve / constraint code, so that the prefetched values are not spilled.
      if (st > 0) issue_loads(st - 1);
      // vector wave: LQR policy of this stage -> HBM (K row-major == Kt column-major)
      if (!impact) {
        copy_s2g_mat<64, NX, NU, LDP>(rr + ro[RTOC_RIC_K], sKt, vt);
        if (vt < NU) {
          rr[ro[RTOC_RIC_KV] + vt] = smem[C::V_KV + vt];
          if (sto) {
            rr[ro[RTOC_RIC_T] + vt] = smem[C::V_TV + vt];
            rr[ro[RTOC_RIC_W] + vt] = smem[C::V_WV + vt];
            rr[ro[RTOC_RIC_PSIU] + vt] = smem[C::V_PSIU + vt];
            rr[ro[RTOC_RIC_PHIU] + vt] = smem[C::V_PHIU + vt];
          }
        }
      }
    }
    __syncthreads();  // B5: P complete in sP

    RTOC_PROF(10);
    if (sto) {
#include "riccati_sto_block.inc"
      __syncthreads();
    }

    RTOC_PROF(11);
    // ---- results -> HBM; roll the LDS "next" state ----
    copy_s2g_mat<NT, NX, NX, LDP>(rr + ro[RTOC_RIC_P], sP, tid);
    if (tid < NX) {
      const double sv = smem[C::V_SNEW + tid];
      const double psi = sto ? smem[C::V_PSI + tid] : 0.0;
      const double phi = sto ? smem[C::V_PHI + tid] : 0.0;
      rr[ro[RTOC_RIC_S] + tid] = sv;
      rr[ro[RTOC_RIC_PSI] + tid] = psi;
      rr[ro[RTOC_RIC_PHI] + tid] = phi;
      if (sto && !impact) {
        rr[ro[RTOC_RIC_PSIX] + tid] = smem[C::V_PSIX + tid];
        rr[ro[RTOC_RIC_PHIX] + tid] = smem[C::V_PHIX + tid];
      }
      smem[C::V_SN + tid] = sv;
      smem[C::V_PSIN + tid] = psi;
      smem[C::V_PHIN + tid] = phi;
    }
    if (tid < 5) {
      const double v = sto ? smem[C::V_SC + tid] : 0.0;
      rr[ro[RTOC_RIC_SCAL] + tid] = v;
      smem[C::V_SCN + tid] = v;
    }
    RTOC_PROF(12);
  }

  // ---- grid[0].sto: trailing phase transition writes sto_policy_[0] (riccati_recursion.cpp:75-79) ----
  __syncthreads();
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
        pr[ro[RTOC_RIC_DTSDX] + tid] = -isg * (smem[C::V_PSIN + tid] - smem[C::V_PHIN + tid]);
      if (tid == 0) {
        pr[ro[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTSDTS] = isg * (xi - chi);
        pr[ro[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0] = -isg * (eta - iota);
      }
    }
  }
  if (stat) atomicOr(&a.status[b], stat);
}

template <int NV, int NU, int NS>
__global__ __launch_bounds__(128, 2) void riccati_backward_rs_kernel(BwdArgs a) {
  if ((int)blockIdx.x >= a.batch) return;
  if (threadIdx.x < 64)
    riccati_backward_rs_body<NV, NU, NS, true>(a);
  else
    riccati_backward_rs_body<NV, NU, NS, false>(a);
}

}  // namespace rtoc

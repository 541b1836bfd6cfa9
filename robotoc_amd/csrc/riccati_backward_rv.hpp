// riccati_backward_rv.hpp -- backward Riccati recursion with the value function in REGISTERS: one wavefront per OCP instance.
//
// Same recursion and HBM record contract as riccati_backward.hpp (reference: src/riccati/riccati_recursion.cpp:32-80,
// riccati_factorizer.cpp:44-56,178-197, backward_riccati_recursion_factorizer.cpp:31-91), different machine mapping.  The
// role-split kernel (riccati_backward_rs.hpp) keeps P+, A, PB, H, K^T and a dozen vectors of an instance in 39 KB of LDS, so a
// CU holds four instances -- ONE dependency chain per SIMD, which waits 41 % of its cycles (profiles/r04_rocprof_summary.txt).
// Here an instance is one wave and needs 20 KB of LDS (A, Bv, Quu, three vectors, the inverse Cholesky factor, a transpose
// scratch): EIGHT instances per CU, two independent chains per SIMD, no hand-off between waves at all.
//
//   * P+ lives in 9 (= T x T) accumulator tiles in the f64 MFMA C layout (lane (li, q), register r <-> row q + 4r, column li).
//     For a symmetric matrix the C layout of tile (kt, mt) IS the A-operand fragment (m = 16 mt + li, k = 16 kt + 4r + q) of
//     k-step (kt, r): P+ [.] and [P+; PB^T] [.] run straight from registers -- no LDS copy of P+, no operand reads for it.
//   * PB = P+[:, v] Bv is computed with its control columns shifted by SCOL lanes (u <-> lane li = u + SCOL, SCOL = NX - 16 (T - 1)):
//     then the accumulators of PB are, lane for lane, the A-operand fragments of the PB^T rows (NX.. of the stacked operand
//     [P+; PB^T], which fills its T row tiles exactly: 36 + 12 = 48) -- PB never leaves the registers either.
//   * The accumulators of the PB^T rows start from Qxu^T (loaded from HBM in the C layout), so they END as H^T = Qxu^T + PB^T A,
//     which is the B-operand fragment of the policy product Z^T = L^-1 H^T: H never exists in LDS.
//   * Every vector rides as a column of a matrix product: Bv^T s+_v as column 0 of the G product; Fx as column NX of [A | Fx]
//     (-> P+ Fx and PB^T Fx); z = s+ - P+ Fx replaces that column in the accumulators, and column NX of F = Qxx + A^T W -- started
//     from -lx -- is then w = A^T z - lx; -lu' rides as column NX of H^T, so that column NX of K is -k and column NX of
//     F -= Z Z^T receives + H G^-1 lu' = -H k: after the last product column NX of F IS the new s.  s+ is carried from stage to
//     stage in exactly the registers it is produced in (lanes li = SCOL of the last column tile).
//   * The stage record comes in by LDS-DMA (global_load_lds_dwordx4: no prefetch registers, no VALU unpack): A into a padded
//     (conflict-free) layout by per-lane source addresses, Bv | Quu | Fx lx lu as one strip; Qxx and Qxu^T by per-lane loads
//     straight into accumulators.  The DMA of stage st - 1 is issued when the F product of stage st has read A for the last time.
//   * new P = sym(F): upper tiles computed, diagonal tiles mirrored, lower tiles transposed through a 16 x 17 LDS scratch;
//     P goes to HBM from the registers during the NEXT stage (coalesced through the symmetric index pair).
//
// Scope (rv_applies, rtoc_capi.hip): shapes whose stacked operand fills its tiles (RvCfg::OK: nv = 18, nu = 12), horizons of at most
// RV_MAX_STAGES grid points, RTOC_OPT_WRITEBACK_KKT = 0, the default RTOC_OPT_BACKWARD_WAVES.  ONE launch walks the whole horizon
// (seg_hi = N - 1, seg_lo = 0): regular, lift, impact and switching-constraint grid points (factorised Schur form, below) are all
// the kernel's own; grids with switching-time optimisation run the STO instantiation (structured Fxx only; the phase transition
// and the five scalars on the carried rider lanes).  Elsewhere the role-split / tile-split kernels run.  tools/rv_model.py states
// the lane algebra in numpy against the oracle.
#pragma once
#include "riccati_backward.hpp"

namespace rtoc {

template <int NV, int NU>
struct RvCfg {
  static constexpr int NX = 2 * NV;
  static constexpr int T = (NX + 15) / 16;            // 16-tiles of the state, and row tiles of [P+; PB^T]
  static constexpr int LDP = lds_ld(NX);
  static constexpr int KG = (NX + 3) / 4;             // aligned k groups of the state
  static constexpr int KSU = (NU + 3) / 4;
  static constexpr int SCOL = NX - 16 * (T - 1);      // lane of column NX in the last column tile = lane shift of the controls
  static constexpr int G0 = NV / 4;                   // first k group that meets the rows [NV, NX)
  static constexpr int GS = 4 * (T - 1) + SCOL / 4;   // k group of row NX (the first PB^T row) of the stacked operand
  static constexpr bool OK = (NX + NU == 16 * T) && (NX % 4 == 0) && (NU % 4 == 0) && (SCOL % 4 == 0) && (SCOL >= 1) &&
                             ((NV * NU) % 2 == 0) && (LDP % 2 == 0);
  // ---- LDS carve (doubles) ----
  static constexpr int HL = LDP / 2;                  // 16-byte chunks per padded column of A
  static constexpr int NCH_A = NX * HL;               // chunk slots of A (one per column is padding)
  static constexpr int PA = (NCH_A + 63) / 64;        // DMA pieces (one wave instruction = 64 chunks = 1 KiB)
  static constexpr int OFF_A = 0;
  static constexpr int OFF_ST = NX * LDP;             // strip: Bv | Quu | Fx .. lu (as they lie in the record)
  static constexpr int CB = NV * NU / 2, CG = NU * NU / 2;
  static constexpr int pad8(int n) { return (n + 7) & ~7; }
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
// LDS destinations of the DMA as 32-bit addresses relative to the kernel's shared array: casting a generic pointer to the LDS address
// space costs a null check (64-bit compare + select into M0) per request once the pointer has passed through a lambda capture.  Used
// by riccati_backward_rw2.hpp, whose wave 1 issues 51 requests in the tail of a stage; in this file's kernel (and in _rw.hpp) it made no
// measurable difference and is not used.  Only from __device__ functions: the host pass of a __global__ function that calls it drops
// the kernel's stub without a diagnostic.
struct LdsAddr {
  unsigned a;
  const char* g;
  __device__ __forceinline__ explicit LdsAddr(const void* smem) : a((unsigned)(uintptr_t)(lds_ptr_t)smem), g((const char*)smem) {}
  __device__ __forceinline__ lds_ptr_t operator()(const void* p) const { return (lds_ptr_t)(uintptr_t)(a + (unsigned)((const char*)p - g)); }
};
constexpr int RV_MAX_STAGES = 64;   // grid points of a horizon the kernel keeps a kind table for (in the slack of its LDS carve)

// Streaming policy of the record traffic (every byte is touched once per sweep): RTOC_RV_NT = 1 marks the loads / DMA / stores
// non-temporal.  Measured both ways.
#ifndef RTOC_RV_NT
#define RTOC_RV_NT 0
#endif
// RTOC_RV_MERGE_PBT = 1: PB^T rides in the idle lanes of P+'s last column tile during the W products (see the stage loop)
// RTOC_RV_ONE_SCRATCH = 1: the transposes at the stage end go through ONE scratch tile (six round trips instead of three)
#ifndef RTOC_RV_ONE_SCRATCH
#define RTOC_RV_ONE_SCRATCH 0
#endif
#ifndef RTOC_RV_MERGE_PBT
#define RTOC_RV_MERGE_PBT 1
#endif
__device__ __forceinline__ double rv_ld(const double* p) {
#if RTOC_RV_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void rv_st(double* p, double v) {
#if RTOC_RV_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
constexpr int RV_DMA_AUX = RTOC_RV_NT ? 2 : 0;

// v <- the value `n` lanes to the RIGHT / LEFT within the 16-lane row (v_mov_b32_dpp row_shl / row_shr), 0 beyond the row.
template <int N>
__device__ __forceinline__ double dpp_from_right(double v) {   // lane li reads lane li + N
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + N, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + N, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ double dpp_from_left(double v) {    // lane li reads lane li - N
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + N, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + N, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// the n-th upper tile (c, t) of a T x T tiling in the order the transposes take them: off-diagonal tiles first, then the diagonal
__host__ __device__ constexpr int rv_tile_row(int T, int n) {
  int k = 0;
  for (int c = 0; c < T; ++c)
    for (int t = c + 1; t < T; ++t) {
      if (k == n) return c;
      ++k;
    }
  return n - k;
}
__host__ __device__ constexpr int rv_tile_col(int T, int n) {
  int k = 0;
  for (int c = 0; c < T; ++c)
    for (int t = c + 1; t < T; ++t) {
      if (k == n) return t;
      ++k;
    }
  return n - k;
}

__device__ __forceinline__ void rv_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// cycle stamps of instance 0 (tools/phase_profile_rv.py; compiled in with make PROF=1 only)
#ifdef RTOC_ENABLE_PROF
#define RV_PROF(k)                                                                                        \
  do {                                                                                                    \
    if (a.prof && b == 0 && lane0 == 0) a.prof[st * 32 + (k)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define RV_PROF(k) do { } while (0)
#endif

// SA ("structured A", RTOC_OPT_FXX_STRUCTURE; the caller has checked it: rtoc_check_fxx_structure): rows [NP, NV) of Fxx are
// a e_k^T | c e_k^T and rows [0, NP) are dense only in the two NP x NP corners (src/dynamics/state_equation.cpp:52-55,80-82).  Then
//   W = [P+; PB^T] A:  for the P+ row tiles of the column tiles without riders, the k groups made of structured rows only are
//                      skipped and the structured rows masked out of the others; their part is W[:, k] += a S[:, k],
//                      W[:, NV + k] += c S[:, k] -- scaled copies of P+ tiles in place (a) or two lanes to the side (c);
//   F += A^T W:        the same k groups skipped / rows masked (and the corner rows only multiplied into the corner column tiles);
//                      their part is F[k][:] += a W[k][:], F[NV + k][:] += c W[k][:] -- the same register, or two q-groups away.
// 114 instead of 135 MFMAs in the two NX^3 products (tools/rv_model.py --dense shows the other count).
// STO: grids with switching-time optimisation (riccati_factorizer.cpp:93-175, brrf.cpp:48-66,94-143,160-174).  Psi+ and Phi+ ride beside
// s+ as columns NX + 1, NX + 2 of every product the vector algebra shares with s (lanes SCOL + 1, SCOL + 2 of the last column
// tile): W's column NX + 1 is P+ fx -> y = P+ fx + Psi+, F's columns start from hx / 0 and end as Psi / Phi, the G product's idle
// lanes 1, 2 give Bv^T Psi+_v, Bv^T Phi+_v, H^T's columns carry +psi_u, +phi_u so that K's columns are T and W, the switching
// constraint's columns -Phit, 0 so that M's are mt, mt_next; the five scalars are eleven dot products of those lanes; the phase
// transition acts on the carried lanes at the stage top.  Its own instantiation: the headline kernel pays nothing for it.
template <int NV, int NU, int NS, bool SA, bool STO = false>
__global__ __launch_bounds__(64, 2) void riccati_backward_rv_kernel(BwdArgs a) {
  using C = RvCfg<NV, NU>;
  static_assert(C::OK, "shape does not fill the tiles of the stacked operand");
  constexpr int NP_ = NV - NU;
  static_assert(!SA || (NV % 16 == 2 && C::T == 3 && NP_ > 0 && NP_ <= 8 && NP_ % 2 == 0),
                "structured form: the lane shift of the c-part is NV mod 16 = 2, corner columns in tiles 0 and 1");
  constexpr int NX = C::NX, T = C::T, LDP = C::LDP, KG = C::KG, KSU = C::KSU, SCOL = C::SCOL, G0 = C::G0, GS = C::GS;
  constexpr int RS = SCOL / 4;   // first register group of the last row tile that holds PB^T rows
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt, RL = SL.ric;
  // strip layout
  constexpr int VOFF_LX = KL.off[RTOC_KKT_LX] - KL.off[RTOC_KKT_FX], VOFF_LU = KL.off[RTOC_KKT_LU] - KL.off[RTOC_KKT_FX];
  static_assert(VOFF_LX > 0 && VOFF_LU > VOFF_LX && VOFF_LX % 2 == 0 && VOFF_LU % 2 == 0, "Fx, lx, lu lie behind one another in the record");
  static_assert(KL.off[RTOC_KKT_FXX] % 2 == 0 && KL.off[RTOC_KKT_FVU] % 2 == 0 && KL.off[RTOC_KKT_QUU] % 2 == 0 && KL.off[RTOC_KKT_FX] % 2 == 0 &&
                    KL.stride % 2 == 0, "16-byte chunks");
  // STO: the strip runs on through fx (FFX), hx, hu and the scalars [Qtt, Qtt_prev, h], which lie behind lu in the record
  constexpr int VOFF_FFX = KL.off[RTOC_KKT_FFX] - KL.off[RTOC_KKT_FX], VOFF_HX = KL.off[RTOC_KKT_HX] - KL.off[RTOC_KKT_FX],
                VOFF_HU = KL.off[RTOC_KKT_HU] - KL.off[RTOC_KKT_FX], VOFF_SCAL = KL.off[RTOC_KKT_SCAL] - KL.off[RTOC_KKT_FX];
  static_assert(!STO || (VOFF_FFX > VOFF_LU && VOFF_HX > VOFF_FFX && VOFF_HU > VOFF_HX && VOFF_SCAL > VOFF_HU && VOFF_SCAL % 2 == 0), "fx, hx, hu, scalars behind lu");
  static_assert(!STO || SCOL + 2 < 16, "two more rider lanes in the last column tile");
  static_assert(!STO || SCOL >= 3, "lanes 0..2 of the G product are idle (the controls are shifted by SCOL lanes)");
  constexpr int CB = C::CB, CG = C::CG, CV = STO ? (VOFF_SCAL + 8) / 2 : (VOFF_LU + NU + 1) / 2, NCH_S = CB + CG + CV, PS = (NCH_S + 63) / 64;
  constexpr int ST_BV = C::OFF_ST, ST_G = ST_BV + 2 * CB, ST_FX = ST_G + 2 * CG, ST_LX = ST_FX + VOFF_LX, ST_LU = ST_FX + VOFF_LU;
  constexpr int ST_FFX = ST_FX + VOFF_FFX, ST_HX = ST_FX + VOFF_HX, ST_HU = ST_FX + VOFF_HU, ST_SCAL = ST_FX + VOFF_SCAL;
  // STO: ONE transpose tile (the longer strip takes the other's place in the 20 KB); the switching constraint's S, Ls, Ws then live
  // in the place of A, and the DMA of the next record waits for them on those grid points
  constexpr int NSCR = (STO || RTOC_RV_ONE_SCRATCH) ? 1 : 2;
  constexpr int OFF_Y = ST_FX + 2 * CV, OFF_LINV = OFF_Y + C::pad8(NU * NU), OFF_SCR = OFF_LINV + C::pad8(NU), SCR_LD = 17,
                SCR_TILE = C::pad8(16 * SCR_LD), LDS_DOUBLES = OFF_SCR + NSCR * SCR_TILE;
  static_assert(NU * NU <= NSCR * SCR_TILE, "the Cholesky factor is parked in the transpose scratch");
  static_assert(!STO || 2 * C::pad8(NX) + NX <= SCR_TILE, "s | Psi | Phi gathered in the scratch");
  // grid-point kinds of the whole horizon, one int each (type | dims << 8), behind the carve: read with a DS instruction at the stage
  // top.  (A vector load carried over the loop's back edge makes the compiler open every stage with s_waitcnt vmcnt(0), which also
  // waits for the stores the counted wait below leaves in flight; a scalar load shares its counter with the LDS traffic.)
  constexpr int OFF_GRID = LDS_DOUBLES, GRID_INTS = 2 * (20 * 1024 / 8 - LDS_DOUBLES);
  static_assert(GRID_INTS >= 64, "room for the grid table");
  static_assert(LDS_DOUBLES * 8 <= 20 * 1024, "eight instances per CU");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const sA = smem + C::OFF_A;
  double* const sBv = smem + ST_BV;
  double* const sG = smem + ST_G;
  double* const sY = smem + OFF_Y;
  double* const scr = smem + OFF_SCR;

  const int b = a.first + (int)blockIdx.x;
  if (b >= a.batch) return;
  const int lane0 = threadIdx.x;
  int lane = lane0 & 63, li = lane & 15, q = lane >> 4;   // (& 63: tells the compiler li < 16, q < 4 -- the tile predicates fold)
  const int N = a.nstages - 1;
  const size_t kinst = (size_t)b * a.nstages * KL.stride;
  const size_t rinst = (size_t)b * a.nstages * RL.stride;
  const int hi = a.seg_hi, lo = a.seg_lo;
  unsigned stat = 0;
#ifdef RTOC_RV_DEBUG_MASK   // timing experiments (wrong results): bit 0 no Qxx loads, 1 no P stores, 2 no record DMA, 3 no K stores, 4 no Qxu loads
  const int dbg = a.scan_ps_soff;
#define RV_DBG(bit) (dbg & (1 << (bit)))
#else
#define RV_DBG(bit) false
#endif

  // The record of grid point `stage` -> LDS by DMA (16 B per lane and instruction, LDS address = piece base + 16 lane).
  auto issue_dma = [&](int stage) {
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride;
#pragma unroll
    for (int p = 0; p < PS; ++p) {   // the strip first: the PB / G products at the stage top need Bv and Quu
      const int n = lane + 64 * p;
      const int off = (n < CB) ? (KL.off[RTOC_KKT_FVU] + 2 * n)
                               : ((n < CB + CG) ? (KL.off[RTOC_KKT_QUU] + 2 * (n - CB)) : (KL.off[RTOC_KKT_FX] + 2 * (n - CB - CG)));
      if (64 * (p + 1) <= NCH_S || n < NCH_S)   // (only the last piece is partial)
        __builtin_amdgcn_global_load_lds(kp + off, (lds_ptr_t)(smem + C::OFF_ST + 128 * p), 16, 0, RV_DMA_AUX);
    }
    // A: CPP whole padded columns per piece (the other lanes idle), so that the per-lane source offset is the same in every
    // piece and a piece differs from the next by a scalar: no per-piece address arithmetic on the vector unit
    constexpr int CPP = 64 / C::HL, NPC = (NX + CPP - 1) / CPP;
    const unsigned col0 = (unsigned)lane / C::HL, ch0 = (unsigned)lane - col0 * C::HL;
    const unsigned voff = col0 * NX + ((ch0 < NX / 2) ? 2 * ch0 : 0);   // (the padding chunk of a column loads anything)
    if (col0 < CPP) {
#pragma unroll
      for (int p = 0; p < NPC; ++p) {
        const double* kpp = kp + KL.off[RTOC_KKT_FXX] + p * CPP * NX;
        if ((p + 1) * CPP <= NX || p * CPP + (int)col0 < NX)
          __builtin_amdgcn_global_load_lds(kpp + voff, (lds_ptr_t)(sA + p * CPP * LDP), 16, 0, RV_DMA_AUX);
      }
    }
  };
  // Qxu^T of grid point `stage` in the C layout of the PB^T rows: hq[c][ks] = Qxu[x = 16c + li][u = 4 ks + q].  RAW: lanes beyond the
  // matrix load a clamped address and are masked where hq is USED -- a select here would wait for the loads on the spot.
  double hq[T][KSU];
  auto issue_hq = [&](int stage) {
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride + KL.off[RTOC_KKT_QXU];
#pragma unroll
    for (int c = 0; c < T; ++c)
#pragma unroll
      for (int ks = 0; ks < KSU; ++ks) {
        const int x = 16 * c + li, u = 4 * ks + q;
        hq[c][ks] = rv_ld(kp + ((x < NX) ? x : NX - 1) + u * NX);   // (a distinct address per ks: a common clamp target becomes a branch)
      }
  };

  // ---- value function of grid point hi + 1 -> registers: the terminal one (P_N = Qxx_N, s_N = -lx_N,
  //      riccati_recursion.cpp:37-38) or the one a previous segment left in the Riccati records ----
  d4 pp[T][T];
  double sv[KG];
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
          const double v = psrc[ok ? i + j * NX : 0];
          pp[kt][mt][r] = ok ? v : 0.0;
        }
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const double v = ssrc[4 * g + q];
      sv[g] = (li == SCOL) ? (term ? -v : v) : 0.0;
    }
    if (term) {
      double* rr = a.ric + rinst + (size_t)N * RL.stride;
      const d2* s2 = reinterpret_cast<const d2*>(psrc);
      d2* t2 = reinterpret_cast<d2*>(rr + RL.off[RTOC_RIC_P]);
      for (int e = lane; e < NX * NX / 2; e += 64) t2[e] = s2[e];
      if (lane < NX) rr[RL.off[RTOC_RIC_S] + lane] = -ssrc[lane];
    }
  }
  issue_dma(hi);
  int* const sGrid = reinterpret_cast<int*>(smem + OFF_GRID);
  for (int e = lane; e < a.nstages; e += 64)   // (the host keeps nstages <= RV_MAX_STAGES)
    sGrid[e] = (a.grid[e].type & 15) | ((a.grid[e].sto != 0) << 4) | ((a.grid[e].sto_next != 0) << 5) | (a.grid[e].dims << 8);
  double sc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // STO: xi, chi, rho, eta, iota of grid point st + 1 (uniform)
  auto phase_transition = [&](int pol, bool has_next) __attribute__((always_inline)) {
    double* pr = a.ric + rinst + (size_t)pol * RL.stride;
    const double xi = sc[0], chi = sc[1], rho = sc[2], eta = sc[3], iota = sc[4];
    double isg = 0.0;
    if (has_next) {
      double sgm = xi - 2.0 * chi + rho;
      const double eps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON)
      if ((sgm * a.max_dts0) < fabs(eta - iota) || sgm < eps) sgm = fabs(sgm) + fabs(eta - iota) / a.max_dts0;
      isg = 1.0 / sgm;
    }
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const double v = sv[g];
      const double r1 = dpp_from_right<1>(v), r2 = dpp_from_right<2>(v), l1 = dpp_from_left<1>(v);
      const double d = (li == SCOL) ? (r1 - r2) : (l1 - v);   // Psi+ - Phi+ as lanes SCOL and SCOL + 2 see it
      double nv = 0.0;                                          // Psi_m = 0
      if (li == SCOL) nv = has_next ? __builtin_fma(isg * (eta - iota), d, v) : v;
      if (li == SCOL + 2) nv = has_next ? __builtin_fma(-isg * (xi - chi), d, l1) : l1;   // Phi_m = Psi - ...
      sv[g] = (li >= SCOL && li <= SCOL + 2) ? nv : 0.0;
      if (has_next && li == SCOL) pr[RL.off[RTOC_RIC_DTSDX] + 4 * g + q] = -isg * d;
    }
    sc[0] = 0.0, sc[1] = 0.0, sc[3] = 0.0;
    if (has_next) {
      if (lane0 == 0) {
        pr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTSDTS] = isg * (xi - chi);
        pr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0] = -isg * (eta - iota);
      }
      sc[2] = xi - isg * (xi - chi) * (xi - chi);
      sc[4] = eta - isg * (xi - chi) * (eta - iota);
    } else {
      sc[2] = xi;
      sc[4] = eta;
    }
  };
  rv_lds_sync();

  for (int st = hi; st >= lo; --st) {
    lane = lane0;
    asm volatile("" : "+v"(lane));   // opaque per stage: keeps LICM from pinning the per-lane addresses of every unrolled loop
    lane &= 63;
    li = lane & 15;
    q = lane >> 4;
    const int gword = __builtin_amdgcn_readfirstlane(sGrid[st]);
    const bool impact = (gword & 15) == RTOC_GRID_IMPACT;
    const bool sto = STO && ((gword >> 4) & 1), sto_next = STO && ((gword >> 5) & 1);
    const int ns = impact ? 0 : (gword >> 8);   // rtoc_grid::dims: rows of the switching constraint
    const double* kr = a.kkt + kinst + (size_t)st * KL.stride;
    double* rr = a.ric + rinst + (size_t)st * RL.stride;
    if constexpr (STO) {
      // ---- phase transition (riccati_factorizer.cpp:145-175; dispatch riccati_recursion.cpp:41-70) on the carried lanes: s+ at lane
      //      SCOL, Psi+ at SCOL + 1, Phi+ at SCOL + 2 of sv, the five scalars in sc ----
      const bool next_lift = (__builtin_amdgcn_readfirstlane(sGrid[st + 1]) & 15) == RTOC_GRID_LIFT;
      const bool prev_sto = st > 0 && ((__builtin_amdgcn_readfirstlane(sGrid[st > 0 ? st - 1 : 0]) >> 4) & 1);
      bool do_pt = false;
      int pol = st;
      if (impact) {
        do_pt = prev_sto || sto;
      } else if (next_lift) {
        do_pt = sto || sto_next;
        pol = st + 1;
      }
      if (do_pt) phase_transition(pol, sto_next);
      // without a next switching time Phi+ plays no part in a control grid point (brrf.cpp:61-65, 124-128, 139-141)
      if (!impact && !sto_next) {
#pragma unroll
        for (int g = 0; g < KG; ++g) sv[g] = (li == SCOL + 2) ? 0.0 : sv[g];
      }
    }
    RV_PROF(0);
    // everything this stage reads from LDS or was promised in registers has landed (DMA of A and the strip, Qxu^T, the grid
    // descriptor); the stores of the previous stage (K, k, s) have left too
    // ... except the store of s, the youngest vector-memory operation of the previous stage (issued after every load and after the K
    // stores): vmcnt counts in order, so "all but the last one" leaves exactly that store in flight instead of waiting out its latency
    if (st < hi)
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RV_PROF(1);
    if constexpr (SA) {
      // ---- BwdArgs::check_fxx: do the rows this form does not multiply have the shape it assumes?  Lane j < NX owns column j of the
      //      LDS copy of A and counts its non-zero entries in rows [NP, NV) -- at most the ONE the structure allows (a on the
      //      diagonal, c on the diagonal of the right half: read back and compared exactly) -- and in rows [0, NP) outside the corners ----
      if (a.check_fxx) {
        const int j = (lane < NX) ? lane : 0;
        const double* colp = sA + j * LDP;
        int cnt_s = 0, cnt_c = 0;
#pragma unroll
        for (int i = 0; i < NV; i += 2) {
          const d2 v = *reinterpret_cast<const d2*>(colp + i);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (i + e < NP_) cnt_c += (v[e] != 0.0) ? 1 : 0;
            else cnt_s += (v[e] != 0.0) ? 1 : 0;
          }
        }
        const bool on_a = j >= NP_ && j < NV, on_c = j >= NV + NP_;
        const double want = on_a ? sA[NP_ + NP_ * LDP] : (on_c ? sA[NP_ + (NV + NP_) * LDP] : 0.0);
        const double got = colp[on_a ? j : (on_c ? j - NV : NP_)];   // the one entry the column may have in these rows
        const bool corner_col = j < NP_ || (j >= NV && j < NV + NP_);
        const int allowed = ((on_a || on_c) && got != 0.0) ? 1 : 0;
        const bool bad = ((on_a || on_c) && got != want) || cnt_s != allowed || (!corner_col && cnt_c != 0);
        if (lane < NX && bad) stat |= RTOC_STAT_FXX_UNSTRUCTURED;
      }
    }

    // Qxu^T of this stage -> start values of the PB^T rows (used after the factorisation, like Qxx below; loaded here, not with the
    // DMA of the record: held across the policy / switching-constraint products its registers would be spilled)
    if (!RV_DBG(4)) issue_hq(st);
    // ---- Qxx of this stage -> accumulators of the F product (upper tiles; off-diagonal ones symmetrised,
    //      brrf.cpp:85 folded into the start value), used two products from here ----
    d4 f[T][T];
    if (RV_DBG(0)) {
#pragma unroll
      for (int c = 0; c < T; ++c)
#pragma unroll
        for (int t = c; t < T; ++t) f[c][t] = zero4();
    } else {
      const double* qb_ = kr + KL.off[RTOC_KKT_QXX];
      const unsigned lic = (li < SCOL) ? li : SCOL - 1;                  // last column tile: lanes beyond the matrix read a clamped column
      const unsigned vx = q + li * NX, vxl = q + lic * NX;               // Qxx[i][j]: i = .. + q, j = .. + li
      const unsigned vt = li + q * NX, vtl = lic + q * NX;               // Qxx[j][i]
#pragma unroll
      for (int c = 0; c < T; ++c)
#pragma unroll
        for (int t = c; t < T; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // unconditional loads (lanes beyond the matrix read a clamped address): the columns >= NX of the last column tile
            // are masked where column NX gets its start value below, rows >= NX are whole register groups
            if (16 * c + 4 * r >= NX) {
              f[c][t][r] = 0.0;
              continue;
            }
            // transposed element Qxx[j][i]: li runs along the contiguous index (16 lanes = one 128-byte line); the direct element
            // Qxx[i][j] puts four lanes on 32 bytes of sixteen different lines.  Diagonal tiles take the transposed one alone: their
            // upper triangle is mirrored at the end of the stage, so which triangle of a (to round-off) symmetric Qxx feeds it is free
            const double vt_ = rv_ld(qb_ + ((t < T - 1) ? vt : vtl) + (16 * t + (16 * c + 4 * r) * NX));
            if (t == c) {
              f[c][t][r] = vt_;
            } else {
              const double v = rv_ld(qb_ + ((t < T - 1) ? vx : vxl) + (16 * c + 4 * r + 16 * t * NX));
              f[c][t][r] = 0.5 * (v + vt_);
            }
          }
    }

    RV_PROF(2);
    // ================= PB = P+[:, v] Bv (controls shifted by SCOL lanes), G = Quu + Bv^T PB[v, :] ==================
    d4 acc[T];
    double bts[KSU];   // Bv^T s+_v, rider column 0 of the G product (valid on lanes li == 0)
#pragma unroll
    for (int c = 0; c < T; ++c) acc[c] = zero4();
#pragma unroll
    for (int ks = 0; ks < KSU; ++ks) bts[ks] = 0.0;
    if (!impact) {
#pragma unroll
      for (int g = G0; g < KG; ++g) {
        const int k = 4 * g + q - NV, u = li - SCOL;
        const bool ok = k >= 0 && u >= 0;
        const double v = sBv[ok ? k + u * NV : 0];
        const double bv = ok ? v : 0.0;
#pragma unroll
        for (int c = 0; c < T; ++c) acc[c] = mfma16(pp[g / 4][c][g % 4], bv, acc[c]);   // P+[16c + li][4g + q] = P+[4g + q][16c + li]
      }
      d4 gacc = zero4();
#pragma unroll
      for (int g = G0; g < KG; ++g) {
        const int k = 4 * g + q - NV;
        const bool ok = k >= 0 && li < NU;
        const double v = sBv[ok ? k + li * NV : 0];
        const double av = ok ? v : 0.0;                       // Bv^T[u' = li][k]
        const double srow = dpp_from_right<SCOL>(sv[g]);      // s+[4g + q], lane SCOL -> lane 0
        const double bv = (li < (STO ? 3 : 1)) ? srow : acc[g / 4][g % 4];   // PB[4g + q][u = li - SCOL]: the C layout is the B fragment
        gacc = mfma16(av, bv, gacc);
      }
#pragma unroll
      for (int r = 0; r < KSU; ++r) {
        const int u0 = q + 4 * r, u1 = li - SCOL;
        if (u0 < NU && u1 >= 0) sG[u0 + u1 * NU] += gacc[r];
        bts[r] = gacc[r];
      }
    }
    RV_PROF(3);
#if RTOC_RV_MERGE_PBT
    // PB^T takes the idle lanes of P+'s last column tile (li >= SCOL: columns beyond the matrix, zero): the row tile T - 1 of the
    // stacked operand [P+; PB^T] is then one register file, no select per product below -- and the accumulators of PB are dead here
#pragma unroll
    for (int g = 0; g < KG; ++g) pp[g / 4][T - 1][g % 4] = (li >= SCOL) ? acc[g / 4][g % 4] : pp[g / 4][T - 1][g % 4];
#endif
    RV_PROF(4);
    if (!impact) {
      rv_lds_sync();
      // LLT(G) (riccati_factorizer.cpp:49) and Y = L^-1 in the same instruction stream; L itself is not used again
      if (wave_llt_inv<NU, NU>(sG, scr, smem + OFF_LINV, sY, NU, lane)) stat |= RTOC_STAT_QUU_NOT_SPD;
      rv_lds_sync();
    }

    RV_PROF(5);
    // ================= column tile by column tile: W[:, t] = [P+; PB^T] [A | Fx][:, t], F[c][t] += A^T[c] W[:, t] ===========
    double xd[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // STO: dot products over the state (set in the last column tile)
    double hT[T][KSU];   // H^T = Qxu^T + PB^T A (rows NX.. of W), B fragments of the policy product
    double lup[KSU];     // lu' = lu - Bv^T s+_v + PB^T Fx on lanes li == SCOL
#pragma unroll
    for (int ks = 0; ks < KSU; ++ks) lup[ks] = 0.0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      d4 w[T];
#pragma unroll
      for (int tm = 0; tm < T; ++tm) w[tm] = zero4();
#pragma unroll
      for (int ks = 0; ks < KSU; ++ks) w[T - 1][RS + ks] = (impact || (t == T - 1 && li >= SCOL)) ? 0.0 : hq[t][ks];
      // B fragment of k group g: A[4g + q][16t + li]; last tile: columns >= NX are Fx (lane SCOL) and nothing
      const double* pb_ = (t < T - 1 || li < SCOL) ? (sA + q + (16 * t + li) * LDP) : (smem + ((STO && li == SCOL + 1) ? ST_FFX : ST_FX) + q);
      const bool okb = (t < T - 1) || li <= SCOL || (STO && sto && !impact && li == SCOL + 1);   // STO: column NX + 1 is fx
      // structured form: does k group g hold a dense row of A (a corner row or a velocity row)?  is row 4g + q one?
      auto group_dense = [](int g) { return 4 * g < NP_ || 4 * g + 3 >= NV; };
      auto row_dense = [&](int g) { return 4 * g + q < NP_ || 4 * g + q >= NV; };
      const bool sa_tile = SA && t < T - 1;   // (the last column tile carries the riders and stays dense)
      double braw[2];
      braw[0] = pb_[0];
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        if (g + 1 < KG) braw[(g + 1) & 1] = pb_[4 * (g + 1)];
        __builtin_amdgcn_sched_barrier(0);
        const double bv = okb ? braw[g & 1] : 0.0;
        const double bm = (!sa_tile || row_dense(g)) ? bv : 0.0;
#pragma unroll
        for (int tm = 0; tm < T; ++tm) {
          double av = pp[g / 4][tm][g % 4];
#if !RTOC_RV_MERGE_PBT
          if (tm == T - 1) av = (li >= SCOL) ? acc[g / 4][g % 4] : av;   // PB^T[u = li - SCOL][4g + q]
#endif
          if (sa_tile && tm < T - 1) {
            if (group_dense(g)) w[tm] = mfma16(av, bm, w[tm]);
          } else {
            w[tm] = mfma16(av, bv, w[tm]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (sa_tile) {
        // the structured rows' part of the P+ row tiles: column j = 16t + li takes a S[:, j] (j in [NP, NV): same tile, lane, register
        // of P+) or c S[:, j - NV] (j in [NV + NP, NX): NV = 2 mod 16, i.e. the previous tile two lanes to the left)
        const double ca = sA[NP_ + NP_ * LDP], cc = sA[NP_ + (NV + NP_) * LDP];
        const int j = 16 * t + li;
        const double ca_l = (j >= NP_ && j < NV) ? ca : 0.0;
        const double cc_l = (j >= NV + NP_ && li >= NV % 16) ? cc : 0.0;
#pragma unroll
        for (int tm = 0; tm < T - 1; ++tm)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            w[tm][r] = __builtin_fma(ca_l, pp[tm][t][r], w[tm][r]);
            if (t >= 1 && 16 * t + 15 >= NV + NP_) w[tm][r] = __builtin_fma(cc_l, dpp_from_left<NV % 16>(pp[tm][t - 1][r]), w[tm][r]);
          }
      }
      if (t == T - 1) {
        // lane masks as numbers: selects on just-loaded values become branches around the loads (with a full vmcnt wait at the join)
        const double m_eq = (li == SCOL) ? 1.0 : 0.0, m_lt = (li < SCOL) ? 1.0 : 0.0, sgn_col = (li == SCOL) ? -1.0 : 1.0;
        // column NX: rows < NX hold P+ Fx -> z = s+ - P+ Fx (brrf.cpp:86); rows NX.. hold PB^T Fx -> lu'
#pragma unroll
        for (int g = 0; g < KG; ++g) w[g / 4][g % 4] = __builtin_fma(sgn_col, w[g / 4][g % 4], sv[g]);   // sv is zero off lane SCOL
        if (!impact) {
#pragma unroll
          for (int ks = 0; ks < KSU; ++ks) {
            if constexpr (!STO) {
              const double luv = smem[ST_LU + 4 * ks + q];
              lup[ks] = luv - dpp_from_left<SCOL>(bts[ks]) + w[T - 1][RS + ks];
            } else {
              // lane SCOL: lu' = lu - Bv^T s+_v + PB^T Fx; SCOL + 1: psi_u = hu + Bv^T Psi+_v + PB^T fx; SCOL + 2: phi_u = Bv^T Phi+_v (brrf.cpp:52-60)
              const double seed = smem[((li == SCOL + 1) ? ST_HU : ST_LU) + 4 * ks + q];
              const double sgn = (li == SCOL) ? -1.0 : 1.0;
              const double v = ((li == SCOL + 2) ? 0.0 : seed) + sgn * dpp_from_left<SCOL>(bts[ks]) + w[T - 1][RS + ks];
              lup[ks] = (li >= SCOL && li <= SCOL + 2) ? v : 0.0;
            }
          }
        }
        // ... and column NX of F starts from -lx, so that it ends as w = A^T z - lx (brrf.cpp:87-88)
#pragma unroll
        for (int g = 0; g < KG; ++g) {
          if constexpr (!STO) {
            const double lxv = smem[ST_LX + 4 * g + q];
            f[g / 4][T - 1][g % 4] = __builtin_fma(-m_eq, lxv, m_lt * f[g / 4][T - 1][g % 4]);
          } else {   // column NX + 1 starts from hx (psi_x = A^T y + hx, brrf.cpp:52-55), NX + 2 from zero
            const double xv = smem[((li == SCOL + 1) ? ST_HX : ST_LX) + 4 * g + q];
            const double cf = (li == SCOL) ? -1.0 : ((li == SCOL + 1 && sto && !impact) ? 1.0 : 0.0);
            f[g / 4][T - 1][g % 4] = __builtin_fma(cf, xv, m_lt * f[g / 4][T - 1][g % 4]);
          }
        }
        if constexpr (STO) {
          // ---- the six dot products over the state the scalars need (brrf.cpp:110-142): on the rider lanes W now holds z (SCOL),
          //      y = P+ fx + Psi+ (SCOL + 1), Phi+ (SCOL + 2), sv holds s+, Psi+, Phi+ ----
          double da = 0.0, db = 0.0, dc = 0.0;
#pragma unroll
          for (int g = 0; g < KG; ++g) {
            const double fxv = smem[ST_FFX + 4 * g + q], Fxv = smem[ST_FX + 4 * g + q];
            da = __builtin_fma(fxv, w[g / 4][g % 4], da);
            db = __builtin_fma(fxv, sv[g], db);
            dc = __builtin_fma(Fxv, sv[g], dc);
          }
          da += __shfl_xor(da, 16, 64), db += __shfl_xor(db, 16, 64), dc += __shfl_xor(dc, 16, 64);
          da += __shfl_xor(da, 32, 64), db += __shfl_xor(db, 32, 64), dc += __shfl_xor(dc, 32, 64);
          xd[0] = readlane_d(da, SCOL);                                 // fz   = fx . z
          xd[1] = readlane_d(da, SCOL + 1) - readlane_d(db, SCOL + 1);  // fPf  = fx . P+ fx
          xd[2] = readlane_d(db, SCOL + 1);                             // psif = Psi+ . fx
          xd[3] = readlane_d(dc, SCOL + 1);                             // psiF = Psi+ . Fx
          xd[4] = readlane_d(db, SCOL + 2);                             // phif = Phi+ . fx
          xd[5] = readlane_d(dc, SCOL + 2);                             // phiF = Phi+ . Fx
        }
      }
      RV_PROF(6 + 2 * t);
      // ---- P of grid point st + 1 -> HBM, a third per column tile, from the registers that hold it as P+ until the last of these
      //      products (element (i, j) is written through its mirror (j, i): li runs along the contiguous index).  Spread over the
      //      stage: a wave has 6 bits of vmcnt, and 36 Qxx loads + 28 stores in one burst stall the issue at the 64th ----
      if (st < hi && !RV_DBG(1)) {
        double* pw = a.ric + rinst + (size_t)(st + 1) * RL.stride + RL.off[RTOC_RIC_P];
        const unsigned vst = li + q * NX;   // per-lane part of the address, 32 bits; the rest is compile-time
#pragma unroll
        for (int mt = 0; mt < T; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * t + 4 * r + q, j = 16 * mt + li;
            if (i < NX && j < NX) rv_st(pw + (vst + (16 * mt + (16 * t + 4 * r) * NX)), pp[t][mt][r]);
          }
      }
#pragma unroll
      for (int ks = 0; ks < KSU; ++ks) hT[t][ks] = w[T - 1][RS + ks];
      // F[c][t] += A^T[c] W[:, t]: A fragment A[k = 4g + q][m = 16c + li] (the same LDS words as above), B fragment W in its C layout
      double araw[2][T];
      auto load_a = [&](int g, double (&d)[T]) {
#pragma unroll
        for (int c = 0; c <= t; ++c) d[c] = sA[q + 4 * g + (16 * c + li) * LDP];
      };
      if constexpr (!SA) {
      load_a(0, araw[0]);
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        if (g + 1 < KG) load_a(g + 1, araw[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const double bw = w[g / 4][g % 4];
#pragma unroll
        for (int c = 0; c <= t; ++c) {
          const double av = (c < T - 1 || li < SCOL) ? araw[g & 1][c] : 0.0;
          f[c][t] = mfma16(av, bw, f[c][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      } else {
        // structured form: only the k groups with dense rows (the others: scaled rows of W below), corner rows only into the
        // column tiles of A that hold corner columns (tiles 0 and 1)
        constexpr int GFIRST = 0;
        static_assert(NP_ >= 1, "group 0 holds corner rows");
        auto next_dense = [&](int g) { int n = g + 1; while (n < KG && !group_dense(n)) ++n; return n; };
        load_a(GFIRST, araw[0]);
        int par = 0;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
          if (!group_dense(g)) continue;
          const int gn = next_dense(g);
          if (gn < KG) load_a(gn, araw[par ^ 1]);
          __builtin_amdgcn_sched_barrier(0);
          const double bw = w[g / 4][g % 4];
          const bool corner_only = 4 * g + 3 < NV;   // a dense group below the velocity rows: corner rows (and masked structured ones)
#pragma unroll
          for (int c = 0; c <= t; ++c) {
            if (corner_only && c == T - 1) continue;
            double av = (c < T - 1 || li < SCOL) ? araw[par][c] : 0.0;
            av = row_dense(g) ? av : 0.0;
            f[c][t] = mfma16(av, bw, f[c][t]);
          }
          __builtin_amdgcn_sched_barrier(0);
          par ^= 1;
        }
        // rows [NP, NV) of A: F[i][:] += a W[i][:]; rows i in [NV + NP, NX): F[i][:] += c W[i - NV][:], and i - NV = 4 (g' - 5) + (q + 2)
        // for q < 2, 4 (g' - 4) + (q - 2) for q >= 2: the value sits two q-groups away (lane ^ 32) in group g' - 5 or g' - 4
        const double ca = sA[NP_ + NP_ * LDP], cc = sA[NP_ + (NV + NP_) * LDP];
#pragma unroll
        for (int c = 0; c <= t; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int gp = 4 * c + r;
            if (4 * gp + 3 >= NP_ && 4 * gp < NV) {
              const double coef = (4 * gp + q >= NP_ && 4 * gp + q < NV) ? ca : 0.0;
              f[c][t][r] = __builtin_fma(coef, w[c][r], f[c][t][r]);
            }
            if (4 * gp + 3 >= NV + NP_ && 4 * gp < NX) {
              constexpr int DHI = (NV - 2) / 4, DLO = (NV + 2) / 4;
              const int ghi = gp - DHI, glo = gp - DLO;
              const double hi_v = (ghi >= 0) ? w[(ghi >= 0 ? ghi : 0) / 4][(ghi >= 0 ? ghi : 0) % 4] : 0.0;
              const double lo_v = (glo >= 0) ? w[(glo >= 0 ? glo : 0) / 4][(glo >= 0 ? glo : 0) % 4] : 0.0;
              const double send = (q < 2) ? hi_v : lo_v;
              const double got = __shfl_xor(send, 32, 64);
              const double coef = (4 * gp + q >= NV + NP_) ? cc : 0.0;
              f[c][t][r] = __builtin_fma(coef, got, f[c][t][r]);
            }
          }
      }
      RV_PROF(7 + 2 * t);
    }

    // ---- A, Bv, Quu and the vectors of this stage have been read for the last time: the record of the next grid point ----
    double ksc[3] = {0.0, 0.0, 0.0};   // STO: Qtt, Qtt_prev, h of this grid point (the strip is about to be overwritten)
    double scn[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // STO: xi, chi, rho, eta, iota of THIS grid point
    if constexpr (STO) {
      if (sto && !impact) {
        ksc[0] = smem[ST_SCAL + RTOC_KKT_SCAL_QTT], ksc[1] = smem[ST_SCAL + RTOC_KKT_SCAL_QTT_PREV], ksc[2] = smem[ST_SCAL + RTOC_KKT_SCAL_H];
        // psi_x = A^T y + hx, phi_x = A^T Phi+ (brrf.cpp:52-65): columns NX + 1, NX + 2 of F before the policy terms
#pragma unroll
        for (int g = 0; g < KG; ++g)
          if (li == SCOL + 1 || li == SCOL + 2) rr[((li == SCOL + 1) ? RL.off[RTOC_RIC_PSIX] : RL.off[RTOC_RIC_PHIX]) + 4 * g + q] = f[g / 4][T - 1][g % 4];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (STO form: on a switching-constraint grid point S, Ls, Ws take the place of A -- the DMA waits for that block)
    const bool dma_late = STO && NS > 0 && ns > 0;
    if (st > lo && !dma_late) {
      if (!RV_DBG(2)) issue_dma(st - 1);
    }
    asm volatile("" ::: "memory");

    RV_PROF(12);
    // ================= policy: Z^T = Y [H^T | -lu'],  [K | -k] = -Y^T Z^T  (riccati_factorizer.cpp:55-56) ============
    if (!impact) {
      d4 zt[T], kk[T];
      double y1[KSU], y2[KSU];
      double scm[3] = {0.0, 0.0, 0.0};   // STO: m . Phit, mt . Phit, mt_next . Phit
#pragma unroll
      for (int ks = 0; ks < KSU; ++ks) {
        const bool ok = li < NU;
        const double v1 = sY[(ok ? li : 0) + (4 * ks + q) * NU];   // A[m = i = li][k = u]: Y[i][u]
        const double v2 = sY[(4 * ks + q) + (ok ? li : 0) * NU];   // A[m = u = li][k = i]: Y[i][u]
        y1[ks] = ok ? v1 : 0.0;
        y2[ks] = ok ? -v2 : 0.0;
      }
#pragma unroll
      for (int c = 0; c < T; ++c) {
        zt[c] = zero4();
        kk[c] = zero4();
      }
#pragma unroll
      for (int ks = 0; ks < KSU; ++ks)
#pragma unroll
        for (int c = 0; c < T; ++c) {
          double bv = hT[c][ks];
          if (c == T - 1) bv = (li < SCOL) ? bv : ((li == SCOL) ? -lup[ks] : ((STO && li <= SCOL + 2) ? lup[ks] : 0.0));   // STO: +psi_u, +phi_u
          zt[c] = mfma16(y1[ks], bv, zt[c]);
        }
      d4 ys[T];
#pragma unroll
      for (int c = 0; c < T; ++c) ys[c] = zt[c];
      if constexpr (NS > 0) {
        if (ns > 0) {
          // ---- switching constraint (riccati_factorizer.cpp:58-89) in factorised form.  With Zh = Z^T above (rider column: -L^-1 lu'),
          //      Zd = Y Phiu^T, S = Phiu G^-1 Phiu^T = Zd^T Zd = Ls Ls^T, Ws = Ls^-1:
          //        Eh = Ws (Zd^T Zh - [Phix | -Pres]),   M = -Ws^T Eh  (rider column: -m),   Ys = Zh + Zd M,   K = -Y^T Ys,
          //        K^T G K + K^T Phiu^T M + M^T Phiu K = Zh^T Zh - Eh^T Eh,
          //      so the update of F (and, in its column NX, s -= Phix^T m + H k) is F -= Zh^T Zh, F += Eh^T Eh: every product runs
          //      from accumulators (tools/rv_model.py checks the identities against the oracle); LDS: S, Ls, Ws in the transpose scratch ----
          constexpr int NSK = (NS + 3) / 4;
          static_assert(NS <= 16 && (STO || 3 * C::pad8(NS * NS) + C::pad8(NS) <= 2 * SCR_TILE), "S, Ls, Ws and 1/diag(Ls) in the transpose scratch");
          static_assert(3 * C::pad8(NS * NS) + C::pad8(NS) <= NX * LDP, "... or in the place of A");
          double* const scb = STO ? sA : scr;   // (STO form: one scratch tile only; A is dead and the next record's DMA is held back)
          double* const sS = scb;
          double* const sLs = scb + C::pad8(NS * NS);
          double* const sWs = scb + 2 * C::pad8(NS * NS);
          double* const sLsInv = scb + 3 * C::pad8(NS * NS);
          d4 t1[T];
          {
            const double* px_ = kr + KL.off[RTOC_KKT_PHIX];
            const double* pr_ = kr + KL.off[RTOC_KKT_PRES];
#pragma unroll
            for (int c = 0; c < T; ++c)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int l = q + 4 * r, x = 16 * c + li;
                if (4 * r >= NS) {
                  t1[c][r] = 0.0;
                  continue;
                }
                const bool okl = l < ns;
                const bool isx = (c < T - 1) || li < SCOL;
                const double vx = px_[(okl ? l : 0) + (isx ? x : 0) * NS];
                double v = (okl && isx) ? -vx : 0.0;
                if (c == T - 1) {
                  const double vp = pr_[okl ? l : 0];
                  v = (okl && li == SCOL) ? vp : v;
                  if constexpr (STO) {   // column NX + 1: -Phit, so that M's column NX + 1 is mt (riccati_factorizer.cpp:117-118)
                    const double vt = kr[KL.off[RTOC_KKT_PHIT] + (okl ? l : 0)];
                    v = (okl && sto && li == SCOL + 1) ? -vt : v;
                  }
                }
                t1[c][r] = v;
              }
          }
          double pu[KSU];
#pragma unroll
          for (int ks = 0; ks < KSU; ++ks) {
            const bool ok = li < ns;
            const double v = kr[KL.off[RTOC_KKT_PHIU] + (ok ? li : 0) + (4 * ks + q) * NS];   // Phiu[l = li][u = 4 ks + q]
            pu[ks] = ok ? v : 0.0;
          }
          d4 zd = zero4(), zdt = zero4(), sacc = zero4();
#pragma unroll
          for (int ks = 0; ks < KSU; ++ks) {
            zd = mfma16(y1[ks], pu[ks], zd);     // Zd = Y Phiu^T            (rows u, columns l)
            zdt = mfma16(pu[ks], y1[ks], zdt);   // Zd^T = Phiu Y^T          (rows l, columns u)
          }
#pragma unroll
          for (int ks = 0; ks < KSU; ++ks) sacc = mfma16(zd[ks], zd[ks], sacc);   // S = Zd^T Zd
#pragma unroll
          for (int r = 0; r < NSK; ++r) {
            const int i = q + 4 * r;
            if (i < NS && li < NS) sS[i + li * NS] = sacc[r];
          }
#pragma unroll
          for (int ks = 0; ks < KSU; ++ks)
#pragma unroll
            for (int c = 0; c < T; ++c) t1[c] = mfma16(zd[ks], zt[c][ks], t1[c]);   // Zd^T Zh - [Phix | -Pres]
          rv_lds_sync();
          if (wave_llt_inv<NS, NS>(sS, sLs, sLsInv, sWs, ns, lane)) stat |= RTOC_STAT_S_NOT_SPD;
          rv_lds_sync();
          d4 eh[T], mm[T];
#pragma unroll
          for (int c = 0; c < T; ++c) {
            eh[c] = zero4();
            mm[c] = zero4();
          }
#pragma unroll
          for (int ks = 0; ks < NSK; ++ks) {
            const bool ok = li < NS && 4 * ks + q < NS;
            const double v = sWs[(ok ? li : 0) + (ok ? 4 * ks + q : 0) * NS];      // Ws[l = li][l' = 4 ks + q]
            const double av = ok ? v : 0.0;
#pragma unroll
            for (int c = 0; c < T; ++c) eh[c] = mfma16(av, t1[c][ks], eh[c]);
          }
#pragma unroll
          for (int ks = 0; ks < NSK; ++ks) {
            const bool ok = li < NS && 4 * ks + q < NS;
            const double v = sWs[(ok ? 4 * ks + q : 0) + (ok ? li : 0) * NS];      // Ws[l' = 4 ks + q][l = li]
            const double av = ok ? -v : 0.0;
#pragma unroll
            for (int c = 0; c < T; ++c) mm[c] = mfma16(av, eh[c][ks], mm[c]);       // M = -Ws^T Eh
          }
#pragma unroll
          for (int ks = 0; ks < NSK; ++ks)
#pragma unroll
            for (int c = 0; c < T; ++c) {
              ys[c] = mfma16(zdt[ks], mm[c][ks], ys[c]);                             // Ys = Zh + Zd M
#pragma unroll
              for (int t = c; t < T; ++t) f[c][t] = mfma16(eh[c][ks], eh[t][ks], f[c][t]);   // F += Eh^T Eh
            }
          // M[l = q + 4r][x = 16c + li] -> HBM (ns x NX, ld NS); column NX is -m
          double chkm = 0.0;
#pragma unroll
          for (int c = 0; c < T; ++c)
#pragma unroll
            for (int r = 0; r < NSK; ++r) {
              const int l = q + 4 * r, x = 16 * c + li;
              if (l < ns && x < NX) rr[RL.off[RTOC_RIC_M] + l + x * NS] = mm[c][r];
              if (c == T - 1 && l < ns && li == SCOL) rr[RL.off[RTOC_RIC_MV] + l] = -mm[c][r];
              if constexpr (STO) {   // columns NX + 1, NX + 2: mt, mt_next (:117-124)
                if (c == T - 1 && l < ns && sto && (li == SCOL + 1 || li == SCOL + 2))
                  rr[((li == SCOL + 1) ? RL.off[RTOC_RIC_MT] : RL.off[RTOC_RIC_MTN]) + l] = mm[c][r];
              }
              chkm = __builtin_fma(mm[c][r], 0.0, chkm);
            }
          if (is_bad(chkm)) stat |= RTOC_STAT_NAN;
          if constexpr (STO) {   // -m . Phit, mt . Phit, mt_next . Phit on lanes SCOL, SCOL + 1, SCOL + 2 (riccati_factorizer.cpp:137-141)
            double dm = 0.0;
#pragma unroll
            for (int r = 0; r < NSK; ++r) {
              const int l = q + 4 * r;
              const double pt = kr[KL.off[RTOC_KKT_PHIT] + ((l < ns) ? l : 0)];
              dm = __builtin_fma(mm[T - 1][r], (l < ns) ? pt : 0.0, dm);
            }
            dm += __shfl_xor(dm, 16, 64);
            dm += __shfl_xor(dm, 32, 64);
            scm[0] = -readlane_d(dm, SCOL), scm[1] = readlane_d(dm, SCOL + 1), scm[2] = readlane_d(dm, SCOL + 2);
          }
        }
      }
      if (dma_late && st > lo) {   // (the switching constraint's scratch in the place of A has been read for the last time)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_dma(st - 1);
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int ks = 0; ks < KSU; ++ks)
#pragma unroll
        for (int c = 0; c < T; ++c) kk[c] = mfma16(y2[ks], ys[c][ks], kk[c]);
      // K[u = q + 4r][x = 16c + li] -> HBM (K row-major, lqr_policy.hpp:18-19); column NX is -k
      double chk = 0.0;
#pragma unroll
      for (int c = 0; c < T; ++c)
#pragma unroll
        for (int r = 0; r < KSU; ++r) {
          const int u = q + 4 * r, x = 16 * c + li;
          if (x < NX && !RV_DBG(3)) rv_st(rr + RL.off[RTOC_RIC_K] + u * NX + x, kk[c][r]);
          if (c == T - 1 && li == SCOL) rr[RL.off[RTOC_RIC_KV] + u] = -kk[c][r];
          if constexpr (STO) {   // columns NX + 1, NX + 2 of K: T, W (riccati_factorizer.cpp:109-127); psi_u, phi_u beside them
            if (c == T - 1 && sto && (li == SCOL + 1 || li == SCOL + 2)) {
              rr[((li == SCOL + 1) ? RL.off[RTOC_RIC_T] : RL.off[RTOC_RIC_W]) + u] = kk[c][r];
              rr[((li == SCOL + 1) ? RL.off[RTOC_RIC_PSIU] : RL.off[RTOC_RIC_PHIU]) + u] = lup[r];
            }
          }
          chk = __builtin_fma((x <= NX + ((STO && sto) ? 2 : 0)) ? kk[c][r] : 0.0, 0.0, chk);
        }
      if (is_bad(chk)) stat |= RTOC_STAT_NAN;
      if constexpr (STO) {
        if (sto) {
          // ---- the five dot products over the controls and the scalars xi, chi, rho, eta, iota (brrf.cpp:110-142).  Rows u = q + 4r of
          //      the last tile of K: -k at lane SCOL, T at SCOL + 1, W at SCOL + 2; lup: lu', psi_u, phi_u at the same lanes ----
          double d0 = 0.0, d1 = 0.0, d2_ = 0.0, d3 = 0.0;
#pragma unroll
          for (int r = 0; r < KSU; ++r) {
            const double kv = kk[T - 1][r], lv = lup[r];
            d0 = __builtin_fma(kv, lv, d0);                      // SCOL + 1: T . psi_u     SCOL + 2: W . phi_u
            d1 = __builtin_fma(kv, dpp_from_right<1>(lv), d1);   // SCOL + 1: T . phi_u
            d2_ = __builtin_fma(lv, dpp_from_left<1>(kv), d2_);  // SCOL + 1: psi_u . (-k)
            d3 = __builtin_fma(lv, dpp_from_left<2>(kv), d3);    // SCOL + 2: phi_u . (-k)
          }
          d0 += __shfl_xor(d0, 16, 64), d1 += __shfl_xor(d1, 16, 64), d2_ += __shfl_xor(d2_, 16, 64), d3 += __shfl_xor(d3, 16, 64);
          d0 += __shfl_xor(d0, 32, 64), d1 += __shfl_xor(d1, 32, 64), d2_ += __shfl_xor(d2_, 32, 64), d3 += __shfl_xor(d3, 32, 64);
          const double Tpsi = readlane_d(d0, SCOL + 1), Wphi = readlane_d(d0, SCOL + 2), Tphi = readlane_d(d1, SCOL + 1);
          const double psik = -readlane_d(d2_, SCOL + 1), phik = -readlane_d(d3, SCOL + 2);
          const double fz = xd[0], fPf = xd[1], psif = xd[2], psiF = xd[3], phif = xd[4], phiF = xd[5];
          scn[0] = fPf + ksc[0] + 2.0 * psif + Tpsi + sc[0] + scm[1];
          scn[3] = -fz + ksc[2] + psiF + psik + sc[3] + scm[0];
          if (sto_next) {
            scn[1] = ksc[1] + phif + Tphi + sc[1] + scm[2];
            scn[2] = Wphi + sc[2];
            scn[4] = phiF + phik + sc[4];
          }
        }
      }
      RV_PROF(13);
      // F -= K^T G K = Z Z^T (brrf.cpp:82-84); column NX: + H G^-1 lu' = - H k
#pragma unroll
      for (int ks = 0; ks < KSU; ++ks)
#pragma unroll
        for (int c = 0; c < T; ++c)
#pragma unroll
          for (int t = c; t < T; ++t) f[c][t] = mfma16(-zt[c][ks], zt[t][ks], f[c][t]);
    }

    RV_PROF(14);
    // ================= s <- column NX of F;  P <- sym(F) as nine operand tiles ==================
#pragma unroll
    for (int g = 0; g < KG; ++g) sv[g] = (li == SCOL) ? f[g / 4][T - 1][g % 4] : 0.0;
    if constexpr (STO) {
      // columns NX + 1, NX + 2 of F are Psi, Phi of this grid point (brrf.cpp:98-107); on an impact grid point Psi = 0, Phi = A^T Phi+ and
      // rho, iota pass through (:160-174); without switching-time optimisation everything is zero (riccati_factorizer.cpp:99-105)
      const bool keep_psi = sto && !impact, keep_phi = sto && (impact || sto_next);
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const double v = f[g / 4][T - 1][g % 4];
        sv[g] = (li == SCOL + 1) ? (keep_psi ? v : 0.0) : ((li == SCOL + 2) ? (keep_phi ? v : 0.0) : sv[g]);
      }
      if (sto && impact) {
        scn[2] = sc[2];
        scn[4] = sc[4] + xd[5];
      }
#pragma unroll
      for (int e = 0; e < 5; ++e) sc[e] = scn[e];
    }
    {
      // mask of tile (c, t): rows 16c + 4r + q < NX is a property of the register (NX % 4 == 0), columns 16t + li < NX of the lane
      auto masked = [&](int c, int t, int r, double v) -> double {
        if (16 * c + 4 * r >= NX) return 0.0;
        if (t == T - 1) return (li < SCOL) ? v : 0.0;
        return v;
      };
      // transposes in pairs through the two scratch tiles: (row q + 4r, column li) in, (row li, column q + 4r) out
      constexpr int NTR = T * (T + 1) / 2;
#pragma unroll
      for (int n0 = 0; n0 < NTR; n0 += NSCR) {
#pragma unroll
        for (int n = n0; n < n0 + NSCR && n < NTR; ++n) {
          const int c = rv_tile_row(T, n), t = rv_tile_col(T, n);
          double* s_ = scr + (n - n0) * SCR_TILE;
#pragma unroll
          for (int r = 0; r < 4; ++r) s_[(q + 4 * r) * SCR_LD + li] = masked(c, t, r, f[c][t][r]);
        }
        rv_lds_sync();
#pragma unroll
        for (int n = n0; n < n0 + NSCR && n < NTR; ++n) {
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
    RV_PROF(15);
    // ---- s and the (zero) switching-time fields Psi, Phi of the record -> HBM as ONE 16-byte-per-lane store: s is gathered from
    //      the four lanes that hold it through the transpose scratch; the three fields lie behind one another in the record.
    //      It is the LAST vector-memory operation of the stage -- the wait at the next stage top counts on that ----
    {
      static_assert(RL.off[RTOC_RIC_PSI] == RL.off[RTOC_RIC_S] + C::pad8(NX) && RL.off[RTOC_RIC_PHI] == RL.off[RTOC_RIC_PSI] + C::pad8(NX) &&
                        RL.off[RTOC_RIC_S] % 2 == 0 && NX % 2 == 0 && 2 * C::pad8(NX) + NX <= 128, "s | Psi | Phi: one strip of <= 64 chunks");
      double zero = 0.0;
      asm volatile("" : "+v"(zero));   // (materialised here: hoisted out of the loop the constant gets spilled and RELOADED -- a scratch load and a full wait)
      d2 out;
      if constexpr (!STO) {
        if (li == SCOL) {
#pragma unroll
          for (int g = 0; g < KG; ++g) scr[4 * g + q] = sv[g];
        }
        if (lane < 5) rr[RL.off[RTOC_RIC_SCAL] + lane] = zero;
        rv_lds_sync();
        const d2 sv2 = *reinterpret_cast<const d2*>(scr + ((2 * lane < NX) ? 2 * lane : 0));
        out.x = (2 * lane < NX) ? sv2.x : zero;
        out.y = (2 * lane < NX) ? sv2.y : zero;
      } else {
        // s, Psi, Phi from the lanes SCOL, SCOL + 1, SCOL + 2 into the scratch as they lie in the record (each field padded to 8)
        if (li >= SCOL && li <= SCOL + 2) {
#pragma unroll
          for (int g = 0; g < KG; ++g) scr[(li - SCOL) * C::pad8(NX) + 4 * g + q] = sv[g];
        }
        if (lane < 5) {
          const double v = (lane == 0) ? sc[0] : ((lane == 1) ? sc[1] : ((lane == 2) ? sc[2] : ((lane == 3) ? sc[3] : sc[4])));
          rr[RL.off[RTOC_RIC_SCAL] + lane] = v;
        }
        rv_lds_sync();
        const bool in = 2 * lane < 2 * C::pad8(NX) + NX && (2 * lane) % C::pad8(NX) < NX;   // (NX even: a pair never straddles a field's padding)
        const d2 sv2 = *reinterpret_cast<const d2*>(scr + (in ? 2 * lane : 0));
        out.x = in ? sv2.x : zero;
        out.y = in ? sv2.y : zero;
      }
      asm volatile("" ::: "memory");
      if (2 * lane < 2 * C::pad8(NX) + NX) *reinterpret_cast<d2*>(rr + RL.off[RTOC_RIC_S] + 2 * lane) = out;
    }
    asm volatile("" ::: "memory");
    RV_PROF(16);
  }

  if constexpr (STO) {
    // ---- grid[0].sto: the trailing phase transition writes sto_policy_[0] (riccati_recursion.cpp:75-79) ----
    const int g0 = __builtin_amdgcn_readfirstlane(sGrid[0]);
    if (lo == 0 && ((g0 >> 4) & 1) && ((g0 >> 5) & 1)) phase_transition(0, true);
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

// the carve plus the grid-kind table fill the 20 KB an eighth of a CU's LDS offers
template <int NV, int NU, int NS>
constexpr int rv_lds_bytes() {
  return 20 * 1024;
}

}  // namespace rtoc

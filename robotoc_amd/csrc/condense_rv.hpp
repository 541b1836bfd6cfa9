// condense_rv.hpp -- ContactDynamics::condenseContactDynamics (reference src/dynamics/contact_dynamics.cpp:55-164, with the evalKKT
// tail scalings of src/ocp/intermediate_stage.cpp:140-148 and Constraints::condenseSlackAndDual of the joint-limit rows,
// src/constraints/constraints.cpp:322-357) of ONE (instance, grid point) on ONE wavefront, every product chained through the
// register layouts of v_mfma_f64_16x16x4_f64.
//
// The role-split condensation (condense.hpp) keeps every matrix in LDS and pays ~5.7k VALU / 2k DS instructions per work item for
// operand fetches, address arithmetic and epilogues around 350 MFMAs: it is issue-bound.  Here the saddle inverse Lam = MJtJinv
// (assembled in LDS by the fragment all condensation kernels share) is read ONCE into the accumulator layout -- for a symmetric
// matrix that layout is at the same time its A-operand and its B-operand fragment -- and everything downstream stays in registers:
//
//   LD  = Lam [D | IDC]                      B operand: D straight from HBM in the B layout (rider column NX: MJtJinv_IDC)
//   Xn  = -Qafqv = W LD + E'                 rows < 16: a row scaling by Qaa; rows 16..31: one 16 x 16 block W1 = diag(Qaa[16..], Qff)
//   WL  = W Lam  (Qafu_full)                 likewise; rider columns NV, NV + 1: laf, haf
//   V   = Xn^T LD + LD^T E'                  = (Qxx update)^T: lanes run along the ROWS of Qxx -> 128-byte runs in HBM;
//                                              rider rows NX, NX + 1: lx, hx; V[NX + 1][NX]: the scalar h
//   V2  = -(Qxu_full)^T = WL^T LD + Lam_a E'  rows < np: Qxu_passive
//   QU  = Lam_a WL                           Quu / Quu_passive_topRight through the mirror index; rider columns: lu, lu_passive, hu
//   Phix -= Phia LD_a, Phiu = Phia Lam_a,u   (grid points with a switching constraint; rider column: Phia MJtJinv_IDC)
//
// A^T-operands and B-operands of all of these are accumulator registers of an earlier product: no LDS round trip, no address
// arithmetic.  tools/cond_model.py states the lane algebra in numpy and checks it against the CPU oracle.  Contact grid points
// only; impact grid points (a handful per horizon) go through condense_kernel (CondArgs::stage_list).
#pragma once
#include <type_traits>

#include "condense.hpp"

namespace rtoc {

#ifdef RTOC_ENABLE_PROF   // cycle stamps of the work item in the middle of the launch, slots 64.. (tools/phase_profile_cond.py)
#define CRV_PROF(k)                                                                                                   \
  do {                                                                                                                \
    if (a.prof && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0) a.prof[64 + (k)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define CRV_PROF(k) do { } while (0)
#endif

template <int NV, int NU, int NF, int NS>
struct CrvCfg {
  static constexpr int NX = 2 * NV, NP = NV - NU, LDV = NV + NF, NFP = NF > 0 ? NF : 1;
  // two row tiles of [a; f] with every row of the first one an acceleration row, three column tiles of the state with room for two
  // rider columns, two column tiles of the accelerations with room for two rider columns
  static constexpr bool OK = NF > 0 && NF == NS && NV >= 16 && LDV <= 32 && NX > 32 && NX + 2 <= 48 && NV + 2 <= 32 && NS <= 16;
  static constexpr int RC = NX - 32;   // lane of the rider column NX in column tile 2
  static constexpr int RU = NV - 16;   // lane of the rider column NV in column tile 1 of W Lam; also: first force row of row tile 1
  static constexpr int pad8(int n) { return (n + 7) & ~7; }
  static constexpr int O_LAM = 0;                                // MJtJinv, LDV x LDV
  static constexpr int O_W = O_LAM + pad8(LDV * LDV);            // factorisation scratch, then the staging of MJtJinv_dIDCdqv (LDV x NX)
  static constexpr int O_L = O_W;                                // NV x NV
  static constexpr int O_J = O_L + pad8(NV * NV);                // NF x NV
  static constexpr int O_JM = O_J + pad8(NFP * NV);
  static constexpr int O_S = O_JM + pad8(NFP * NV);
  static constexpr int O_BR = O_S + pad8(NFP * NFP);
  static constexpr int W_A = O_BR + pad8(NFP * NFP) - O_W;
  static constexpr int Y_ROOM = O_BR + pad8(NFP * NFP) - O_JM;
  static constexpr int W_SZ = W_A > pad8(LDV * NX) ? W_A : pad8(LDV * NX);
  static constexpr int O_VEC = O_W + W_SZ;
  static constexpr int V_LINV = O_VEC, V_SINV = V_LINV + pad8(NV), V_R36 = V_SINV + pad8(NFP), V_R37 = V_R36 + 32, V_QAA = V_R37 + 32,
                       V_LR = V_QAA + 32, V_LAF = V_LR + 32, V_PH = V_LAF + 32, V_PG = V_PH + pad8(2 * NV + NU);
  // friction-cone rows condensed in the kernel (cone_rows == RTOC_FRICTION_ROWS): the lower triangle of their Qqq contribution and
  // their gradient column [lq; lf], parked until the seeds / riders that take them exist
  static constexpr int V_CQQ = V_PG + pad8(2 * NV + NU), V_CG = V_CQQ + pad8(NV * (NV + 1) / 2);
  static constexpr int LDS_DOUBLES = V_CG + 32;
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
  static constexpr int MAXC = NF / 3, CONE_K = 5 * MAXC, CONE_KS = (CONE_K + 3) / 4;   // friction cones of point contacts
  static constexpr bool CONES = NF % 3 == 0 && NV + NF + 1 <= 32 && 4 * CONE_KS * 32 + 4 * CONE_KS <= W_SZ && LDS_BYTES <= 20 * 1024;
};

// WITH_CONES = false: the same kernel without the cone-row code (contexts that have none: its registers and LDS traffic cost 0.2-0.3 ms
// per 4096 x 46 even when no row is active)
template <int NV, int NU, int NF, int NS, bool WITH_CONES = true>
__global__ __launch_bounds__(64, 2) void condense_rv_kernel(CondArgs a) {
  using C = CrvCfg<NV, NU, NF, NS>;
  constexpr bool CONES = C::CONES && WITH_CONES;
  static_assert(C::OK, "shape outside the register plan");
  constexpr int NX = C::NX, NP = C::NP, LDV = C::LDV, LDF = C::NFP, RC = C::RC, RU = C::RU, LDS_ = NS > 0 ? NS : 1;
  constexpr int NT = 64, NW = 1;
  constexpr bool SPLIT = false;
  (void)SPLIT;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const Lam = smem + C::O_LAM;
  double* const stg = smem + C::O_W;
  double* const LD = stg;   // (named by the fragment for fixed-base shapes only)
  double* const sL = smem + C::O_L;
  double* const sJ = smem + C::O_J;
  double* const sJM = smem + C::O_JM;
  double* const sS = smem + C::O_S;
  double* const sBR = smem + C::O_BR;
  double* const sLinv = smem + C::V_LINV;
  double* const sSinv = smem + C::V_SINV;
  double* const sR36 = smem + C::V_R36;
  double* const sR37 = smem + C::V_R37;
  double* const sQaa = smem + C::V_QAA;
  double* const sLr = smem + C::V_LR;
  double* const sLaf = smem + C::V_LAF;
  double* const sPH = smem + C::V_PH;
  double* const sPG = smem + C::V_PG;
  double* const sCqq = smem + C::V_CQQ;
  double* const sCg = smem + C::V_CG;
  const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
  const int wv = 0, wl = lane;
  const int item = blockIdx.x;
  const int nst1 = a.stage_list ? a.nlist : a.nstages - 1;
  const int b = item / nst1;
  const int st = a.stage_list ? a.stage_list[item - b * nst1] : item - b * nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  if (g.type == RTOC_GRID_IMPACT) return;   // (condense_kernel's: CondArgs::stage_list of the impact grid points)
  const int nf = g.dimf, nvf = NV + nf, ns = g.dims;
  const double dt = grid_dt(a.grid, a.dt_inst, b, a.nstages, st);
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt, CL = SL.cdd;
  double* kr = a.kkt + ((size_t)b * a.nstages + st) * KL.stride;
  double* cr = a.cdd + ((size_t)b * a.nstages + st) * CL.stride;
  double* const Fxx = kr + KL.off[RTOC_KKT_FXX];
  double* const Fvu = kr + KL.off[RTOC_KKT_FVU];
  double* const Qxx = kr + KL.off[RTOC_KKT_QXX];
  double* const Qxu = kr + KL.off[RTOC_KKT_QXU];
  double* const Quu = kr + KL.off[RTOC_KKT_QUU];
  double* const Fx = kr + KL.off[RTOC_KKT_FX];
  double* const lx = kr + KL.off[RTOC_KKT_LX];
  double* const lu = kr + KL.off[RTOC_KKT_LU];
  double* const fx = kr + KL.off[RTOC_KKT_FFX];
  double* const hx = kr + KL.off[RTOC_KKT_HX];
  double* const hu = kr + KL.off[RTOC_KKT_HU];
  double* const scal = kr + KL.off[RTOC_KKT_SCAL];
  double* const Phix = kr + KL.off[RTOC_KKT_PHIX];
  double* const Phiu = kr + KL.off[RTOC_KKT_PHIU];
  double* const Phit = kr + KL.off[RTOC_KKT_PHIT];
  double* const Pres = kr + KL.off[RTOC_KKT_PRES];
  const double* const Phia = cr + CL.off[RTOC_CDD_PHIA];
  double* const lup = cr + CL.off[RTOC_CDD_LUP];
  double* const Qxup = cr + CL.off[RTOC_CDD_QXUP];
  double* const Quuptr = cr + CL.off[RTOC_CDD_QUUPTR];
  const double* const Dg = cr + CL.off[RTOC_CDD_DIDCDQV];
  const double* const Qffg = cr + CL.off[RTOC_CDD_QFF];
  const double* const Qqfg = cr + CL.off[RTOC_CDD_QQF];
  double* const Qffg_w = cr + CL.off[RTOC_CDD_QFF];
  double* const Qqfg_w = cr + CL.off[RTOC_CDD_QQF];
  unsigned stat = 0;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  CRV_PROF(0);

  // ================= HBM -> registers =================
  // what the factorisation starts from: M and J (dCda), 16 B / 8 B per lane
  constexpr int H_L = (NV * NV + 1) / 2, N_L0 = (H_L + 63) / 64, N_J0 = (C::NFP * NV + 63) / 64;
  dbl2 fL[N_L0];
  double fJ[N_J0];
#pragma unroll
  for (int k = 0; k < N_L0; ++k) {
    const int e = lane + k * 64;
    fL[k] = reinterpret_cast<const dbl2*>(cr + CL.off[RTOC_CDD_DIDDA])[e < H_L ? e : 0];
  }
#pragma unroll
  for (int k = 0; k < N_J0; ++k) {
    const int e = lane + k * 64;
    fJ[k] = cr[CL.off[RTOC_CDD_DCDA] + (e < C::NFP * NV ? e : 0)];
  }
  // joint-limit rows: the packed descriptor of the lane's primal entries (the same for every work item: an L2 hit), then their data
  constexpr int NE = 3 * NV + NU, NEP = (NE + 63) / 64;
  const bool box_on = a.con != nullptr;
  double* const nr = box_on ? a.con + ((size_t)b * a.nstages + st) * a.nl.stride : nullptr;
  int4 pd[NEP];
  int ent0[NEP], ent1[NEP];
#pragma unroll
  for (int k = 0; k < NEP; ++k) {
    pd[k] = make_int4(-1, -1, 0, 0);
    ent0[k] = ent1[k] = 0;
  }
  if (box_on) {
#pragma unroll
    for (int k = 0; k < NEP; ++k) {
      const int t = lane + 64 * k < NE ? lane + 64 * k : 0;
      pd[k] = a.pair[t];
      ent0[k] = a.entry[t] + 2;
      ent1[k] = a.entry[t + 1];
    }
  }
  // ================= friction-cone rows (Constraints::condenseSlackAndDual, friction_cone.cpp:194-233) inside the kernel ============
  // With G the stacked Jacobian of the active cone rows (5 per active contact; columns: q and the contact's own force), R = diag(dual /
  // slack) and c the condensing coefficients, every contribution is a block of the ONE Gram product G^T [R G | c] (cone_condense_body,
  // friction_cone.hpp -- which read-modify-writes them in HBM ahead of the condensation).  Here its tiles go where the condensation
  // takes them from: (q, q) -> the seeds of V (lower triangle parked in LDS), (f, q) -> E' (same C layout), (f, f) -> W1 (symmetric:
  // its C layout IS the A fragment), the rider column -> lx / lf.  The updated Qqf, Qff, lf go back to the record for the expansion.
  const bool cones = CONES && a.cone_rows == RTOC_FRICTION_ROWS && a.cone_dim == 3;
  const int nact = cones ? nf / 3 : 0;
  d4 cep[2];      // what the cone rows add to E' (tile row 1: rows f = q + 4r - RU, columns li + 16 tc)
  double cw1[4];  // ... and to W1 (A fragment)
#pragma unroll
  for (int tc = 0; tc < 2; ++tc) cep[tc] = zero4();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) cw1[ks] = 0.0;
  if constexpr (CONES) {
    if (lane < 32) sCg[lane] = 0.0;
    for (int e = lane; e < NV * (NV + 1) / 2; e += 64) sCqq[e] = 0.0;
    if (nact > 0) {
      constexpr int K = C::CONE_K, KS = C::CONE_KS, GLD = 32, NC = NV + NF, MAXC = C::MAXC;
      constexpr int ND = (MAXC * 5 * NV + 63) / 64, NG = (MAXC * 15 + 63) / 64;
      const double* const cone = a.cone + ((size_t)b * a.nstages + st) * a.cone_stride;
      double* const cnr = a.cone_con + ((size_t)b * a.nstages + st) * a.nl.stride;
      double gdq[ND], gdf[NG];
#pragma unroll
      for (int p = 0; p < ND; ++p) {
        const int e = lane + 64 * p;
        gdq[p] = cone[e < a.cone_contacts * 5 * NV ? e : 0];
      }
#pragma unroll
      for (int p = 0; p < NG; ++p) {
        const int e = lane + 64 * p;
        gdf[p] = cone[a.cone_dgdf_off + (e < a.cone_contacts * 15 ? e : 0)];
      }
      const int rw = a.cone_row0 + (lane < 5 * a.cone_contacts ? lane : 0);
      const double cslack = cnr[a.nl.off[RTOC_CON_SLACK] + rw], cdual = cnr[a.nl.off[RTOC_CON_DUAL] + rw],
                   cresid = cnr[a.nl.off[RTOC_CON_RESIDUAL] + rw], ccmpl = cnr[a.nl.off[RTOC_CON_CMPL] + rw];
      // G (K-major, leading dimension 32: columns q | f | rider c) and diag R into the work region (free until the factorisation)
      double* const Gs = stg;
      double* const rs = stg + 4 * KS * GLD;
      for (int e = lane; e < 4 * KS * GLD; e += 64) Gs[e] = 0.0;
      wave_lds_sync_();
#pragma unroll
      for (int p = 0; p < ND; ++p) {   // dg_dq of contact k: 5 x NV, column-major
        const int e = lane + 64 * p, k = e / (5 * NV), w = e % (5 * NV);
        if (e < nact * 5 * NV) Gs[(5 * k + w % 5) * GLD + w / 5] = gdq[p];
      }
#pragma unroll
      for (int p = 0; p < NG; ++p) {   // dg_df of contact k: 5 x 3, on the contact's own force columns
        const int e = lane + 64 * p, k = e / 15, w = e % 15;
        if (e < nact * 15) Gs[(5 * k + w % 5) * GLD + NV + k * 3 + w / 5] = gdf[p];
      }
      if (lane < 4 * KS) {
        const bool on = lane < 5 * nact;
        if (on) {
          const double c = (cdual * cresid - ccmpl) / cslack;   // computeCondensingCoeffcient<5> (:202)
          cnr[a.nl.off[RTOC_CON_COND] + rw] = c;
          Gs[lane * GLD + NC] = c;
        }
        rs[lane] = on ? cdual / cslack : 0.0;                   // (:211-212)
      }
      wave_lds_sync_();
      d4 g4[2][2];
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) g4[ta][tb] = zero4();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kk = 4 * ks + q;
        const double r = rs[kk];
        double gv[2], bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          gv[t] = Gs[kk * GLD + 16 * t + li];
          bv[t] = gv[t] * ((16 * t + li == NC) ? 1.0 : r);
        }
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) g4[ta][tb] = mfma16(gv[ta], bv[tb], g4[ta][tb]);
      }
      // where the tiles go
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = q + 4 * r + 16 * ta, col = li + 16 * tb;
            if (row < NV && col <= row) sCqq[row * (row + 1) / 2 + col] = g4[ta][tb][r];   // (q, q): Qqq (:215-216), lower triangle
            if (tb == 1 && col == NC && row < NC) sCg[row] = g4[ta][tb][r];               // rider column: lq (:206) | lf (:207-208)
          }
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = q + 4 * r - RU, col = li + 16 * tc;
          cep[tc][r] = (f >= 0 && f < NF && col < NV) ? g4[1][tc][r] : 0.0;                 // (f, q): Qqf^T (:217-218)
        }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int m = li - RU, k = 4 * ks + q - RU;
        cw1[ks] = (m >= 0 && m < NF && k >= 0 && k < NF && m / 3 == k / 3) ? g4[1][1][ks] : 0.0;   // (f, f): Qff, the contact's own 3 x 3 block (:219-220)
      }
      wave_lds_sync_();   // the work region is the factorisation's from here on
    }
  }

  // ---- everything else the work item reads is requested by issue_loads(), called right behind the factorisation of M (hook of the
  //      fragment): ~150 fewer registers live across its column steps, and the loads arrive under the rest of the assembly (requested
  //      at the very top instead: the same time to 0.5 %, measured in one process) ----
  double dB[3][8];
  const int rowl = lane < 32 ? lane : 0;
  const bool arow = rowl < NV;
  const int fl = (rowl - NV >= 0 && rowl - NV < NF) ? rowl - NV : 0;
  double vLa, vHa, vQaa;
  double bs0[NEP], bd0[NEP], bq0[NEP], bc0[NEP], bs1[NEP], bd1[NEP], bq1[NEP], bc1[NEP];
  double w1raw[4];
  d4 epraw[2];
  // the accumulator seeds of the Schur products (see below); the first one is requested here, ahead of the factorisation
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  auto seed_v = [&](auto TM, d4 (&raw)[3]) {
    constexpr int tm = decltype(TM)::value;
#pragma unroll
    for (int tn = 0; tn < 3; ++tn) {
      const int n = li + 16 * tn, nc = n < NX ? n : NX - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = q + 4 * r + 16 * tm;
        if (tm < 2 || r == 0) {
          raw[tn][r] = Qxx[nc + (size_t)m * NX];
        } else if (r == 1) {
          // q = 0: lx, q = 1: hx, and h in column NX; q = 2, 3: nobody's
          const double* p = (q == 0) ? lx + nc : ((tn == 2 && li == RC) ? scal + RTOC_KKT_SCAL_H : hx + nc);
          raw[tn][r] = *p;
        } else {
          raw[tn][r] = 0.0;
        }
      }
    }
  };
  d4 sa[3], sb[3];
  auto issue_loads = [&]() {
    // D = dIDCdqv in the B layout of its product with Lam: dB[tc][ks] = D[4 ks + q][li + 16 tc]; column NX: IDC.  RAW values: rows
    // beyond the active contact dimension are unspecified in the record and are dropped where dB is USED.
    {
      const double* p01 = Dg + q + li * LDV;
      const double* p2 = (li < RC) ? Dg + q + (li + 32) * LDV : ((li == RC) ? cr + CL.off[RTOC_CDD_IDC] + q : Dg + q + (NX - 1) * LDV);
  #pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = (4 * ks + 3 < LDV) ? 4 * ks : ((4 * ks + q < LDV) ? 4 * ks : 4 * ks - 4);   // (the last rows of the padding: any valid address)
        dB[0][ks] = p01[kk];
        dB[1][ks] = p01[kk + 16 * LDV];
        dB[2][ks] = p2[kk];
      }
    }
    // the vectors: one lane per row of [a; f]
    vLa = cr[(arow ? CL.off[RTOC_CDD_LA] + rowl : CL.off[RTOC_CDD_LF] + fl)];
    vHa = cr[(arow ? CL.off[RTOC_CDD_HA] + rowl : CL.off[RTOC_CDD_HF] + fl)];
    vQaa = cr[CL.off[RTOC_CDD_QAA] + (arow ? rowl : 0)];
    // joint-limit row data (descriptor -> data: a dependent pair of round trips, the second one under the factorisation)
  #pragma unroll
    for (int k = 0; k < NEP; ++k) bs0[k] = bs1[k] = 1.0, bd0[k] = bq0[k] = bc0[k] = bd1[k] = bq1[k] = bc1[k] = 0.0;
    if (box_on) {
      const int* no = a.nl.off;
  #pragma unroll
      for (int k = 0; k < NEP; ++k) {
        const int r0 = pd[k].x >= 0 ? pd[k].x : 0, r1 = pd[k].y >= 0 ? pd[k].y : 0;
        bs0[k] = nr[no[RTOC_CON_SLACK] + r0], bd0[k] = nr[no[RTOC_CON_DUAL] + r0], bq0[k] = nr[no[RTOC_CON_RESIDUAL] + r0],
        bc0[k] = nr[no[RTOC_CON_CMPL] + r0];
        bs1[k] = nr[no[RTOC_CON_SLACK] + r1], bd1[k] = nr[no[RTOC_CON_DUAL] + r1], bq1[k] = nr[no[RTOC_CON_RESIDUAL] + r1],
        bc1[k] = nr[no[RTOC_CON_CMPL] + r1];
      }
    }

    // raw: the force block of W1 = diag(Qaa[16..], Qff) in the A layout, E' = Qqf^T in the rows of the forces (row tile 1; C layout = B
    // layout); clamped addresses, masked where they are consumed
  #pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int m = li, k = 4 * ks + q;
      const bool in = m >= RU && k >= RU && m - RU < NF && k - RU < NF;
      w1raw[ks] = Qffg[in ? (m - RU) + (k - RU) * LDF : ks];
    }
  #pragma unroll
    for (int tc = 0; tc < 2; ++tc)
  #pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = q + 4 * r - RU, col = li + 16 * tc;
        epraw[tc][r] = Qqfg[(f >= 0 && f < NF && col < NV) ? col + f * NV : r];
      }
    seed_v(I0{}, sa);
  };

  // ================= computeMJtJinv (robot.hxx:642-684) in LDS: the fragment of condense.hpp on one wave =================
  for (int e = lane; e < LDV * LDV; e += 64) Lam[e] = 0.0;   // inactive rows / columns stay zero
#pragma unroll
  for (int k = 0; k < N_L0; ++k) {
    const int e = lane + k * 64;
    if (e < H_L) reinterpret_cast<dbl2*>(sL)[e] = fL[k];
  }
#pragma unroll
  for (int k = 0; k < N_J0; ++k) {
    const int e = lane + k * 64;
    if (e < C::NFP * NV) sJ[e] = (e % LDF < nf) ? fJ[k] : 0.0;
  }
  wave_lds_sync_();
  {
    const double* const J = sJ;
    (void)J;
#define RTOC_MJ_SYNC() wave_lds_sync_()
#define RTOC_J_IN_D false
#define RTOC_MJ_AFTER_LLT asm volatile("" ::: "memory"); issue_loads();
#define RTOC_MJ_AFTER_B1
#pragma push_macro("RTOC_CPROF")
#undef RTOC_CPROF
#define RTOC_CPROF(k) CRV_PROF(32 + (k))   // (the fragment's stamps in this kernel's own slots)
#include "condense_mjtjinv.inc"
#pragma pop_macro("RTOC_CPROF")
#undef RTOC_MJ_AFTER_LLT
#undef RTOC_MJ_AFTER_B1
#undef RTOC_J_IN_D
#undef RTOC_MJ_SYNC
  }
  copy_s2g_flat16<64>(cr + CL.off[RTOC_CDD_MJTJINV], Lam, LDV * LDV, lane);
  CRV_PROF(3);
  if (lane < 32) {   // (their loads came in behind the 24 loads of D: written here, not ahead of the factorisation)
    sQaa[lane] = arow ? vQaa : 0.0;
    const double lfc = vLa + (CONES ? sCg[lane] : 0.0);       // lf with the cone rows' gradient (zero without cone rows)
    sR36[lane] = arow ? -vLa : ((lane - NV < nf) ? lfc : 0.0);   // -[la; -lf]
    if (CONES && nact > 0 && !arow && lane - NV < nf) cr[CL.off[RTOC_CDD_LF] + lane - NV] = lfc;
    sR37[lane] = arow ? -vHa : ((lane - NV < nf) ? vHa : 0.0);   // -[ha; -hf]
  }
  wave_lds_sync_();

  // ================= PDIPM slack / dual elimination of the joint-limit rows (constraints.cpp:322-357) =================
  // one lane per primal entry (q_k, v_k, u_k, a_k), every entry accumulated by a single lane in row order (no atomics); the
  // acceleration limits act on Qaa.diagonal() / la ahead of everything that reads them (contact_dynamics.cpp:68-86)
#pragma unroll
  for (int k = 0; k < NEP; ++k) {
    const int t = lane + 64 * k;
    if (t < NE) {
      double hess = 0.0, grad = 0.0;
      if (box_on) {
        const int* no = a.nl.off;
        if (pd[k].x >= 0 && g.time_stage >= (pd[k].z >> 8)) {
          const double cond = (bd0[k] * bq0[k] - bc0[k]) / bs0[k];
          nr[no[RTOC_CON_COND] + pd[k].x] = cond;
          hess += bd0[k] / bs0[k];
          grad += (double)(signed char)(pd[k].z & 0xff) * cond;
        }
        if (pd[k].y >= 0 && g.time_stage >= (pd[k].w >> 8)) {
          const double cond = (bd1[k] * bq1[k] - bc1[k]) / bs1[k];
          nr[no[RTOC_CON_COND] + pd[k].y] = cond;
          hess += bd1[k] / bs1[k];
          grad += (double)(signed char)(pd[k].w & 0xff) * cond;
        }
        const int* rowid = a.entry + (NE + 1);
        for (int e = ent0[k]; e < ent1[k]; ++e) {   // further rows on the same entry (none for joint limits)
          const int r = rowid[e];
          const rtoc_box_row row = a.rows[r];
          if (g.time_stage >= row.level) {
            const double slack = nr[no[RTOC_CON_SLACK] + r], dual = nr[no[RTOC_CON_DUAL] + r];
            const double cond = (dual * nr[no[RTOC_CON_RESIDUAL] + r] - nr[no[RTOC_CON_CMPL] + r]) / slack;
            nr[no[RTOC_CON_COND] + r] = cond;
            hess += dual / slack;
            grad += row.sign * cond;
          }
        }
      }
      if (t < 2 * NV + NU) {
        sPH[t] = hess;
        sPG[t] = grad;
      } else if (box_on && pd[k].x >= 0) {
        const int i = t - (2 * NV + NU);
        const double qn = sQaa[i] + hess;
        sQaa[i] = qn;
        sR36[i] -= grad;   // la += grad
        cr[CL.off[RTOC_CDD_QAA] + i] = qn;
      }
    }
  }
  wave_lds_sync_();

  // ================= Lam -> registers: lam[tr][tc][r] = Lam[q + 4r + 16 tr][li + 16 tc] (read through the mirror index) =========
  d4 lam[2][2];
  auto read_lam = [&]() {
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = q + 4 * r + 16 * tr, col = li + 16 * tc;
          const bool ok = row < LDV && col < LDV;
          const double v = Lam[ok ? col + row * LDV : 0];
          lam[tr][tc][r] = ok ? v : 0.0;
        }
  };
  read_lam();
  // W1 = diag(Qaa[16..], Qff) as the A operand of its products: w1[ks] = W1[li][4 ks + q]; E' masked (both requested ahead of the
  // factorisation)
  double w1[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int m = li, k = 4 * ks + q;
    const bool fblk = m >= RU && k >= RU && m - RU < nf && k - RU < nf;
    const double va = sQaa[16 + (m < RU ? m : 0)];
    w1[ks] = (m < RU) ? ((k == m) ? va : 0.0) : (fblk ? w1raw[ks] + cw1[ks] : 0.0);
    if (CONES && nact > 0 && fblk && (m - RU) / 3 == (k - RU) / 3) Qffg_w[(m - RU) + (k - RU) * LDF] = w1[ks];   // for the expansion
  }
  d4 ep[2];
#pragma unroll
  for (int tc = 0; tc < 2; ++tc)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = q + 4 * r - RU, col = li + 16 * tc;
      const bool ok = f >= 0 && f < nf && col < NV;
      ep[tc][r] = ok ? epraw[tc][r] + cep[tc][r] : 0.0;
      if (CONES && nact > 0 && ok) Qqfg_w[col + f * NV] = ep[tc][r];   // for the expansion
    }
  double qaa0[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) qaa0[r] = sQaa[q + 4 * r];
  CRV_PROF(4);

  // ================= LD = Lam [D | IDC] (contact_dynamics.cpp:64-65) =================
  d4 ld[2][3];
#pragma unroll
  for (int tr = 0; tr < 2; ++tr)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) ld[tr][tc] = zero4();
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) {
      const double bv = (4 * ks + 3 < NV || 4 * ks + q < nvf) ? dB[tc][ks] : 0.0;
#pragma unroll
      for (int tr = 0; tr < 2; ++tr) ld[tr][tc] = mfma16(lam[ks / 4][tr][ks % 4], bv, ld[tr][tc]);
    }
  // out through LDS: MJtJinv_dIDCdqv for the expansion, the velocity rows of Fxx (:132-136), MJtJinv_IDC
#pragma unroll
  for (int tr = 0; tr < 2; ++tr)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = q + 4 * r + 16 * tr, col = li + 16 * tc;
        if (row < LDV && col < NX) stg[row + col * LDV] = ld[tr][tc][r];
        if (tc == 2 && col == NX) sLr[row] = ld[tr][tc][r];
      }
  CRV_PROF(5);

  // ================= Xn = -Qafqv = W LD + E' (:67-80); column NX: -laf, column NX + 1: -haf =================
  d4 xn[2][3];
#pragma unroll
  for (int tc = 0; tc < 3; ++tc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) xn[0][tc][r] = qaa0[r] * ld[0][tc][r];
    d4 acc = tc < 2 ? ep[tc] : zero4();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = mfma16(w1[ks], ld[1][tc][ks], acc);
    xn[1][tc] = acc;
  }
  {
    const double m36 = (li == RC) ? 1.0 : 0.0;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = q + 4 * r + 16 * tr;
        const double v36 = __builtin_fma(m36, sR36[row], xn[tr][2][r]);
        xn[tr][2][r] = (li == RC + 1) ? sR37[row] : v36;
        if (li == RC) sLaf[row] = -v36;
      }
  }
  // LD column NX + 1 := MJtJinv_IDC / dt: the hx rider of the second term (as a column of the B operand it only reaches V[:, NX + 1],
  // which nobody reads)
  {
    const double rdt = 1.0 / dt;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double sh = dpp_from_left<1>(ld[tr][2][r]);
        ld[tr][2][r] = (li == RC + 1) ? sh * rdt : ld[tr][2][r];
      }
  }
  static_assert(RC > RU, "the rider columns move towards lower lanes");
  wave_lds_sync_();   // the staging of LD, Lr, laf is complete
  CRV_PROF(6);
  // ---- stores that only need the staging: MJtJinv_dIDCdqv, the velocity rows of Fxx, Fx, the expansion's vectors ----
  copy_s2g_flat16<64>(cr + CL.off[RTOC_CDD_MJD], stg, LDV * NX, lane);
#pragma unroll
  for (int k = 0; k < (NV * NX + 63) / 64; ++k) {
    const int e = lane + 64 * k;
    const int j = e / NV, i = e - j * NV;
    if (e < NV * NX) Fxx[(NV + i) + (size_t)j * NX] = -dt * stg[(e < NV * NX) ? i + j * LDV : 0] + ((j == NV + i) ? 1.0 : 0.0);
  }
  if (lane < NV) Fx[NV + lane] = Fx[NV + lane] - dt * sLr[lane];
  if (lane < nvf) {
    cr[CL.off[RTOC_CDD_MJIDC] + lane] = sLr[lane];
    cr[CL.off[RTOC_CDD_LAF] + lane] = sLaf[lane];
    cr[CL.off[RTOC_CDD_HAF] + lane] = -sR37[lane];
  }
  // Fvu = dt Lam[a, u] (:136) through the mirror index: lanes along the rows
#pragma unroll
  for (int tr = 0; tr < 2; ++tr)
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = q + 4 * r + 16 * tr, i = li + 16 * tc;
        if (i < NV && col >= NP && col < NV) Fvu[i + (size_t)(col - NP) * NV] = dt * lam[tr][tc][r];
      }
  CRV_PROF(7);
  const double inv = 1.0 / (double)g.num_grids_in_phase;
  if (lane < NX) fx[lane] = fx[lane] * inv;
  if (lane == 0) {
    const double qtt = scal[RTOC_KKT_SCAL_QTT] * inv * inv;
    scal[RTOC_KKT_SCAL_QTT] = qtt;
    scal[RTOC_KKT_SCAL_QTT_PREV] = -qtt;
  }

  d4 wlm[2][2];
  // ================= the three Schur products, software-pipelined: the accumulator seeds of a product (the stored block it updates,
  // read through the index its C layout stores to) are requested one product ahead, RAW -- masks and the joint-limit rows' terms are
  // applied where the seed is consumed, so that nothing waits for the loads before the previous product's MFMAs are issued ==========
  // ---- V = Xn^T LD + LD^T E' = (Qxx update)^T (:90-93, :110-113, :123-130): V[m][n], m = q + 4r + 16 tm, n = li + 16 tn, is entry
  //      (n, m) of the condensed Qxx (lanes along a column of Qxx: 128-byte runs); row NX: lx, row NX + 1: hx, V[NX + 1][NX]: h ----
  auto run_v = [&](auto TM, d4 (&acc)[3]) {
    constexpr int tm = decltype(TM)::value;
#pragma unroll
    for (int tn = 0; tn < 3; ++tn) {
      const int n = li + 16 * tn, nc = n < NX ? n : NX - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = q + 4 * r + 16 * tm;
        if (tm < 2 || r == 0) {
          if (tm == tn) acc[tn][r] = __builtin_fma((m == n) ? 1.0 : 0.0, sPH[nc], acc[tn][r]);   // the joint-limit rows' Hessian
          if (CONES && tm < 2 && tn < 2) {                                                    // the cone rows' Qqq (symmetric)
            const int hi = m > n ? m : n, lo = m > n ? n : m;
            const bool in = hi < NV;
            acc[tn][r] += in ? sCqq[in ? hi * (hi + 1) / 2 + lo : 0] : 0.0;
          }
        } else if (r == 1) {
          acc[tn][r] = __builtin_fma((q == 0) ? 1.0 : 0.0, sPG[nc] + ((CONES && nc < NV) ? sCg[nc < NV ? nc : 0] : 0.0), acc[tn][r]);   // ... and gradient, the cone rows' lq
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int tn = 0; tn < 3; ++tn) acc[tn] = mfma16(xn[ks / 4][tm][ks % 4], ld[ks / 4][tn][ks % 4], acc[tn]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) acc[tn] = mfma16(ld[1][tm][ks], ep[tn][ks], acc[tn]);
#pragma unroll
    for (int tn = 0; tn < 3; ++tn) {
      const int n = li + 16 * tn;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = q + 4 * r + 16 * tm;
        if (tm < 2 || r == 0) {
          if (n < NX) Qxx[n + (size_t)m * NX] = acc[tn][r];
        } else if (r == 1) {
          if (n < NX && q == 0) lx[n] = acc[tn][r];
          if (n < NX && q == 1) hx[n] = acc[tn][r] * inv;
          if (tn == 2 && n == NX && q == 1) scal[RTOC_KKT_SCAL_H] = acc[tn][r] * inv;
        }
      }
    }
  };
  // ---- V2 = -(Qxu_full)^T = -old + WL^T LD + Lam_a E' (:95-100); rows < np: Qxu_passive ----
  auto seed_v2 = [&](auto TM, d4 (&raw)[3]) {
    constexpr int tm = decltype(TM)::value;
#pragma unroll
    for (int tn = 0; tn < 3; ++tn) {
      const int j = li + 16 * tn, jc = j < NX ? j : NX - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = q + 4 * r + 16 * tm;
        raw[tn][r] = Qxu[jc + (size_t)((i >= NP && i < NV) ? i - NP : r) * NX];   // (a distinct clamp target per register)
      }
    }
  };
  auto run_v2 = [&](auto TM, d4 (&acc)[3]) {
    constexpr int tm = decltype(TM)::value;
#pragma unroll
    for (int tn = 0; tn < 3; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = q + 4 * r + 16 * tm;
        acc[tn][r] = (i >= NP && i < NV) ? -acc[tn][r] : 0.0;
      }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int tn = 0; tn < 3; ++tn) acc[tn] = mfma16(wlm[ks / 4][tm][ks % 4], ld[ks / 4][tn][ks % 4], acc[tn]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) acc[tn] = mfma16(lam[1][tm][ks], ep[tn][ks], acc[tn]);
#pragma unroll
    for (int tn = 0; tn < 3; ++tn) {
      const int j = li + 16 * tn;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = q + 4 * r + 16 * tm;
        if (j < NX && i < NP) Qxup[j + (size_t)i * NX] = -acc[tn][r];
        if (j < NX && i >= NP && i < NV) Qxu[j + (size_t)(i - NP) * NX] = -acc[tn][r];
      }
    }
  };
  // ---- QU = Lam_a WL (:102-107, :114-121): the a x a block is symmetric -> stored through the mirror index; column NV: lu_passive / lu,
  //      column NV + 1: hu ----
  auto seed_qu = [&](auto TM, d4 (&raw)[3]) {
    constexpr int tm = decltype(TM)::value;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int c = li + 16 * tn;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = q + 4 * r + 16 * tm;
        const bool iu = i >= NP && i < NV, cu = c >= NP && c < NV;
        const int iuc = iu ? i - NP : 0;
        const double* p = (c == NV) ? ((i < NP) ? lup + i : lu + iuc) : ((c == NV + 1) ? hu + iuc : Quu + (cu ? c - NP : r) + (size_t)iuc * NU);
        raw[tn][r] = *p;
      }
    }
  };
  auto run_qu = [&](auto TM, d4 (&acc)[3]) {
    constexpr int tm = decltype(TM)::value;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int c = li + 16 * tn;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = q + 4 * r + 16 * tm;
        const bool iu = i >= NP && i < NV, cu = c >= NP && c < NV;
        const int iuc = iu ? i - NP : 0;
        const bool use = (c == NV) ? (i < NV) : (iu && (cu || c == NV + 1));
        const double add = sPH[2 * NV + iuc] * ((iu && c == i) ? 1.0 : 0.0) + sPG[2 * NV + iuc] * ((iu && c == NV) ? 1.0 : 0.0);
        acc[tn][r] = use ? acc[tn][r] + add : 0.0;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) acc[tn] = mfma16(lam[ks / 4][tm][ks % 4], wlm[ks / 4][tn][ks % 4], acc[tn]);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int c = li + 16 * tn;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = q + 4 * r + 16 * tm;
        const bool iu = i >= NP && i < NV;
        if (iu && c >= NP && c < NV) Quu[(c - NP) + (size_t)(i - NP) * NU] = acc[tn][r];
        if (iu && c < NP) Quuptr[c + (size_t)(i - NP) * NP] = acc[tn][r];
        if (c == NV && i < NP) lup[i] = acc[tn][r];
        if (c == NV && iu) lu[i - NP] = acc[tn][r];
        if (c == NV + 1 && iu) hu[i - NP] = acc[tn][r] * inv;
      }
    }
  };
  {
    seed_v(I1{}, sb);
    run_v(I0{}, sa);
    seed_v(I2{}, sa);
    run_v(I1{}, sb);
    seed_v2(I0{}, sb);
    run_v(I2{}, sa);
    CRV_PROF(10);
    // Lam again from LDS: its 32 registers were free for the seeds during V
    asm volatile("" ::: "memory");
    read_lam();
    // ================= WL = W Lam (Qafu_full, :81-88); rider columns NV, NV + 1: laf, haf (after V: Xn is dead) =================
#pragma unroll
    for (int tc = 0; tc < 2; ++tc) {
#pragma unroll
      for (int r = 0; r < 4; ++r) wlm[0][tc][r] = qaa0[r] * lam[0][tc][r];
      d4 acc = zero4();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = mfma16(w1[ks], lam[1][tc][ks], acc);
      wlm[1][tc] = acc;
    }
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // laf, haf (the columns NX, NX + 1 of -Xn) from their LDS copies
        const int row = q + 4 * r + 16 * tr;
        wlm[tr][1][r] = (li == RU) ? sLaf[row] : ((li == RU + 1) ? -sR37[row] : wlm[tr][1][r]);
      }
    seed_v2(I1{}, sa);
    run_v2(I0{}, sb);
    seed_qu(I0{}, sb);
    run_v2(I1{}, sa);
    CRV_PROF(11);
    seed_qu(I1{}, sa);
    run_qu(I0{}, sb);
    run_qu(I1{}, sa);
    CRV_PROF(12);
  }

  // ================= switching constraint (:138-153): Phix -= Phia LD_a, Phiu = Phia Lam[a, u]; rider column NX: Phia MJtJinv_IDC ==
  if (NS > 0 && ns > 0) {
    constexpr int KSA = (NV + 3) / 4;
    double pa[KSA];
#pragma unroll
    for (int ks = 0; ks < KSA; ++ks) {
      const int k = 4 * ks + q;
      const bool ok = li < ns && k < NV;
      const double v = Phia[ok ? li + k * LDS_ : 0];
      pa[ks] = ok ? v : 0.0;
    }
    d4 acc[3];
#pragma unroll
    for (int tn = 0; tn < 3; ++tn) acc[tn] = zero4();
#pragma unroll
    for (int ks = 0; ks < KSA; ++ks)
#pragma unroll
      for (int tn = 0; tn < 3; ++tn) acc[tn] = mfma16(pa[ks], ld[ks / 4][tn][ks % 4], acc[tn]);
#pragma unroll
    for (int tn = 0; tn < 3; ++tn) {
      const int j = li + 16 * tn;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s = q + 4 * r;
        if (s < ns && j < NX) Phix[s + j * LDS_] -= acc[tn][r];
        if (s < ns && tn == 2 && j == NX) {
          Phit[s] = (Phit[s] - acc[tn][r]) * inv;
          Pres[s] -= acc[tn][r];
        }
      }
    }
    d4 acu[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) acu[tn] = zero4();
#pragma unroll
    for (int ks = 0; ks < KSA; ++ks)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) acu[tn] = mfma16(pa[ks], lam[ks / 4][tn][ks % 4], acu[tn]);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int c = li + 16 * tn;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s = q + 4 * r;
        if (s < ns && c >= NP && c < NV) Phiu[s + (c - NP) * LDS_] = acu[tn][r];
      }
    }
  }
  CRV_PROF(9);
  if (stat) atomicOr(&a.status[b], stat);
}


}  // namespace rtoc

// riccati_scan.hpp -- HIP kernels of the horizon scan (see riccati_scan_core.hpp for the algorithm).
//
// Launch sequence of one backward recursion with RTOC_OPT_BACKWARD_SCAN (rtoc_capi.hip: launch_backward_scan):
//   scan_element_kernel    grid (nstages, batch)      every grid point -> its interval element
//   scan_combine_kernel    grid (nstages - d, batch, 2) for d = 1, 2, 4, ... : element(i) <- element(i) o element(i+d);
//                                                     elements that reach the terminal grid point become value
//                                                     records (P_i, s_i)
//   riccati_backward_kernel in its one-stage mode, grid (batch, nstages): policies of all grid points
// One workgroup per (grid point, instance); the elements ping-pong between two HBM buffers
// (3 NX^2 + 2 NX doubles per grid point: 32 KB for ANYmal), which stay in the 4 MB L2 of the XCD for the
// handful of instances this path is meant for.
#pragma once
#include "device_utils.hpp"
#include "riccati_scan_core.hpp"
#include "riccati_scan_sto.hpp"

namespace rtoc {

struct ScanArgs {
  const double* kkt;      // [batch][nstages][kkt stride]
  const rtoc_grid* grid;  // [nstages] (device)
  uint32_t* status;       // [batch]
  const double* src;      // elements before this level [batch][nstages][EltLayout::STRIDE]
  double* dst;            // elements after this level
  double* ps;             // value records [batch][nstages][EltLayout::PS_STRIDE]
  int nstages;
  int batch;  // instances [first, batch) are processed by this launch
  int first;
  int dist;   // distance d of this combination level
};

constexpr int SCAN_ELT_NT = 256;  // element kernel
// combination kernel, two workgroups per combination (grid z): the waves share the MFMA tiles; up to 8 lanes per
// column of the elimination (2 NX + 1 columns per workgroup)
#ifdef RTOC_SCAN_FORCE_NT
constexpr int scan_comb_nt(int) { return RTOC_SCAN_FORCE_NT; }  // tuning probes only
#else
constexpr int scan_comb_nt(int nv) {
  const int want = (8 * (4 * nv + 1) + 63) / 64 * 64;
  return want < 256 ? 256 : want > 1024 ? 1024 : want;
}
#endif

template <int NV, int NU, int NS>
__global__ __launch_bounds__(SCAN_ELT_NT) void scan_element_kernel(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using E = scan::EltLayout<NV>;
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  const int st = blockIdx.x, b = a.first + blockIdx.y;
  if (b >= a.batch || st >= a.nstages) return;
  const rtoc_grid g = a.grid[st];
  const size_t rec = (size_t)b * a.nstages + st;
  const unsigned stat = scan::element_body<NV, NU, NS, SCAN_ELT_NT>(
      g, a.kkt + rec * SL.kkt.stride, a.dst + rec * E::STRIDE, a.ps + rec * E::PS_STRIDE, smem, threadIdx.x);
  if (stat && threadIdx.x == 0) atomicOr(&a.status[b], stat);
}

// two workgroups per CU where a workgroup is at most 10 waves (ANYmal): a few instances then share the chip better
constexpr int scan_comb_min_waves(int nv) { return scan_comb_nt(nv) <= 640 ? 5 : 1; }
template <int NV>
__global__ __launch_bounds__(scan_comb_nt(NV), scan_comb_min_waves(NV)) void scan_combine_kernel(ScanArgs a) {
  constexpr int SCAN_NT = scan_comb_nt(NV);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using E = scan::EltLayout<NV>;
  const int i = blockIdx.x, b = a.first + blockIdx.y, part = blockIdx.z;
  if (b >= a.batch) return;
  int j;
  bool closed2;
  if (!scan::level_plan(a.nstages, a.dist, i, &j, &closed2)) return;
  if (part == 1 && closed2) return;  // value records have no C
  const size_t inst = (size_t)b * a.nstages;
  const double* e1 = a.src + (inst + i) * E::STRIDE;
  const double* e2 = a.src + (inst + j) * E::STRIDE;
  const double* p2 = a.ps + (inst + j) * E::PS_STRIDE;
  const unsigned stat = scan::combine_body<NV, SCAN_NT>(
      e1, closed2 ? p2 + E::PS_P : e2 + E::OFF_J, closed2 ? p2 + E::PS_S : e2 + E::OFF_ETA, e2 + E::OFF_A,
      e2 + E::OFF_B, e2 + E::OFF_C, closed2, part, a.dst + (inst + i) * E::STRIDE,
      a.ps + (inst + i) * E::PS_STRIDE, smem, threadIdx.x);
  if (stat && threadIdx.x == 0) atomicOr(&a.status[b], stat);
}

// ---- grids with switching-time optimisation: stage-parallel preparation + serial vector pass (riccati_scan_sto.hpp) ----
struct StoScanArgs {
  const double* kkt;
  double* ric;
  const rtoc_grid* grid;
  uint32_t* status;
  const double* ps;   // the scan's value records [batch][nstages][PS_STRIDE]
  double* scr;        // [batch][nstages][StoScratch::STRIDE]
  int nstages, batch, first;
  double max_dts0;
  long long* prof;    // phase stamps of instance 0 (RTOC_ENABLE_PROF builds), else nullptr
};
constexpr int SCAN_STO_PREP_NT = 128;

template <int NV, int NU, int NS>
__global__ __launch_bounds__(SCAN_STO_PREP_NT) void scan_sto_prep_kernel(StoScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using E = scan::EltLayout<NV>;
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  const int st = blockIdx.x, b = a.first + blockIdx.y;
  if (b >= a.batch || st >= a.nstages - 1) return;
  const size_t rec = (size_t)b * a.nstages + st;
  const unsigned stat = scan::sto_prep_body<NV, NU, NS, SCAN_STO_PREP_NT>(
      a.grid[st], a.kkt + rec * SL.kkt.stride, a.ps + (rec + 1) * E::PS_STRIDE, a.scr + rec * scan::StoScratch<NV, NU, NS>::STRIDE, smem,
      threadIdx.x);
  if (stat && threadIdx.x == 0) atomicOr(&a.status[b], stat);
}

// five wavefronts per instance: a row of the step's products is shared by up to four neighbouring lanes of the first four, the fifth
// stores the previous grid point's results meanwhile (the bundle of a grid point is prefetched into registers: 14 doubles per lane for ANYmal)
constexpr int SCAN_STO_MAX_STAGES = 512;   // = scan::StoVecCfg::MAX_STAGES (grids beyond take the serial kernel)
constexpr int scan_sto_vec_nt(int) { return 320; }
template <int NV, int NU, int NS, int SCAN_STO_VEC_NT = scan_sto_vec_nt(NV)>
__global__ __launch_bounds__(SCAN_STO_VEC_NT) void scan_sto_vector_kernel(StoScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  const int b = a.first + blockIdx.x;
  if (b >= a.batch) return;
  const size_t inst = (size_t)b * a.nstages;
  const unsigned stat = scan::sto_vector_body<NV, NU, NS, SCAN_STO_VEC_NT>(
      a.grid, a.nstages, a.kkt + inst * SL.kkt.stride, a.ric + inst * SL.ric.stride, a.scr + inst * scan::StoScratch<NV, NU, NS>::STRIDE,
      a.max_dts0, smem, threadIdx.x, b == 0 ? a.prof : nullptr);
  if (stat) atomicOr(&a.status[b], stat);
}

// ---- forward recursion as a prefix scan of the closed-loop maps (riccati_scan_core.hpp) ------------------
struct FwdScanArgs {
  const double* kkt;
  const double* ric;
  double* dir;
  const double* dx0;  // [batch][nx] or nullptr (then dir[...][0].dx is used as given)
  const rtoc_grid* grid;
  const double* src;  // maps before this level [batch][nstages][EltLayout::STRIDE]
  double* dst;
  int nstages;
  int batch;
  int first;
  int dist;
};
constexpr int SCAN_FWD_NT = 256;

template <int NV, int NU, int NS>
__global__ __launch_bounds__(SCAN_FWD_NT) void fwd_scan_element_kernel(FwdScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using E = scan::EltLayout<NV>;
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr int NX = 2 * NV;
  const int st = blockIdx.x, b = a.first + blockIdx.y;
  if (b >= a.batch || st >= a.nstages - 1) return;
  const size_t rec = (size_t)b * a.nstages + st;
  double* dir0 = a.dir + (size_t)b * a.nstages * SL.dir.stride;
  const double* dx0 = a.dx0 ? a.dx0 + (size_t)b * NX : dir0 + SL.dir.off[RTOC_DIR_DX];
  scan::fwd_element_body<NV, NU, NS, SCAN_FWD_NT>(a.grid[st], st, a.kkt + rec * SL.kkt.stride,
                                                  a.ric + rec * SL.ric.stride, dx0, a.dst + rec * E::STRIDE, dir0,
                                                  smem, threadIdx.x);
}

template <int NV, int NU, int NS>
__global__ __launch_bounds__(SCAN_FWD_NT) void fwd_scan_combine_kernel(FwdScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using E = scan::EltLayout<NV>;
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  // maps i = d .. N-1 are still open before the level with distance d (those below are vectors already)
  const int i = a.dist + blockIdx.x, b = a.first + blockIdx.y;
  if (b >= a.batch || i >= a.nstages - 1) return;
  const int j = i - a.dist;
  const bool closed2 = j < a.dist;
  const size_t inst = (size_t)b * a.nstages;
  double* dirb = a.dir + inst * SL.dir.stride;
  scan::fwd_combine_body<NV, NU, SCAN_FWD_NT>(
      a.src + (inst + i) * E::STRIDE, a.src + (inst + j) * E::STRIDE,
      dirb + (size_t)(j + 1) * SL.dir.stride + SL.dir.off[RTOC_DIR_DX], closed2, a.dst + (inst + i) * E::STRIDE,
      dirb + (size_t)(i + 1) * SL.dir.stride + SL.dir.off[RTOC_DIR_DX], smem, threadIdx.x);
}

template <int NV, int NU, int NS>
__global__ __launch_bounds__(64) void fwd_scan_finish_kernel(FwdScanArgs a) {
  __shared__ double smem[2 * NV + 8];
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  const int st = blockIdx.x, b = a.first + blockIdx.y;
  if (b >= a.batch || st >= a.nstages) return;
  const size_t rec = (size_t)b * a.nstages + st;
  scan::fwd_finish_body<NV, NU, NS, 64>(a.grid[st], st == a.nstages - 1, a.ric + rec * SL.ric.stride,
                                        a.dir + rec * SL.dir.stride, smem, threadIdx.x);
}

}  // namespace rtoc

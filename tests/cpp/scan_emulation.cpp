// Host emulation of the horizon-scan workgroup bodies (robotoc_amd/csrc/riccati_scan_core.hpp) with
// one "thread" per workgroup: the same source the HIP kernels instantiate, compiled by g++, so that
// the algebra and the indexing are checked on the CPU against tests/scan_reference.py.
// TEST INFRASTRUCTURE ONLY: nothing in the product links or loads this file.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../robotoc_amd/csrc/riccati_scan_core.hpp"
#include "../../robotoc_amd/csrc/riccati_scan_sto.hpp"

using namespace rtoc::scan;

template <int NV, int NU, int NS>
static int run(const rtoc_grid* grid, int n, const double* kkt, double* ps, unsigned* stat) {
  using E = EltLayout<NV>;
  constexpr rtoc_layout SL = ScanLayout<NV, NU, NS>::make();
  const int kstride = SL.kkt.stride;
  std::vector<double> buf0((size_t)n * E::STRIDE, 0.0), buf1((size_t)n * E::STRIDE, 0.0);
  std::vector<double> smem_e(ElementCfg<NV, NU, NS>::LDS_DOUBLES), smem_c(CombineCfg<NV, 1>::LDS_DOUBLES);
  *stat = 0;
  for (int i = 0; i < n; ++i)
    *stat |= element_body<NV, NU, NS, 1>(grid[i], kkt + (size_t)i * kstride, buf0.data() + (size_t)i * E::STRIDE,
                                         ps + (size_t)i * E::PS_STRIDE, smem_e.data(), 0);
  double* src = buf0.data();
  double* dst = buf1.data();
  int levels = 0;
  for (int d = 1; d < n; d *= 2, ++levels) {
    for (int i = 0; i < n; ++i) {
      int j;
      bool closed2;
      if (!level_plan(n, d, i, &j, &closed2)) continue;
      const double* e1 = src + (size_t)i * E::STRIDE;
      const double* e2 = src + (size_t)j * E::STRIDE;
      const double* p2 = ps + (size_t)j * E::PS_STRIDE;
      for (int part = 0; part < 2; ++part)  // the two workgroups of a combination (grid z of the kernel)
        *stat |= combine_body<NV, 1>(e1, closed2 ? p2 + E::PS_P : e2 + E::OFF_J, closed2 ? p2 + E::PS_S : e2 + E::OFF_ETA,
                                     e2 + E::OFF_A, e2 + E::OFF_B, e2 + E::OFF_C, closed2, part,
                                     dst + (size_t)i * E::STRIDE, ps + (size_t)i * E::PS_STRIDE, smem_c.data(), 0);
    }
    double* t = src;
    src = dst;
    dst = t;
  }
  return levels;
}

// forward prefix scan of ONE instance: dir [n][dir stride] out
template <int NV, int NU, int NS>
static int run_fwd(const rtoc_grid* grid, int n, const double* kkt, const double* ric, const double* dx0, double* dir) {
  using E = EltLayout<NV>;
  constexpr rtoc_layout SL = ScanLayout<NV, NU, NS>::make();
  const int N = n - 1;
  std::vector<double> buf0((size_t)n * E::STRIDE, 0.0), buf1((size_t)n * E::STRIDE, 0.0);
  std::vector<double> smem(FwdCfg<NV, NU>::LDS_DOUBLES);
  for (int i = 0; i < N; ++i)
    fwd_element_body<NV, NU, NS, 1>(grid[i], i, kkt + (size_t)i * SL.kkt.stride, ric + (size_t)i * SL.ric.stride, dx0,
                                    buf0.data() + (size_t)i * E::STRIDE, dir, smem.data(), 0);
  double* src = buf0.data();
  double* dst = buf1.data();
  int levels = 0;
  for (int d = 1; d < N; d *= 2, ++levels) {
    for (int i = d; i < N; ++i) {
      const int j = i - d;
      fwd_combine_body<NV, NU, 1>(src + (size_t)i * E::STRIDE, src + (size_t)j * E::STRIDE,
                                  dir + (size_t)(j + 1) * SL.dir.stride + SL.dir.off[RTOC_DIR_DX], j < d,
                                  dst + (size_t)i * E::STRIDE,
                                  dir + (size_t)(i + 1) * SL.dir.stride + SL.dir.off[RTOC_DIR_DX], smem.data(), 0);
    }
    double* t = src;
    src = dst;
    dst = t;
  }
  for (int i = 0; i < n; ++i)
    fwd_finish_body<NV, NU, NS, 1>(grid[i], i == N, ric + (size_t)i * SL.ric.stride, dir + (size_t)i * SL.dir.stride,
                                   smem.data(), 0);
  return levels;
}

extern "C" int scan_emu_forward(int nv, int nu, int ns_max, const rtoc_grid* grid, int n, const double* kkt,
                                const double* ric, const double* dx0, double* dir) {
  if (nv == 18 && nu == 12 && ns_max == 12) return run_fwd<18, 12, 12>(grid, n, kkt, ric, dx0, dir);
  if (nv == 35 && nu == 29 && ns_max == 12) return run_fwd<35, 29, 12>(grid, n, kkt, ric, dx0, dir);
  if (nv == 32 && nu == 26 && ns_max == 12) return run_fwd<32, 26, 12>(grid, n, kkt, ric, dx0, dir);
  if (nv == 7 && nu == 7 && ns_max == 0) return run_fwd<7, 7, 0>(grid, n, kkt, ric, dx0, dir);
  return -1;
}

extern "C" int scan_emu_ps_stride(int nv) { return ((4 * nv * nv + 7) & ~7) + ((2 * nv + 7) & ~7); }

// kkt: [n][kkt stride] of ONE instance; ps: [n][ps stride] out (P | s of every grid point).
// Returns the number of combination levels, or -1 for unsupported dimensions.
extern "C" int scan_emu_backward(int nv, int nu, int ns_max, const rtoc_grid* grid, int n, const double* kkt,
                                 double* ps, unsigned* stat) {
  if (nv == 18 && nu == 12 && ns_max == 12) return run<18, 12, 12>(grid, n, kkt, ps, stat);
  if (nv == 35 && nu == 29 && ns_max == 12) return run<35, 29, 12>(grid, n, kkt, ps, stat);
  if (nv == 32 && nu == 26 && ns_max == 12) return run<32, 26, 12>(grid, n, kkt, ps, stat);
  if (nv == 7 && nu == 7 && ns_max == 0) return run<7, 7, 0>(grid, n, kkt, ps, stat);
  return -1;
}

// Grids with switching-time optimisation (riccati_scan_sto.hpp): the stage-parallel preparation of every grid point from the
// scan's value records `ps` (P+ first), then the serial vector pass of ONE instance.  ric: [n][ric stride], in: the matrix half
// (K, M of every grid point and the terminal s -- the caller takes them from the serial recursion), out: s, k, m and every STO
// quantity.  Returns the status bits, or -1 for unsupported dimensions.
template <int NV, int NU, int NS>
static int run_sto(const rtoc_grid* grid, int n, const double* kkt, const double* ps, double* ric, double max_dts0) {
  using E = EltLayout<NV>;
  using W = StoScratch<NV, NU, NS>;
  constexpr rtoc_layout SL = ScanLayout<NV, NU, NS>::make();
  std::vector<double> scr((size_t)n * W::STRIDE, 0.0), smem_p(StoPrepCfg<NV, NU, NS>::LDS_DOUBLES), smem_v(StoVecCfg<NV, NU, NS>::LDS_DOUBLES);
  unsigned stat = 0;
  for (int i = 0; i + 1 < n; ++i)
    stat |= sto_prep_body<NV, NU, NS, 1>(grid[i], kkt + (size_t)i * SL.kkt.stride, ps + (size_t)(i + 1) * E::PS_STRIDE,
                                         scr.data() + (size_t)i * W::STRIDE, smem_p.data(), 0);
  stat |= sto_vector_body<NV, NU, NS, 1>(grid, n, kkt, ric, scr.data(), max_dts0, smem_v.data(), 0);
  return (int)stat;
}

extern "C" int scan_emu_backward_sto(int nv, int nu, int ns_max, const rtoc_grid* grid, int n, const double* kkt, const double* ps,
                                     double* ric, double max_dts0) {
  if (nv == 18 && nu == 12 && ns_max == 12) return run_sto<18, 12, 12>(grid, n, kkt, ps, ric, max_dts0);
  if (nv == 35 && nu == 29 && ns_max == 12) return run_sto<35, 29, 12>(grid, n, kkt, ps, ric, max_dts0);
  if (nv == 32 && nu == 26 && ns_max == 12) return run_sto<32, 26, 12>(grid, n, kkt, ps, ric, max_dts0);
  if (nv == 7 && nu == 7 && ns_max == 0) return run_sto<7, 7, 0>(grid, n, kkt, ps, ric, max_dts0);
  return -1;
}

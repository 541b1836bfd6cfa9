// Timing probe of scan_combine_kernel: one combination level on random (valid) elements, with parts
// of the body compiled out (-DRTOC_SCAN_PROBE=bitmask: 1 no elimination loop, 2 no products after it,
// 4 no M product) to see where the time goes.  Results are wrong by construction when parts are off.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DNVP=18 -DRTOC_SCAN_PROBE=0 scan_probe.hip -o scan_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../robotoc_amd/csrc/riccati_scan.hpp"
#ifndef NVP
#define NVP 18
#endif
using namespace rtoc;

int main() {
  using E = scan::EltLayout<NVP>;
  constexpr int NX = 2 * NVP, n = 47;
  std::vector<double> elt((size_t)n * E::STRIDE), ps((size_t)n * E::PS_STRIDE);
  srand(1);
  auto rnd = [] { return 2.0 * rand() / RAND_MAX - 1.0; };
  std::vector<double> G(NX * NX);
  for (int s = 0; s < n; ++s) {
    double* e = elt.data() + (size_t)s * E::STRIDE;
    for (int i = 0; i < NX * NX; ++i) e[E::OFF_A + i] = 0.3 * rnd() + ((i % (NX + 1)) == 0 ? 1.0 : 0.0);
    for (int which = 0; which < 2; ++which) {
      for (auto& g : G) g = rnd();
      double* dst = e + (which ? E::OFF_J : E::OFF_C);
      for (int i = 0; i < NX; ++i)
        for (int j = 0; j < NX; ++j) {
          double acc = 0;
          for (int k = 0; k < NX; ++k) acc += G[i + k * NX] * G[j + k * NX];
          dst[i + j * NX] = acc * (which ? 1.0 : 0.01);
        }
    }
    for (int i = 0; i < NX; ++i) e[E::OFF_B + i] = rnd(), e[E::OFF_ETA + i] = rnd();
    double* p = ps.data() + (size_t)s * E::PS_STRIDE;
    for (int i = 0; i < NX * NX; ++i) p[E::PS_P + i] = e[E::OFF_J + i];
    for (int i = 0; i < NX; ++i) p[E::PS_S + i] = rnd();
  }
  double *d0, *d1, *dps;
  uint32_t* dstat;
  hipMalloc(&d0, elt.size() * 8);
  hipMalloc(&d1, elt.size() * 8);
  hipMalloc(&dps, ps.size() * 8);
  hipMalloc(&dstat, 4);
  hipMemset(dstat, 0, 4);
  hipMemcpy(d0, elt.data(), elt.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dps, ps.data(), ps.size() * 8, hipMemcpyHostToDevice);
  auto k = scan_combine_kernel<NVP>;
  constexpr int SCAN_NT = scan_comb_nt(NVP);
  const int lds = scan::CombineCfg<NVP, SCAN_NT>::LDS_BYTES;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  ScanArgs a = {};
  a.status = dstat;
  a.src = d0;
  a.dst = d1;
  a.ps = dps;
  a.nstages = n;
  a.batch = 1;
  a.first = 0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int d : {1, 32}) {  // d = 1: all open; d = 32: right operands closed
    a.dist = d;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(n - d, 1, 2), dim3(SCAN_NT), lds, 0, a);
    hipEventRecord(e0, 0);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(n - d, 1, 2), dim3(SCAN_NT), lds, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t st;
    hipMemcpy(&st, dstat, 4, hipMemcpyDeviceToHost);
    printf("NV=%d probe=%d lds=%d d=%d: %.2f us per level (status %u)\n", NVP, RTOC_SCAN_PROBE, lds, d,
           1000.0 * ms / reps, st);
  }
  return 0;
}

// How many single-wave workgroups does a CU of this device hold, by dynamic LDS size and register budget?
// Every workgroup spins for a fixed number of cycles; 256 CUs x 16 workgroups are launched, so the elapsed time is
// 16 / (resident workgroups per CU) spin periods.      hipcc --offload-arch=gfx950 -O3 occupancy_probe.hip -o occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ double smem[];
template <int REGS>
__global__ __launch_bounds__(64, (REGS > 128 ? 2 : 4)) void spin(long long cycles, double* out) {
  double acc[REGS / 2];
#pragma unroll
  for (int i = 0; i < REGS / 2; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < REGS / 2; ++i) acc[i] = __builtin_fma(acc[i], 1.0000001, 1e-9);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < REGS / 2; ++i) s += acc[i];
  smem[threadIdx.x] = s;
  if (s == 12345.678) out[blockIdx.x] = smem[(threadIdx.x + 1) & 63];
}
template <int REGS>
void run(const char* name, int cus) {
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)spin<REGS>);
  printf("%s: numRegs %d\n", name, fa.numRegs);
  double* out;
  hipMalloc(&out, 1 << 20);
  const int ldss[] = {1024, 8192, 16384, 18432, 20192, 20480, 21504, 24576, 26624, 32768, 40960};
  for (int lds : ldss) {
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)spin<REGS>, 64, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<REGS>, dim3(cus * 16), dim3(64), lds, 0, 200000LL, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<REGS>, dim3(cus * 16), dim3(64), lds, 0, 200000LL, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  lds %6d B: runtime says %2d workgroups / CU; 16 per CU took %.3f ms\n", lds, occ, ms);
  }
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.name, p.multiProcessorCount, p.sharedMemPerBlock,
         p.maxSharedMemoryPerMultiProcessor);
  run<32>("32 registers", p.multiProcessorCount);
  run<200>("200 registers", p.multiProcessorCount);
  return 0;
}

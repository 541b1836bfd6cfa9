// Where do the two waves of a 128-thread workgroup land?  Prints (cu, simd, slot) per wave for the
// launch shape of the role-split backward kernel (39.3 KB LDS, 4 workgroups per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(128, 2) void probe(unsigned* out, int spin) {
  extern __shared__ double smem[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
  smem[threadIdx.x] = x;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
  if (smem[(threadIdx.x + 1) & 127] == 12345.0) out[0] = 0;
}
int main() {
  const int nb = 4096;
  unsigned* d;
  hipMalloc(&d, nb * 4 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40256);
  probe<<<nb, 128, 40256>>>(d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 4);
  hipMemcpy(h.data(), d, nb * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
  // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
  std::map<unsigned, int> pair_hist;
  for (int b = 0; b < nb; ++b) {
    unsigned h0 = h[b * 4], h1 = h[b * 4 + 2];
    unsigned s0 = (h0 >> 4) & 3, s1 = (h1 >> 4) & 3, w0 = h0 & 15, w1 = h1 & 15;
    if (b < 24)
      printf("block %4d: xcc %u se %u cu %2u | wave0 simd %u slot %u | wave1 simd %u slot %u\n", b,
             h[b * 4 + 1] & 15, (h0 >> 13) & 7, (h0 >> 8) & 15, s0, w0, s1, w1);
    pair_hist[(s0 << 12) | (w0 << 8) | (s1 << 4) | w1]++;
  }
  printf("histogram of (simd0,slot0 | simd1,slot1):\n");
  for (auto& kv : pair_hist)
    printf("  wave0 simd %u slot %u | wave1 simd %u slot %u : %d\n", (kv.first >> 12) & 15,
           (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15, kv.second);
  return 0;
}

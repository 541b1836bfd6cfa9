// Issue interval of v_mfma_f64_16x16x4_f64 on gfx950 in the patterns the backward kernel uses:
// independent accumulators with shared operands, operands arriving from LDS, one wave or two per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
__global__ __launch_bounds__(512) void probe(long long* out, int n, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ double lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 1e-3 * i;
  __syncthreads();
  double a0 = lane * 1e-3, a1 = a0 + 1, a2 = a0 + 2, b0 = 1.0 + lane * 1e-4, b1 = b0 + 1, b2 = b0 + 2;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0, c8 = c0;
  long long t0 = __builtin_readcyclecounter();
  if (mode == 0) {  // 4 independent accumulators, same operands
    for (int i = 0; i < n; ++i) { MF(c0, a0, b0); MF(c1, a0, b0); MF(c2, a0, b0); MF(c3, a0, b0); }
  } else if (mode == 1) {  // 9 accumulators, 3 x 3 operands (the P+ A step)
    for (int i = 0; i < n; ++i) {
      MF(c0, a0, b0); MF(c1, a0, b1); MF(c2, a0, b2); MF(c3, a1, b0); MF(c4, a1, b1); MF(c5, a1, b2);
      MF(c6, a2, b0); MF(c7, a2, b1); MF(c8, a2, b2);
    }
  } else if (mode == 2) {  // same, operands re-read from LDS each step, software-pipelined by hand
    const double* p = lds + lane;
    double na0 = p[0], na1 = p[64], na2 = p[128], nb0 = p[192], nb1 = p[256], nb2 = p[320];
    for (int i = 0; i < n; ++i) {
      a0 = na0; a1 = na1; a2 = na2; b0 = nb0; b1 = nb1; b2 = nb2;
      const int o = (i & 7) * 8;
      na0 = p[o]; na1 = p[o + 64]; na2 = p[o + 128]; nb0 = p[o + 192]; nb1 = p[o + 256]; nb2 = p[o + 320];
      __builtin_amdgcn_sched_barrier(0);
      MF(c0, a0, b0); MF(c1, a0, b1); MF(c2, a0, b2); MF(c3, a1, b0); MF(c4, a1, b1); MF(c5, a1, b2);
      MF(c6, a2, b0); MF(c7, a2, b1); MF(c8, a2, b2);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (mode == 3) {  // one accumulator (dependent chain)
    for (int i = 0; i < n; ++i) { MF(c0, a0, b0); MF(c0, a0, b0); MF(c0, a0, b0); MF(c0, a0, b0); }
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[wave] = t1 - t0;
  if (c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[0] + c6[0] + c7[0] + c8[0] == 1234.5) out[100] = 1;
}
int main() {
  long long* d;
  hipMalloc(&d, 128 * 8);
  const int n = 2048;
  const char* nm[4] = {"4 accumulators, shared operands", "9 accumulators, 3x3 operands", "9 accumulators, operands from LDS (pipelined)", "1 accumulator (dependent)"};
  const int per[4] = {4, 9, 9, 4};
  for (int threads : {64, 256, 512})
    for (int mode = 0; mode < 4; ++mode) {
      hipMemset(d, 0, 128 * 8);
      probe<<<1, threads>>>(d, n, mode);
      hipDeviceSynchronize();
      long long h[8];
      hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      printf("%3d threads (%d wave(s)/SIMD) | %-46s | %.1f cycles per MFMA (wave 0)\n", threads, threads > 256 ? 2 : 1, nm[mode], (double)h[0] / (per[mode] * (double)n));
    }
  return 0;
}

// Issue cost of v_mfma_f64_4x4x4 (4 independent 4x4x4 blocks per instruction) against
// v_mfma_f64_16x16x4 and v_fma_f64 on gfx950: one wave, independent accumulators (throughput)
// and one accumulator (latency).  Cycles from s_memtime scaled like pipe_share_probe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void probe(long long* out, int n) {
  const int lane = threadIdx.x;
  double a = lane * 1e-3, b = 1.0 + lane * 1e-4;
  {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[0] = t1 - t0;
    if (c0[0] + c1[1] + c2[2] + c3[3] == 1234.5) out[100] = 1;
  }
  {
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[1] = t1 - t0;
    if (c0 + c1 + c2 + c3 == 1234.5) out[100] = 1;
  }
  {
    double c0 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[2] = t1 - t0;
    if (c0 == 1234.5) out[100] = 1;
  }
  {
    double c0 = 1, c1 = 2, c2 = 3, c3 = 4;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_fma(a, b, c0); c1 = __builtin_fma(a, b, c1);
      c2 = __builtin_fma(a, b, c2); c3 = __builtin_fma(a, b, c3);
      asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[3] = t1 - t0;
    if (c0 + c1 + c2 + c3 == 1234.5) out[100] = 1;
  }
  {
    d4 c0 = {0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[4] = t1 - t0;
    if (c0[0] == 1234.5) out[100] = 1;
  }
}
int main() {
  long long* d;
  hipMalloc(&d, 128 * 8);
  const int n = 4096;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(d, 0, 128 * 8);
    probe<<<1, 64>>>(d, n);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("16x16x4 indep %.1f | 4x4x4 indep %.1f | 4x4x4 dependent %.1f | v_fma_f64 indep %.1f | 16x16x4 dependent %.1f  (cycles per instruction)\n",
           h[0] / (4.0 * n), h[1] / (4.0 * n), h[2] / (4.0 * n), h[3] / (4.0 * n), h[4] / (4.0 * n));
  }
  return 0;
}

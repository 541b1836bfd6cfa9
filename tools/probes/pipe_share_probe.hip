// Does a dependent chain of VALU ops slow down when ANOTHER wave on the same SIMD streams f64 MFMAs?
// One workgroup of 8 waves: waves 0-3 ("matrix") issue mfma_f64_16x16x4 in a chosen pattern,
// waves 4-7 ("vector", same SIMDs) run a dependent f64 FMA chain of fixed length and time it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define NOPS do { if (NOPK == 1) asm volatile("s_nop 3"); if (NOPK == 2) asm volatile("s_nop 7"); if (NOPK == 3) asm volatile("s_nop 11"); if (NOPK == 4) asm volatile("s_nop 13"); if (NOPK == 5) asm volatile("s_nop 15"); if (NOPK == 6) asm volatile("s_nop 15\ns_nop 3"); if (NOPK == 7) asm volatile("s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7"); } while (0)
#define MFMA(acc) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0)
template <int NOPK>
__global__ __launch_bounds__(512) void probe(long long* out, int mode, int n, int prio, int swap) {
  int wave = threadIdx.x >> 6; const int lane = threadIdx.x & 63;
  if (swap) wave ^= 4;  // matrix role on the YOUNGER waves
  __shared__ double lds[512];
  lds[threadIdx.x] = threadIdx.x * 0.001;
  __syncthreads();
  if (wave < 4) {
    long long tm0 = __builtin_readcyclecounter();
    d4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    double a = lane * 1e-3, b = 1.0 + lane * 1e-4;
    if (mode == 1) {
      for (int i = 0; i < n; ++i) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
    } else if (mode == 2) {
      for (int i = 0; i < n; ++i) { MFMA(acc0); MFMA(acc0); MFMA(acc0); MFMA(acc0); }
    } else if (mode == 3) {
      for (int i = 0; i < n; ++i) {
        MFMA(acc0); NOPS; MFMA(acc1); NOPS; MFMA(acc2); NOPS; MFMA(acc3); NOPS;
      }
    } else if (mode == 4) {
      for (int i = 0; i < n; ++i) {
        MFMA(acc0); asm volatile("s_sleep 0"); MFMA(acc1); asm volatile("s_sleep 0");
        MFMA(acc2); asm volatile("s_sleep 0"); MFMA(acc3); asm volatile("s_sleep 0");
      }
    } else if (mode == 5) {  // an LDS read (+wait) between MFMAs, like the real operand stream
      int idx = lane;
      for (int i = 0; i < n; ++i) {
        MFMA(acc0); a += lds[(idx++) & 511]; MFMA(acc1); b += lds[(idx++) & 511];
        MFMA(acc2); a += lds[(idx++) & 511]; MFMA(acc3); b += lds[(idx++) & 511];
      }
    }
    if (lane == 0) out[wave] = __builtin_readcyclecounter() - tm0;
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 1234.5) out[100] = 1;
  } else {
    if (prio) __builtin_amdgcn_s_setprio(3);
    long long t0 = __builtin_readcyclecounter();
    double x = 1.0 + lane * 1e-6, y = 0.999999;
#pragma unroll 8
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, y, 1e-9);
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[wave] = t1 - t0;
    if (x == 1234.5678) out[101] = 1;
  }
}
int main() {
  long long* d;
  hipMalloc(&d, 128 * 8);
  const int n = 4096;
  auto run = [&](auto kern, const char* nm) {
    for (int prio = 0; prio < 2; ++prio) {
      hipMemset(d, 0, 128 * 8);
      kern<<<1, 512>>>(d, 3, n, prio, 0);
      hipDeviceSynchronize();
      long long h[8];
      hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      printf("nop %-14s prio %d | vector chain %.1f cycles/fma | matrix %.1f cycles/mfma\n", nm, prio, (double)h[4] / n, (double)h[0] / (4.0 * n));
    }
  };
  run(probe<0>, "none"); run(probe<1>, "s_nop 3"); run(probe<2>, "s_nop 7"); run(probe<3>, "s_nop 11");
  run(probe<4>, "s_nop 13"); run(probe<5>, "s_nop 15"); run(probe<6>, "s_nop 15+3"); run(probe<7>, "6 x s_nop 7");
  return 0;
}

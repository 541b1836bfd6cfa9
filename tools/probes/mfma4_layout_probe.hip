// Lane layout of v_mfma_f64_4x4x4_4b_f64 (__builtin_amdgcn_mfma_f64_4x4x4f64: 4 independent 4x4x4 products per instruction, one
// f64 of A, B and the accumulator per lane), found by one-hot probing: workgroup (p, q) feeds a = e_p, b = e_q and records the
// output lanes that become 1, i.e. the products a[p] b[q] every output lane sums.
// hipcc --offload-arch=gfx950 -O2 mfma4_layout_probe.hip -o mfma4_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* hit) {
  const int l = threadIdx.x, p = blockIdx.x, q = blockIdx.y;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == p ? 1.0 : 0.0, l == q ? 1.0 : 0.0, 0.0, 0, 0, 0);
  const unsigned long long m = __ballot(d != 0.0);
  if (l == 0) hit[p * 64 + q] = m;
}
int main() {
  static unsigned long long h[4096];
  unsigned long long* d;
  if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return 1;
  k<<<dim3(64, 64), 64>>>(d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int l = 0; l < 64; ++l) {
    printf("out lane %2d sums a[p] b[q] over (p,q) =", l);
    for (int p = 0; p < 64; ++p)
      for (int q = 0; q < 64; ++q)
        if ((h[p * 64 + q] >> l) & 1ull) printf(" (%d,%d)", p, q);
    printf("\n");
  }
  return 0;
}

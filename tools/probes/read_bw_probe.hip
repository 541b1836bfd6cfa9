// What read bandwidth does the access pattern of riccati_forward_kernel admit on this box?  4096 wavefronts (16 per CU, the
// forward kernel's launch), each reading 1.2 MB (1216 KB):
//   front    : the waves sweep memory together (chunk = wave + k * 4096), 1 KB per instruction (16 B / lane) -- the ideal
//   streams  : every wave walks its OWN contiguous 1.2 MB region (= the records of one OCP instance), 1 KB per instruction
//   rows288  : own region, 288 B per instruction (36 lanes x 8 B: one column of Fxx / P per load, the forward kernel's)
// with U loads in flight per wave.  hipcc --offload-arch=gfx950 -O3 read_bw_probe.hip -o read_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr size_t REGION = 1216 * 1024;  // bytes per wave
constexpr int WAVES = 4096;

template <int U, bool FRONT>
__global__ __launch_bounds__(64) void read16(const char* __restrict__ base, double* out) {
  const int w = blockIdx.x, lane = threadIdx.x;
  constexpr int CH = REGION / 1024;  // 1 KB chunks per wave
  double acc = 0.0;
  for (int c = 0; c + U <= CH; c += U) {
    d2 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const size_t chunk = FRONT ? (size_t)w + (size_t)(c + k) * WAVES : (size_t)w * CH + (c + k);
      v[k] = *reinterpret_cast<const d2*>(base + chunk * 1024 + lane * 16);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) acc += v[k][0] + v[k][1];
  }
  if (acc == 1234.5) out[w] = acc;
}

template <int U>
__global__ __launch_bounds__(64) void read288(const char* __restrict__ base, double* out) {
  const int w = blockIdx.x, lane = threadIdx.x;
  constexpr int CH = REGION / 288;
  const char* p = base + (size_t)w * REGION + (lane < 36 ? lane : 0) * 8;
  double acc = 0.0;
  for (int c = 0; c + U <= CH; c += U) {
    double v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = *reinterpret_cast<const double*>(p + (size_t)(c + k) * 288);
#pragma unroll
    for (int k = 0; k < U; ++k) acc += v[k];
  }
  if (acc == 1234.5) out[w] = acc;
}

template <class F>
static void run(const char* name, double bytes, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f, sum = 0.f;
  const int reps = 6;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
    sum += ms;
  }
  printf("%-28s %7.3f ms (best %7.3f)  %6.2f TB/s (best %6.2f)\n", name, sum / reps, best, bytes / (sum / reps) * 1e-9,
         bytes / best * 1e-9);
}

int main() {
  const size_t total = REGION * WAVES;
  char* buf;
  double* out;
  if (hipMalloc(&buf, total) != hipSuccess || hipMalloc(&out, WAVES * 8) != hipSuccess) return 1;
  hipMemset(buf, 0, total);
  const double B = (double)total;
#define R16(U, FRONT, NAME) run(NAME, B, [&] { read16<U, FRONT><<<WAVES, 64>>>(buf, out); })
#define R288(U, NAME) run(NAME, B, [&] { read288<U><<<WAVES, 64>>>(buf, out); })
  printf("%d waves x %zu KB = %.2f GB per launch\n", WAVES, REGION / 1024, B * 1e-9);
  R16(4, true, "front   1KB/instr  U=4");
  R16(8, true, "front   1KB/instr  U=8");
  R16(16, true, "front   1KB/instr  U=16");
  R16(4, false, "streams 1KB/instr  U=4");
  R16(8, false, "streams 1KB/instr  U=8");
  R16(16, false, "streams 1KB/instr  U=16");
  R16(32, false, "streams 1KB/instr  U=32");
  R288(8, "rows288 288B/instr U=8");
  R288(16, "rows288 288B/instr U=16");
  R288(32, "rows288 288B/instr U=32");
  R288(48, "rows288 288B/instr U=48");
  return 0;
}

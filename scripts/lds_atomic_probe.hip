// LDS atomic-add throughput on gfx950 by type and address pattern (what the N4 VJP scatters can expect from
// privatising their accumulation in LDS):  hipcc -O3 --offload-arch=gfx950 scripts/lds_atomic_probe.hip -o build/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <class T, int PATTERN>
__global__ __launch_bounds__(512) void probe(T* out, int iters, unsigned long long* cyc) {
  __shared__ T plane[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) plane[i] = T(0);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned a = 0;
  const unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    unsigned idx;
    if (PATTERN == 0) idx = (wv * 64 + lane + i * 517) & 8191;                 // consecutive lanes, consecutive words
    else if (PATTERN == 1) idx = (wv * 64 + (lane >> 1) + i * 517) & 8191;      // pairs of lanes share a word
    else if (PATTERN == 2) idx = ((lane & 7) * 160 + (lane >> 3) + i * 37 + wv * 11) & 8191;  // 8 rows of a 160-wide plane
    else idx = ((lane * 2654435761u + i * 40503u + wv) >> 7) & 8191;            // scattered
    atomicAdd(&plane[idx], T(1));
    a += idx;
  }
  const unsigned long long t1 = clock64();
  __syncthreads();
  if (a == 0xffffffffu) out[0] = plane[0];
  out[blockIdx.x * 512 + threadIdx.x] = plane[threadIdx.x];
  if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

template <class T, int PATTERN>
void run(const char* name, T* out, unsigned long long* cyc, int grid) {
  const int iters = 4000;
  probe<T, PATTERN><<<grid, 512>>>(out, iters, cyc);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  probe<T, PATTERN><<<grid, 512>>>(out, iters, cyc);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.3f ms -> %.1f G lane-adds/s chip, %.2f lane-adds per cycle per CU (2.4 GHz, %d CUs)\n", name, ms,
         (double)grid * 512 * iters / (ms * 1e-3) / 1e9, (double)grid * 512 * iters / (ms * 1e-3) / 2.4e9 / 256, 256);
}

int main() {
  const int grid = 256 * 4;
  void* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)grid * 512 * 8);
  (void)hipMalloc(&cyc, (size_t)grid * 8 * 8);
  run<float, 0>("f32 consecutive", (float*)out, cyc, grid);
  run<float, 1>("f32 pairs share", (float*)out, cyc, grid);
  run<float, 2>("f32 8 rows x 160", (float*)out, cyc, grid);
  run<float, 3>("f32 scattered", (float*)out, cyc, grid);
  run<unsigned, 0>("u32 consecutive", (unsigned*)out, cyc, grid);
  run<unsigned, 1>("u32 pairs share", (unsigned*)out, cyc, grid);
  run<unsigned, 2>("u32 8 rows x 160", (unsigned*)out, cyc, grid);
  run<unsigned, 3>("u32 scattered", (unsigned*)out, cyc, grid);
  run<unsigned long long, 0>("u64 consecutive", (unsigned long long*)out, cyc, grid);
  run<unsigned long long, 3>("u64 scattered", (unsigned long long*)out, cyc, grid);
  return 0;
}

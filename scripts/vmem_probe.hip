// Vector-memory issue-rate probe for gfx950: how many dword wave-loads per cycle does one CU sustain from its L1
// when the 64 lanes of a load fall into 1, 2, 4, 8 or 16 different 128-byte lines?  (The ceiling the gather kernels of
// the cost volume / epipolar lookup -- N3, N4 -- are priced against: they are bound by the texture-address path, not
// by HBM.)
//   hipcc -O3 --offload-arch=gfx950 scripts/vmem_probe.hip -o build/vmem_probe && build/vmem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// every lane loads `iters` x 8 independent dwords from a 16 KB table (L1-resident); lane l of a load reads line
// (l % lines) (+ a rotating offset), word (l / lines) of it
__global__ __launch_bounds__(256) void probe(const float* tab, float* out, int iters, int lines, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  const unsigned base = (unsigned)((lane % lines) * 32 + (lane / lines) % 32);  // in floats; a line = 32 floats
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned o = (base + (unsigned)((i * 8 + j) & 63) * 64u) & 4095u;  // stays inside the 16 KB table
      acc[j] += tab[o];
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float *tab, *out;
  unsigned long long* cyc;
  const int waves_per_cu = 8, grid = cus * waves_per_cu / 4, iters = 2000;
  (void)hipMalloc(&tab, 4096 * 4);
  (void)hipMemset(tab, 0, 4096 * 4);
  (void)hipMalloc(&out, (size_t)grid * 256 * 4);
  (void)hipMalloc(&cyc, (size_t)grid * 4 * 8);
  printf("CUs %d, %d waves per CU, %d dword wave-loads per wave\n", cus, waves_per_cu, iters * 8);
  for (int lines : {1, 2, 4, 8, 16, 32, 64}) {
    for (int rep = 0; rep < 2; ++rep) probe<<<grid, 256>>>(tab, out, iters, lines, cyc);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<<<grid, 256>>>(tab, out, iters, lines, cyc);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double loads = 1.0 * iters * 8;  // per wave
    printf("lanes over %2d lines: %.3f ms; per CU %.2f cycles per dword wave-load (a wave: %.1f cycles per load, %d waves per CU); "
           "chip %.1f G wave-loads/s\n", lines, ms, med / loads / waves_per_cu, med / loads, waves_per_cu,
           loads * grid * 4 / (ms * 1e-3) / 1e9);
  }
  return 0;
}

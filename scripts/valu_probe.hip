// VALU issue-rate probe for gfx950: how many cycles does one SIMD need per wave64 VALU instruction
// when 1, 2, 3, 4 or 8 waves share it?  (Decides whether the marching kernels are bound by a SIMD's VALU
// throughput or by each wave's own issue cadence.)
//   hipcc -O3 --offload-arch=gfx950 scripts/valu_probe.hip -o scripts/valu_probe && scripts/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_add_f32 with DPP (wave_shr), 3: mix fma + s_ ops (not used)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* cyc, unsigned long long* real) {
  float a[8];
  f2 p[8];
  const float x = 1.0f + 1e-7f * threadIdx.x, y = 1e-9f;
  const f2 x2 = {x, x}, y2 = {y, y};
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = threadIdx.x * 0.001f + j; p[j] = (f2){a[j], a[j] + 1.f}; }
  __syncthreads();
  const unsigned long long r0 = wall_clock64();
  const unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(x), "v"(y));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(x2), "v"(y2));
        if (MODE == 2) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(x));
        if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(x2));
      }
    }
  }
  const unsigned long long t1 = clock64();
  const unsigned long long r1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j] + p[j].x + p[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    cyc[w] = t1 - t0;
    real[w] = r1 - r0;
  }
}

template <int MODE>
void run(const char* name, int cus, int wps) {
  const int iters = 4000, instr = iters * 32;
  const int grid = cus * wps;
  float* out; unsigned long long *cyc, *real;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipMalloc(&cyc, (size_t)grid * 4 * 8);
  hipMalloc(&real, (size_t)grid * 4 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) probe<MODE><<<grid, 256>>>(out, iters, cyc, real);
  hipEventRecord(e0);
  probe<MODE><<<grid, 256>>>(out, iters, cyc, real);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * 4), hr(grid * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), real, hr.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  std::sort(hr.begin(), hr.end());
  const double med = (double)h[h.size() / 2], medr = (double)hr[hr.size() / 2];
  // clock64 = s_memtime ticks; wall_clock64 = 100 MHz
  printf("%-14s waves/SIMD %d: kernel %.3f ms; per wave: %.2f memtime-ticks/instr (median), memtime/realtime = %.2f ticks per 10ns;"
         " SIMD throughput %.3f instr per tick; wall: %.2f ns per instr per SIMD\n",
         name, wps, ms, med / instr, med / medr, wps * (double)instr / med, ms * 1e6 / ((double)instr * wps));
  hipFree(out); hipFree(cyc); hipFree(real);
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  printf("CUs %d\n", cus);
  for (int wps : {1, 2, 3, 4, 8}) run<0>("v_fma_f32", cus, wps);
  for (int wps : {1, 2, 3, 4, 8}) run<1>("v_pk_fma_f32", cus, wps);
  for (int wps : {1, 2, 4}) run<2>("v_add_f32_dpp", cus, wps);
  for (int wps : {1, 2, 4}) run<3>("v_pk_mul_f32", cus, wps);
  return 0;
}

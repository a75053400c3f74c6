// Price of the cross-lane forms a horizontal 3-sum could use, at the marching kernels' occupancy (2 waves per SIMD, every
// CU busy) -- VERDICT r03 item 2(i): is v_add_f32_dpp wave_shr:1 (4.3 cycles in scripts/valu_probe.hip) the cheapest way?
//   hipcc -O3 --offload-arch=gfx950 scripts/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* cyc) {
  float a[8];
  __shared__ float lds[8][260];
  const float x = 1.0f + 1e-7f * threadIdx.x;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = threadIdx.x * 0.001f + j; lds[j][threadIdx.x + 1] = a[j]; }
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(x));
        if (MODE == 1) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(x));
        if (MODE == 2) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(x));
        if (MODE == 3) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(x));
        if (MODE == 4) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(x));
        if (MODE == 5) asm volatile("v_fmac_f32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(x));
        if (MODE == 6) asm volatile("v_add_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[j]) : "v"(x));
        if (MODE == 7) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(x));
      }
    }
    if (MODE == 8) {  // the same 32 values through LDS: write own, read both neighbours, two adds each (per 8 values: 8 w, 8 r2, 16 add)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int j = 0; j < 8; ++j) lds[j][threadIdx.x + 1] = a[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (lds[j][threadIdx.x] + a[j]) + lds[j][threadIdx.x + 2];
      }
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int cus, int wps) {
  const int iters = 2000, instr = iters * 32;
  const int grid = cus * wps;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipMalloc(&cyc, (size_t)grid * 4 * 8);
  for (int rep = 0; rep < 3; ++rep) probe<MODE><<<grid, 256>>>(out, iters, cyc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  printf("%-44s waves/SIMD %d: %.2f ticks per value-op per wave; SIMD: %.2f cycles per value-op\n", name, wps, med / instr,
         med / instr / wps);
  hipFree(out); hipFree(cyc);
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  for (int wps : {1, 2}) {
    run<0>("v_add_f32 (plain)", cus, wps);
    run<1>("v_add_f32_dpp wave_shr:1", cus, wps);
    run<7>("v_add_f32_dpp wave_shl:1", cus, wps);
    run<2>("v_add_f32_dpp row_shr:1", cus, wps);
    run<3>("v_add_f32_dpp quad_perm", cus, wps);
    run<4>("v_mov_b32_dpp wave_shr:1", cus, wps);
    run<5>("v_fmac_f32_dpp wave_shr:1", cus, wps);
    run<6>("v_add_f32_dpp row_bcast:15 row_mask:0xa", cus, wps);
    run<8>("3-sum via LDS (per value: 1 write, 2 reads, 2 adds)", cus, wps);
  }
  return 0;
}

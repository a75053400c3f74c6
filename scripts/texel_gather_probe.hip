// What does ONE texel gather of the marching passes cost the CU's vector-memory path?  A lane loads N dwords (N = 1, 2, 3, 4:
// global_load_dword .. dwordx4) at byte offset lane * STRIDE (+ a rotating base) from an L1-resident table -- for N = 3,
// STRIDE = 12 that is exactly the bilinear tap of the marching kernels on a locally translational warp: 64 neighbouring
// (B,H,W,3) texels, 768 bytes over 6-7 lines.  8 waves per CU, 8 independent loads in flight per wave.
//   hipcc -O3 --offload-arch=gfx950 scripts/texel_gather_probe.hip -o /tmp/tgp && /tmp/tgp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f3 __attribute__((ext_vector_type(3), aligned(4)));  // dwordx3 at 4-byte alignment, as the kernels' texel_t
template <int N> struct V;
template <> struct V<1> { typedef float t; };
template <> struct V<2> { typedef float2 t; };
template <> struct V<3> { typedef f3 t; };
template <> struct V<4> { typedef float4 t; };
template <int N> __device__ float first(typename V<N>::t v);
template <> __device__ float first<1>(float v) { return v; }
template <> __device__ float first<2>(float2 v) { return v.x + v.y; }
template <> __device__ float first<3>(f3 v) { return v.x + v.y + v.z; }
template <> __device__ float first<4>(float4 v) { return v.x + v.y + v.z + v.w; }

template <int N>
__global__ __launch_bounds__(256) void probe(const char* tab, float* out, int iters, int stride, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  const unsigned base = (unsigned)lane * (unsigned)stride;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // rotating base in multiples of 48 bytes (keeps 4-, 8-, 12- and 16-byte elements aligned); table 16 KB + slack
      const unsigned o = base + (unsigned)(((i * 8 + j) * 7) & 127) * 48u;
      typename V<N>::t v;
      v = *reinterpret_cast<const typename V<N>::t*>(tab + o);
      acc[j] += first<N>(v);
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int N>
static void run(int stride, int waves_per_cu, int cus, const char* tab, float* out, unsigned long long* cyc) {
  const int grid = cus * waves_per_cu / 4, iters = 2000;
  for (int rep = 0; rep < 2; ++rep) probe<N><<<grid, 256>>>(tab, out, iters, stride, cyc);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  probe<N><<<grid, 256>>>(tab, out, iters, stride, cyc);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * 4);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2], loads = 2000.0 * 8;
  const int span = 63 * stride + 4 * N;
  printf("dwordx%d, lane stride %3d B (a wave-load spans %4d B = %.1f lines), %d waves/CU: %.3f ms; per CU %6.2f cycles per wave-load; "
         "%.1f B/cycle/CU useful\n", N, stride, span, span / 128.0, waves_per_cu, ms, med / loads / waves_per_cu,
         64.0 * 4 * N / (med / loads / waves_per_cu));
}

int main() {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  char* tab; float* out; unsigned long long* cyc;
  (void)hipMalloc(&tab, 64 * 1024);
  (void)hipMemset(tab, 0, 64 * 1024);
  (void)hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
  (void)hipMalloc(&cyc, (size_t)cus * 4 * 4 * 8);
  for (int wpc : {8, 4}) {
    run<1>(4, wpc, cus, tab, out, cyc);
    run<1>(12, wpc, cus, tab, out, cyc);
    run<2>(8, wpc, cus, tab, out, cyc);
    run<3>(12, wpc, cus, tab, out, cyc);
    run<4>(16, wpc, cus, tab, out, cyc);
    run<2>(16, wpc, cus, tab, out, cyc);
    run<3>(24, wpc, cus, tab, out, cyc);
    run<4>(32, wpc, cus, tab, out, cyc);
  }
  return 0;
}

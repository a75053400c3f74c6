// Materialising warp (forward + backward) and the granular geometry operators behind the
// reference's module-level API: disp_to_depth, BackprojectDepth, Project3D, F.grid_sample.
//
//   mal_warp_fwd/bwd      Trainer.generate_images_pred       manydepth/trainer.py:1093-1125
//   mal_disp_to_depth     disp_to_depth                       manydepth/layers.py:14-23
//   mal_backproject       BackprojectDepth.forward            manydepth/layers.py:163-168
//   mal_project3d         Project3D.forward                   manydepth/layers.py:184-199, dualrefine/layers.py:216-226
//   mal_grid_sample       F.grid_sample(bilinear, border)     manydepth/trainer.py:1122-1125
//
// All of these are one-thread-per-pixel streaming kernels: wave lanes walk consecutive
// pixels of a row, so plane reads/writes are 256-byte coalesced pieces and the four
// bilinear taps of neighbouring lanes fall in the same or adjacent 128-byte lines.
// The pose gradient is a per-sample sum over pixels: wave shuffle -> LDS -> one partial
// per workgroup -> fixed-order second stage (bitwise reproducible, no atomics).
#include "mal_common.h"
#include "mal_device.h"

namespace mal {

constexpr int kPxPerBlock = 1024;  // 256 threads x 4 pixels; blocks per sample <= tile count

struct WarpParams {
  const float* disp; const float* K; const float* invK;
  const float* T[2]; const float* src[2];
  const float* g_warped[2]; const float* g_grid[2]; const float* g_depth;
  float* depth_out; float* grid_out[2]; float* warped_out[2];
  float* g_disp; float* block_gP;
  int B, H, W, F; float min_disp, range, eps; int convention; int bps;  // blocks per sample
  int packed;  // src[f] are texel copies made by mal_pack_texels (MAL_WARP_TEXELS): one gather per tap instead of three
};

__device__ inline void load_cam(const WarpParams& p, int b, float (*sP)[12], float* sik) {
  const int tid = threadIdx.x;
  if (tid < 12 * p.F) {
    int f = tid / 12, e = tid % 12, i = e >> 2, j = e & 3;
    const float* K = p.K + b * 16;
    const float* T = p.T[f] + b * 16;
    float acc = K[i * 4 + 0] * T[0 * 4 + j];
    acc = fma_(K[i * 4 + 1], T[1 * 4 + j], acc);
    acc = fma_(K[i * 4 + 2], T[2 * 4 + j], acc);
    acc = fma_(K[i * 4 + 3], T[3 * 4 + j], acc);
    sP[f][e] = acc;
  } else if (tid >= 32 && tid < 41) {
    int e = tid - 32;
    sik[e] = p.invK[b * 16 + (e / 3) * 4 + (e % 3)];
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void warp_fwd_kernel(WarpParams p) {
  __shared__ float sP[2][12];
  __shared__ float sik[9];
  const int b = blockIdx.x / p.bps, chunk = blockIdx.x % p.bps;
  load_cam(p, b, sP, sik);
  const int HW = p.H * p.W;
  for (int k = 0; k < 4; ++k) {
    const int pix = chunk * kPxPerBlock + k * 256 + threadIdx.x;
    if (pix >= HW) break;
    const int gy = pix / p.W, gx = pix - gy * p.W;
    const size_t gi = (size_t)b * HW + pix;
    const float depth = depth_of(p.disp[gi], p.min_disp, p.range);
    if (p.depth_out) p.depth_out[gi] = depth;
    float ray[3], X[3];
    ray_of(sik, (float)gx, (float)gy, ray);
    X[0] = depth * ray[0]; X[1] = depth * ray[1]; X[2] = depth * ray[2];
    for (int f = 0; f < p.F; ++f) {
      Sample s = project_pixel(sP[f], X, p.eps, p.W, p.H, p.convention);
      if (p.grid_out[f]) { p.grid_out[f][gi * 2] = s.gx; p.grid_out[f][gi * 2 + 1] = s.gy; }
      if (p.warped_out[f]) {
        Taps t = make_taps(s.ix, s.iy, p.W, p.H);
        float* ob = p.warped_out[f] + (size_t)b * 3 * HW;
        float ta[3], tb[3], tc[3], td[3];
        load_taps(p.src[f], p.packed, b, HW, t, ta, tb, tc, td);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) ob[ch * HW + pix] = blend(t, ta[ch], tb[ch], tc[ch], td[ch]);
      }
    }
  }
}

// d(normalised grid)/d(u): Project3D's own scaling (layers.py:192-194 / dualrefine 224-225)
MAL_DEV float grid_scale(int size, int convention) {
  return convention == 0 ? 2.0f / (float)(size - 1) : 2.0f / (float)size;
}

template <bool POSE>
__global__ __launch_bounds__(256) void warp_bwd_kernel(WarpParams p) {
  __shared__ float sP[2][12];
  __shared__ float sik[9];
  __shared__ float s_redP[4][24];
  const int b = blockIdx.x / p.bps, chunk = blockIdx.x % p.bps;
  load_cam(p, b, sP, sik);
  const int HW = p.H * p.W, tid = threadIdx.x;
  float gP[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) gP[i] = 0.f;
  const float sgx = grid_scale(p.W, p.convention), sgy = grid_scale(p.H, p.convention);
  for (int k = 0; k < 4; ++k) {
    const int pix = chunk * kPxPerBlock + k * 256 + tid;
    if (pix >= HW) break;
    const int gy = pix / p.W, gx = pix - gy * p.W;
    const size_t gi = (size_t)b * HW + pix;
    const float depth = depth_of(p.disp[gi], p.min_disp, p.range);
    float ray[3], X[4];
    ray_of(sik, (float)gx, (float)gy, ray);
    X[0] = depth * ray[0]; X[1] = depth * ray[1]; X[2] = depth * ray[2]; X[3] = 1.0f;
    float gX[3] = {0.f, 0.f, 0.f};
    for (int f = 0; f < p.F; ++f) {
      Sample s = project_pixel(sP[f], X, p.eps, p.W, p.H, p.convention);
      float gu = 0.f, gv = 0.f;
      if (p.g_warped[f]) {
        Taps t = make_taps(s.ix, s.iy, p.W, p.H);
        const float* gb = p.g_warped[f] + (size_t)b * 3 * HW;
        float gix = 0.f, giy = 0.f;
        float ta[3], tb[3], tc[3], td[3];
        load_taps(p.src[f], p.packed, b, HW, t, ta, tb, tc, td);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          float dx, dy;
          blend_grad(t, ta[ch], tb[ch], tc[ch], td[ch], &dx, &dy);
          const float g = gb[ch * HW + pix];
          gix = fma_(g, dx, gix);
          giy = fma_(g, dy, giy);
        }
        gu = gix * s.mx;
        gv = giy * s.my;
      }
      if (p.g_grid[f]) {  // gradient arriving on the ("sample", f, s) output itself
        gu = fma_(p.g_grid[f][gi * 2], sgx, gu);
        gv = fma_(p.g_grid[f][gi * 2 + 1], sgy, gv);
      }
      float a[3];
      a[0] = gu * s.rz;
      a[1] = gv * s.rz;
      a[2] = -(gu * s.u + gv * s.v) * s.rz;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) gX[j] = fma_(a[i], sP[f][i * 4 + j], gX[j]);
        if (POSE)
#pragma unroll
          for (int j = 0; j < 4; ++j) gP[f * 12 + i * 4 + j] = fma_(a[i], X[j], gP[f * 12 + i * 4 + j]);
      }
    }
    float gdepth = gX[0] * ray[0] + gX[1] * ray[1] + gX[2] * ray[2];
    if (p.g_depth) gdepth += p.g_depth[gi];
    p.g_disp[gi] = gdepth * (-(depth * depth) * p.range);
  }
  if (POSE) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      float v = wave_sum(gP[i]);
      if (lane == 0) s_redP[wv][i] = v;
    }
    __syncthreads();
    if (tid < 24)
      p.block_gP[(size_t)blockIdx.x * 24 + tid] = (s_redP[0][tid] + s_redP[1][tid]) + (s_redP[2][tid] + s_redP[3][tid]);
  }
}

// g_T[f][b] = K_b^T [sum_blocks gP_fb ; 0]
__global__ __launch_bounds__(256) void pose_grad_finalize_kernel(const float* block_gP, const float* K, int bps, int F,
                                                                 float* gT0, float* gT1) {
  __shared__ double s_gP[24];
  __shared__ double s_part[240];
  const int b = blockIdx.x, tid = threadIdx.x;
  // 24 sums of bps partials: 10 threads per value (independent loads issued together), then a 10-term sum, fixed order
  if (tid < 240) {
    const int v = tid % 24, sub = tid / 24;
    double acc = 0.0;
#pragma unroll 4
    for (int t = sub; t < bps; t += 10) acc += (double)block_gP[((size_t)b * bps + t) * 24 + v];
    s_part[tid] = acc;
  }
  __syncthreads();
  if (tid < 24) {
    double acc = 0.0;
    for (int k = 0; k < 10; ++k) acc += s_part[k * 24 + tid];
    s_gP[tid] = acc;
  }
  __syncthreads();
  if (tid < 16 * F) {
    int f = tid >> 4, e = tid & 15, k = e >> 2, j = e & 3;
    const float* Kb = K + b * 16;
    double acc = 0.0;
    for (int i = 0; i < 3; ++i) acc += (double)Kb[i * 4 + k] * s_gP[f * 12 + i * 4 + j];
    float* out = f ? gT1 : gT0;
    if (out) out[b * 16 + e] = (float)acc;
  }
}

// ---------------------------------------------------------------- granular elementwise ops
__global__ void disp_to_depth_kernel(const float* disp, size_t n, float min_disp, float range, float* scaled,
                                     float* depth) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s;
    float d = depth_of(disp[i], min_disp, range, &s);
    if (scaled) scaled[i] = s;
    if (depth) depth[i] = d;
  }
}

__global__ void disp_to_depth_bwd_kernel(const float* disp, const float* g_scaled, const float* g_depth, size_t n,
                                         float min_disp, float range, float* g_disp) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float d = depth_of(disp[i], min_disp, range);
    float g = 0.f;
    if (g_scaled) g += g_scaled[i] * range;
    if (g_depth) g += g_depth[i] * (-(d * d) * range);
    g_disp[i] = g;
  }
}

__global__ void backproject_kernel(const float* depth, const float* invK, int B, int H, int W, float* points) {
  const int HW = H * W;
  const size_t n = (size_t)B * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW), pix = (int)(i - (size_t)b * HW);
    const int gy = pix / W, gx = pix - gy * W;
    const float* ik4 = invK + b * 16;
    float ik[9] = {ik4[0], ik4[1], ik4[2], ik4[4], ik4[5], ik4[6], ik4[8], ik4[9], ik4[10]};
    float ray[3];
    ray_of(ik, (float)gx, (float)gy, ray);
    const float d = depth[i];
    float* o = points + (size_t)b * 4 * HW + pix;
    o[0] = d * ray[0]; o[HW] = d * ray[1]; o[2 * HW] = d * ray[2]; o[3 * HW] = 1.0f;
  }
}

__global__ void backproject_bwd_kernel(const float* g_points, const float* invK, int B, int H, int W, float* g_depth) {
  const int HW = H * W;
  const size_t n = (size_t)B * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW), pix = (int)(i - (size_t)b * HW);
    const int gy = pix / W, gx = pix - gy * W;
    const float* ik4 = invK + b * 16;
    float ik[9] = {ik4[0], ik4[1], ik4[2], ik4[4], ik4[5], ik4[6], ik4[8], ik4[9], ik4[10]};
    float ray[3];
    ray_of(ik, (float)gx, (float)gy, ray);
    const float* g = g_points + (size_t)b * 4 * HW + pix;
    g_depth[i] = g[0] * ray[0] + g[HW] * ray[1] + g[2 * HW] * ray[2];
  }
}

struct ProjParams {
  const float* points; const float* K; const float* T; const float* g_grid; const float* g_z;
  float* grid; float* z_out; float* g_points; float* block_gP;
  int B, H, W; float eps; int convention; int bps;
};

template <bool BWD>
__global__ __launch_bounds__(256) void project_kernel(ProjParams p) {
  __shared__ float sP[12];
  __shared__ float s_redP[4][12];
  const int b = blockIdx.x / p.bps, chunk = blockIdx.x % p.bps, tid = threadIdx.x;
  if (tid < 12) {
    int i = tid >> 2, j = tid & 3;
    const float* K = p.K + b * 16;
    const float* T = p.T + b * 16;
    float acc = K[i * 4 + 0] * T[0 * 4 + j];
    acc = fma_(K[i * 4 + 1], T[1 * 4 + j], acc);
    acc = fma_(K[i * 4 + 2], T[2 * 4 + j], acc);
    acc = fma_(K[i * 4 + 3], T[3 * 4 + j], acc);
    sP[tid] = acc;
  }
  __syncthreads();
  const int HW = p.H * p.W;
  float gP[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) gP[i] = 0.f;
  const float sgx = grid_scale(p.W, p.convention), sgy = grid_scale(p.H, p.convention);
  for (int k = 0; k < 4; ++k) {
    const int pix = chunk * kPxPerBlock + k * 256 + tid;
    if (pix >= HW) break;
    const size_t gi = (size_t)b * HW + pix;
    const float* pt = p.points + (size_t)b * 4 * HW + pix;
    float X[4] = {pt[0], pt[HW], pt[2 * HW], pt[3 * HW]};
    float c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float acc = sP[4 * i] * X[0];
      acc = fma_(sP[4 * i + 1], X[1], acc);
      acc = fma_(sP[4 * i + 2], X[2], acc);
      c[i] = fma_(sP[4 * i + 3], X[3], acc);
    }
    const float zp = c[2] + p.eps;
    const float u = div_safe_(c[0], zp), v = div_safe_(c[1], zp);
    if (!BWD) {
      float gx, gy;
      if (p.convention == 0) {
        gx = (div_(u, (float)(p.W - 1)) - 0.5f) * 2.0f;
        gy = (div_(v, (float)(p.H - 1)) - 0.5f) * 2.0f;
      } else {
        gx = div_(2.0f * (u + 0.5f), (float)p.W) - 1.0f;
        gy = div_(2.0f * (v + 0.5f), (float)p.H) - 1.0f;
      }
      p.grid[gi * 2] = gx;
      p.grid[gi * 2 + 1] = gy;
      if (p.z_out) p.z_out[gi] = c[2];
    } else {
      const float rz = __builtin_amdgcn_rcpf(zp);
      const float gu = p.g_grid[gi * 2] * sgx, gv = p.g_grid[gi * 2 + 1] * sgy;
      float a[3];
      a[0] = gu * rz;
      a[1] = gv * rz;
      a[2] = -(gu * u + gv * v) * rz;
      if (p.g_z) a[2] += p.g_z[gi];
      float gX[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          gX[j] = fma_(a[i], sP[i * 4 + j], gX[j]);
          gP[i * 4 + j] = fma_(a[i], X[j], gP[i * 4 + j]);
        }
      if (p.g_points) {
        float* o = p.g_points + (size_t)b * 4 * HW + pix;
        o[0] = gX[0]; o[HW] = gX[1]; o[2 * HW] = gX[2]; o[3 * HW] = gX[3];
      }
    }
  }
  if (BWD) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      float v = wave_sum(gP[i]);
      if (lane == 0) s_redP[wv][i] = v;
    }
    __syncthreads();
    if (tid < 12) {
      p.block_gP[(size_t)blockIdx.x * 24 + tid] = (s_redP[0][tid] + s_redP[1][tid]) + (s_redP[2][tid] + s_redP[3][tid]);
      p.block_gP[(size_t)blockIdx.x * 24 + 12 + tid] = 0.f;
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void grid_sample_kernel(const float* src, const float* grid, const float* g_out,
                                                          int B, int C, int H, int W, int Ho, int Wo, int ac,
                                                          float* out, float* g_grid) {
  const int HWo = Ho * Wo, HW = H * W;
  const size_t n = (size_t)B * HWo;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HWo), pix = (int)(i - (size_t)b * HWo);
    float mx, my;
    const float ix = unnormalize_clip(grid[i * 2], W, ac, &mx);
    const float iy = unnormalize_clip(grid[i * 2 + 1], H, ac, &my);
    Taps t = make_taps(ix, iy, W, H);
    const float* sb = src + (size_t)b * C * HW;
    if (!BWD) {
      float* ob = out + (size_t)b * C * HWo;
      for (int ch = 0; ch < C; ++ch) {
        const float* pl = sb + (size_t)ch * HW;
        ob[(size_t)ch * HWo + pix] = blend(t, pl[t.o00], pl[t.o01], pl[t.o10], pl[t.o11]);
      }
    } else {
      const float* gb = g_out + (size_t)b * C * HWo;
      float gix = 0.f, giy = 0.f;
      for (int ch = 0; ch < C; ++ch) {
        const float* pl = sb + (size_t)ch * HW;
        float dx, dy;
        blend_grad(t, pl[t.o00], pl[t.o01], pl[t.o10], pl[t.o11], &dx, &dy);
        const float g = gb[(size_t)ch * HWo + pix];
        gix = fma_(g, dx, gix);
        giy = fma_(g, dy, giy);
      }
      g_grid[i * 2] = gix * mx;
      g_grid[i * 2 + 1] = giy * my;
    }
  }
}

__global__ void matching_mask_kernel(const float* lowest_cost, const float* mono_depth, const float* cmask, size_t n,
                                     float* out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float mono = mono_depth[i];
    const float matching = div_safe_(1.0f, lowest_cost[i]);
    const bool ok = (div_safe_(matching - mono, mono) < 1.0f) && (div_safe_(mono - matching, matching) < 1.0f);
    const float m = ok ? 1.0f : 0.0f;
    out[i] = cmask ? cmask[i] * m : m;
  }
}

// (B,3,H,W) -> (B,H,W,kTexel): one texel (r, g, b) per pixel
__global__ void pack_texels_kernel(const float* src, int B, int HW, float* dst) {
  const size_t n = (size_t)B * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, pix = i - b * HW;
    const float* s = src + b * 3 * HW + pix;
    st_texel(dst, i, make_texel(s[0], s[HW], s[2 * (size_t)HW]));
  }
}

inline int ew_grid(size_t n) {
  size_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

inline void fill_warp_params(WarpParams& p, const float* disp, const float* K, const float* inv_K, const float* const* T,
                             const float* const* src, int B, int H, int W, int F, float min_depth, float max_depth,
                             float eps, int convention) {
  p.disp = disp; p.K = K; p.invK = inv_K;
  for (int f = 0; f < 2; ++f) {
    p.T[f] = f < F ? T[f] : nullptr;
    p.src[f] = (f < F && src) ? src[f] : nullptr;
    p.g_warped[f] = p.g_grid[f] = nullptr;
    p.grid_out[f] = p.warped_out[f] = nullptr;
  }
  p.g_depth = nullptr; p.depth_out = nullptr; p.g_disp = nullptr; p.block_gP = nullptr;
  p.B = B; p.H = H; p.W = W; p.F = F;
  p.min_disp = (float)(1.0 / (double)max_depth);
  p.range = (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth);
  p.eps = eps; p.convention = convention;
  p.bps = (H * W + kPxPerBlock - 1) / kPxPerBlock;
}

}  // namespace mal

using namespace mal;

extern "C" int mal_warp_fwd(const float* disp, const float* K, const float* inv_K, const float* const* T,
                            const float* const* src, int B, int H, int W, int F, float min_depth, float max_depth,
                            float eps, int convention, float* depth_out, float* const* grid_out,
                            float* const* warped_out, void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (F < 1 || F > MAL_MAX_FRAMES || !disp || !K || !inv_K || !T) return MAL_EINVAL;
  const int texels = (convention & MAL_WARP_TEXELS) ? 1 : 0;
  convention &= ~MAL_WARP_TEXELS;
  if (convention != 0 && convention != 1) return MAL_EINVAL;
  WarpParams p;
  fill_warp_params(p, disp, K, inv_K, T, src, B, H, W, F, min_depth, max_depth, eps, convention);
  p.packed = texels;
  p.depth_out = depth_out;
  for (int f = 0; f < F; ++f) {
    if (!T[f]) return MAL_EINVAL;
    p.grid_out[f] = grid_out ? grid_out[f] : nullptr;
    p.warped_out[f] = warped_out ? warped_out[f] : nullptr;
    if (p.warped_out[f] && (!src || !src[f])) return MAL_EINVAL;
  }
  hipLaunchKernelGGL(warp_fwd_kernel, dim3(p.bps * B), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

extern "C" int mal_warp_bwd(const float* disp, const float* K, const float* inv_K, const float* const* T,
                            const float* const* src, const float* const* g_warped, const float* const* g_grid,
                            const float* g_depth, int B, int H, int W, int F, float min_depth, float max_depth,
                            float eps, int convention, float* g_disp, float* const* g_T, void* ws, size_t ws_bytes,
                            void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (F < 1 || F > MAL_MAX_FRAMES || !disp || !K || !inv_K || !T || !g_disp || !ws) return MAL_EINVAL;
  const int texels = (convention & MAL_WARP_TEXELS) ? 1 : 0;
  convention &= ~MAL_WARP_TEXELS;
  if (convention != 0 && convention != 1) return MAL_EINVAL;
  Workspace w = carve(ws, B, H, W);
  if (ws_bytes < w.bytes) return MAL_EWORKSPACE;
  WarpParams p;
  fill_warp_params(p, disp, K, inv_K, T, src, B, H, W, F, min_depth, max_depth, eps, convention);
  p.packed = texels;
  for (int f = 0; f < F; ++f) {
    if (!T[f]) return MAL_EINVAL;
    p.g_warped[f] = g_warped ? g_warped[f] : nullptr;
    p.g_grid[f] = g_grid ? g_grid[f] : nullptr;
    if (p.g_warped[f] && (!src || !src[f])) return MAL_EINVAL;
  }
  p.g_depth = g_depth; p.g_disp = g_disp; p.block_gP = w.block_gP;
  const bool pose = g_T && (g_T[0] || (F > 1 && g_T[1]));
  hipStream_t st = (hipStream_t)stream;
  if (pose) hipLaunchKernelGGL(warp_bwd_kernel<true>, dim3(p.bps * B), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(warp_bwd_kernel<false>, dim3(p.bps * B), dim3(256), 0, st, p);
  rc = launch_status();
  if (rc || !pose) return rc;
  hipLaunchKernelGGL(pose_grad_finalize_kernel, dim3(B), dim3(256), 0, st, w.block_gP, K, p.bps, F, g_T[0],
                     F > 1 ? g_T[1] : nullptr);
  return launch_status();
}

extern "C" int mal_disp_to_depth(const float* disp, size_t n, float min_depth, float max_depth, float* scaled_out,
                                 float* depth_out, void* stream) {
  if (!disp || n == 0) return MAL_EINVAL;
  hipLaunchKernelGGL(disp_to_depth_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, disp, n,
                     (float)(1.0 / (double)max_depth), (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth),
                     scaled_out, depth_out);
  return launch_status();
}

extern "C" int mal_disp_to_depth_bwd(const float* disp, const float* g_scaled, const float* g_depth, size_t n,
                                     float min_depth, float max_depth, float* g_disp, void* stream) {
  if (!disp || !g_disp || n == 0) return MAL_EINVAL;
  hipLaunchKernelGGL(disp_to_depth_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, disp, g_scaled,
                     g_depth, n, (float)(1.0 / (double)max_depth),
                     (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth), g_disp);
  return launch_status();
}

extern "C" int mal_backproject(const float* depth, const float* inv_K, int B, int H, int W, float* points,
                               void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (!depth || !inv_K || !points) return MAL_EINVAL;
  hipLaunchKernelGGL(backproject_kernel, dim3(ew_grid((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, depth,
                     inv_K, B, H, W, points);
  return launch_status();
}

extern "C" int mal_backproject_bwd(const float* g_points, const float* inv_K, int B, int H, int W, float* g_depth,
                                   void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (!g_points || !inv_K || !g_depth) return MAL_EINVAL;
  hipLaunchKernelGGL(backproject_bwd_kernel, dim3(ew_grid((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream,
                     g_points, inv_K, B, H, W, g_depth);
  return launch_status();
}

extern "C" int mal_project3d(const float* points, const float* K, const float* T, int B, int H, int W, float eps,
                             int convention, float* grid, float* z_out, void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (!points || !K || !T || !grid || (convention != 0 && convention != 1)) return MAL_EINVAL;
  ProjParams p = {points, K, T, nullptr, nullptr, grid, z_out, nullptr, nullptr, B, H, W, eps, convention,
                  (H * W + kPxPerBlock - 1) / kPxPerBlock};
  hipLaunchKernelGGL(project_kernel<false>, dim3(p.bps * B), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

extern "C" int mal_project3d_bwd(const float* points, const float* K, const float* T, const float* g_grid,
                                 const float* g_z, int B, int H, int W, float eps, int convention, float* g_points,
                                 float* g_T, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (!points || !K || !T || !g_grid || !ws || (convention != 0 && convention != 1)) return MAL_EINVAL;
  Workspace w = carve(ws, B, H, W);
  if (ws_bytes < w.bytes) return MAL_EWORKSPACE;
  ProjParams p = {points, K, T, g_grid, g_z, nullptr, nullptr, g_points, w.block_gP, B, H, W, eps, convention,
                  (H * W + kPxPerBlock - 1) / kPxPerBlock};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(project_kernel<true>, dim3(p.bps * B), dim3(256), 0, st, p);
  rc = launch_status();
  if (rc || !g_T) return rc;
  hipLaunchKernelGGL(pose_grad_finalize_kernel, dim3(B), dim3(256), 0, st, w.block_gP, K, p.bps, 1, g_T, nullptr);
  return launch_status();
}

extern "C" int mal_grid_sample(const float* src, const float* grid, int B, int C, int H, int W, int Ho, int Wo,
                               int align_corners, float* out, void* stream) {
  if (!src || !grid || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return MAL_EINVAL;
  hipLaunchKernelGGL(grid_sample_kernel<false>, dim3(ew_grid((size_t)B * Ho * Wo)), dim3(256), 0, (hipStream_t)stream,
                     src, grid, nullptr, B, C, H, W, Ho, Wo, align_corners, out, nullptr);
  return launch_status();
}

extern "C" int mal_grid_sample_bwd(const float* src, const float* grid, const float* g_out, int B, int C, int H, int W,
                                   int Ho, int Wo, int align_corners, float* g_grid, void* stream) {
  if (!src || !grid || !g_out || !g_grid || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0)
    return MAL_EINVAL;
  hipLaunchKernelGGL(grid_sample_kernel<true>, dim3(ew_grid((size_t)B * Ho * Wo)), dim3(256), 0, (hipStream_t)stream,
                     src, grid, g_out, B, C, H, W, Ho, Wo, align_corners, nullptr, g_grid);
  return launch_status();
}

extern "C" int mal_texel_floats(void) { return kTexel; }

extern "C" int mal_pack_texels(const float* src, int B, int H, int W, float* dst, void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (!src || !dst) return MAL_EINVAL;
  hipLaunchKernelGGL(pack_texels_kernel, dim3(ew_grid((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, src, B,
                     H * W, dst);
  return launch_status();
}

extern "C" int mal_matching_mask(const float* lowest_cost, const float* mono_depth, const float* consistency_mask,
                                 size_t n, float* out, void* stream) {
  if (!lowest_cost || !mono_depth || !out || n == 0) return MAL_EINVAL;
  hipLaunchKernelGGL(matching_mask_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, lowest_cost,
                     mono_depth, consistency_mask, n, out);
  return launch_status();
}

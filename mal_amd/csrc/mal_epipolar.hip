// N4, forward: DualRefine's epipolar correlation lookup as DEQDepthPose's fixed-point function
// calls it every iteration (dualrefine/networks/depth_pose.py:433-435):
//
//   epi_coords_kernel   Reprojections.depth2epipolarcoords   dualrefine/networks/utils/utils.py:180-217
//                       (_iproj :132-153, _proj :155-168): per pixel, L levels x (2r+1) depth hypotheses around the
//                       current depth (spacing 2^level * softplus(delta) * depth / ratio / r), each back-projected,
//                       moved by the relative pose and projected -- the a2/a3 geometry of the hot path again.
//   epi_sample_kernel   CoordSampler.__call__                dualrefine/networks/corr.py:25-50: bilinear, zero-padded
//                       samples of the level's pooled feature map at those coordinates and the mean over each channel
//                       group of |fmap1 - sample|.  Upstream materialises (B,C,h,w,2r+1) per level through
//                       F.grid_sample and reduces it with three more launches; here lane = pixel, a wavefront owns 64
//                       pixels and kEpiG hypotheses, derives their taps once and walks the channel planes (the same
//                       scheme as the cost volume's match kernel, mal_costvol.hip).
#include "mal_common.h"
#include "mal_device.h"

namespace mal {

constexpr int kEpiMaxLevels = 4;
constexpr int kEpiG = 6;  // hypotheses per wavefront

struct EpiCoordParams {
  const float* depth; const float* poses; const float* K;
  int B, h, w, r, L; float dd, ratio;
  float* coords; float* max_dx; float* depths;
};

__global__ __launch_bounds__(256) void epi_coords_kernel(EpiCoordParams p) {
  const int hw = p.h * p.w, d1 = 2 * p.r + 1, D = p.L * d1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * hw) return;
  const int b = i / hw, pix = i - b * hw, y = pix / p.w, x = pix - y * p.w;
  const float* Kb = p.K + b * 16;
  const float* T = p.poses + b * 16;
  const float fx = Kb[0], fy = Kb[5], cx = Kb[2], cy = Kb[6];
  const float z0 = p.depth[i];
  const float X = div_((float)x - cx, fx), Y = div_((float)y - cy, fy);  // utils.py:147-148
  const float gap = div_(div_(p.dd * z0, p.ratio), (float)p.r);        // utils.py:196
  float mx = -INFINITY;
  for (int s = 0; s < D; ++s) {
    const int level = s / d1, j = s - level * d1;
    const float dxj = (float)(j - p.r);                    // torch.linspace(-r, r, 2r+1)
    const float dl = ((float)(1 << level) * 1.0f) * gap;   // (2**level) * ones * gap
    const float delta = dxj * dl;
    if (level == 0) mx = fmaxf(mx, delta);
    const float Z = z0 + delta;
    const float X0[4] = {Z * X, Z * Y, Z, 1.0f};
    float X1[3];
    for (int r_ = 0; r_ < 3; ++r_) {  // poses @ X0, rows 0..2
      float acc = T[r_ * 4] * X0[0];
      acc = fma_(T[r_ * 4 + 1], X0[1], acc);
      acc = fma_(T[r_ * 4 + 2], X0[2], acc);
      X1[r_] = fma_(T[r_ * 4 + 3], X0[3], acc);
    }
    const float inv = div_(1.0f, X1[2]);
    const float d = inv > 100.0f ? 100.0f : inv;           // clamp(max=100): NaN passes
    const float u = fx * (X1[0] * d) + cx, v = fy * (X1[1] * d) + cy;
    const size_t o = ((size_t)b * 2 * D + s) * hw + pix;
    p.coords[o] = u;
    p.coords[o + (size_t)D * hw] = v;
    p.depths[((size_t)b * D + s) * hw + pix] = Z;
  }
  p.max_dx[i] = mx;
}

struct EpiSampleParams {
  const float* fmap1; const float* f2[kEpiMaxLevels];
  const float* coords;
  int B, C, h, w, L, d1, heads;
  float* out;
};

__global__ __launch_bounds__(256) void epi_sample_kernel(EpiSampleParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hw = p.h * p.w, D = p.L * p.d1;
  // a wavefront's hypotheses belong to ONE level, so the level's map, its size and the channel plane are wave-uniform
  const int gpl = (p.d1 + kEpiG - 1) / kEpiG;  // groups per level
  const int level = blockIdx.y / gpl, j0 = (blockIdx.y - level * gpl) * kEpiG, b = blockIdx.z;
  const int pix0 = (blockIdx.x * 4 + wv) * 64;
  if (pix0 >= hw) return;
  const bool live = pix0 + lane < hw;
  const int pix = min(pix0 + lane, hw - 1);
  const int hl = p.h >> level, wl = p.w >> level, hwl = hl * wl;
  const float* f2l = p.f2[level];
  // taps of this lane's pixel for its hypotheses: clamped offsets, weights zeroed outside the level's map
  unsigned off[kEpiG][4];
  float wt[kEpiG][4];
#pragma unroll
  for (int g = 0; g < kEpiG; ++g) {
    const int s = level * p.d1 + min(j0 + g, p.d1 - 1);
    const size_t o = ((size_t)b * 2 * D + s) * hw + pix;
    const float u = p.coords[o], v = p.coords[o + (size_t)D * hw];
    // corr.py:38-39 then grid_sample's unnormalise for align_corners=False on the level's size
    const float gx = div_(2.0f * (u + 0.5f), (float)p.w) - 1.0f, gy = div_(2.0f * (v + 0.5f), (float)p.h) - 1.0f;
    const float ix = ((gx + 1.0f) * (float)wl - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)hl - 1.0f) / 2.0f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float xc = fminf(fmaxf(x0f, -2.0f), (float)wl + 1.0f), yc = fminf(fmaxf(y0f, -2.0f), (float)hl + 1.0f);
    const bool wild = !(x0f == xc && y0f == yc);  // far outside (or NaN): every tap is padding
    const int x0 = (int)xc, y0 = (int)yc, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = ix - x0f, ty = iy - y0f, ex = (x0f + 1.0f) - ix, ey = (y0f + 1.0f) - iy;
    const bool vx0 = x0 >= 0 && x0 < wl, vx1 = x1 >= 0 && x1 < wl, vy0 = y0 >= 0 && y0 < hl, vy1 = y1 >= 0 && y1 < hl;
    const int cx0 = min(max(x0, 0), wl - 1), cx1 = min(max(x1, 0), wl - 1), cy0 = min(max(y0, 0), hl - 1), cy1 = min(max(y1, 0), hl - 1);
    off[g][0] = (unsigned)(cy0 * wl + cx0) * 4u; off[g][1] = (unsigned)(cy0 * wl + cx1) * 4u;
    off[g][2] = (unsigned)(cy1 * wl + cx0) * 4u; off[g][3] = (unsigned)(cy1 * wl + cx1) * 4u;
    wt[g][0] = (!wild && vx0 && vy0) ? ex * ey : 0.f;
    wt[g][1] = (!wild && vx1 && vy0) ? tx * ey : 0.f;
    wt[g][2] = (!wild && vx0 && vy1) ? ex * ty : 0.f;
    wt[g][3] = (!wild && vx1 && vy1) ? tx * ty : 0.f;
  }
  const int cg = p.C / p.heads;  // channels per head
  const float inv_cg = 1.0f / (float)cg;
  for (int head = 0; head < p.heads; ++head) {
    float sum[kEpiG];
#pragma unroll
    for (int g = 0; g < kEpiG; ++g) sum[g] = 0.f;
#pragma unroll 4  // more planes' taps in flight (the channel count is a run-time value: no automatic unrolling)
    for (int c = head * cg; c < (head + 1) * cg; ++c) {
      const float f1 = p.fmap1[((size_t)b * p.C + c) * hw + pix];
      const char* pl = reinterpret_cast<const char*>(f2l + ((size_t)b * p.C + c) * hwl);
#pragma unroll
      for (int g = 0; g < kEpiG; ++g) {
        const float a = *reinterpret_cast<const float*>(pl + off[g][0]), bb = *reinterpret_cast<const float*>(pl + off[g][1]);
        const float cc = *reinterpret_cast<const float*>(pl + off[g][2]), d = *reinterpret_cast<const float*>(pl + off[g][3]);
        float o = a * wt[g][0];
        o = fma_(bb, wt[g][1], o);
        o = fma_(cc, wt[g][2], o);
        o = fma_(d, wt[g][3], o);
        sum[g] += fabsf(f1 - o);
      }
    }
#pragma unroll
    for (int g = 0; g < kEpiG; ++g) {
      const int j = j0 + g;
      if (live && j < p.d1)
        p.out[((size_t)b * D * p.heads + (size_t)level * p.heads * p.d1 + (size_t)head * p.d1 + j) * hw + pix] = sum[g] * inv_cg;
    }
  }
}

}  // namespace mal

using namespace mal;

extern "C" int mal_epipolar_coords(const float* depth, const float* poses, const float* K, int B, int h, int w, int r, int L,
                                   float softplus_delta, float ratio, float* coords, float* max_dx, float* depths,
                                   void* stream) {
  if (B <= 0 || h < 1 || w < 1 || r < 1 || L < 1 || L > kEpiMaxLevels) return MAL_ESHAPE;
  if ((double)B * 2 * L * (2 * r + 1) * h * w > 2.0e9 / 4) return MAL_ESHAPE;
  if (!depth || !poses || !K || !coords || !max_dx || !depths) return MAL_EINVAL;
  EpiCoordParams p = {depth, poses, K, B, h, w, r, L, softplus_delta, ratio, coords, max_dx, depths};
  hipLaunchKernelGGL(epi_coords_kernel, dim3((B * h * w + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

extern "C" int mal_coord_sample_l1(const float* fmap1, const float* const* f2_pyramid, const float* coords, int B, int C, int h,
                                   int w, int L, int d1, int heads, float* out, void* stream) {
  if (B <= 0 || C < 1 || h < 1 || w < 1 || L < 1 || L > kEpiMaxLevels || d1 < 1 || heads < 1 || C % heads) return MAL_ESHAPE;
  if ((h >> (L - 1)) < 1 || (w >> (L - 1)) < 1) return MAL_ESHAPE;
  if ((double)B * C * h * w > 2.0e9 / 4 || (double)B * 2 * L * d1 * h * w > 2.0e9 / 4) return MAL_ESHAPE;
  if (!fmap1 || !f2_pyramid || !coords || !out) return MAL_EINVAL;
  EpiSampleParams p = {};
  p.fmap1 = fmap1; p.coords = coords; p.B = B; p.C = C; p.h = h; p.w = w; p.L = L; p.d1 = d1; p.heads = heads; p.out = out;
  for (int l = 0; l < L; ++l) {
    if (!f2_pyramid[l]) return MAL_EINVAL;
    p.f2[l] = f2_pyramid[l];
  }
  hipLaunchKernelGGL(epi_sample_kernel, dim3((h * w + 255) / 256, L * ((d1 + kEpiG - 1) / kEpiG), B), dim3(256), 0,
                     (hipStream_t)stream, p);
  return launch_status();
}

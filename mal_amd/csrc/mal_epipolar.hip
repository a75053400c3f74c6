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
#ifndef MAL_EPI_G
#define MAL_EPI_G 3  // 2-3: 0.70 ms, 4: 0.77, 6: 0.73 at B=8, 128 channels, 48x160, 51 hypotheses
#endif
constexpr int kEpiG = MAL_EPI_G;  // hypotheses per wavefront
opt_t g_epi_bwd_planes{2};  // 2: two channel planes per workgroup where they fit (round 4); 1: one (round 3); 0: global atomics
opt_t g_epi_probe{0};  // option "epi_probe": timing experiments of the plane kernel (wrong results), see EpiSampleBwdParams::probe         // option "epi_bwd_planes": 0 = the global-atomic scatter everywhere (A/B)

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

// _iproj / pose / _proj for GIVEN depth hypotheses (Reprojections.depthbins2coords, utils.py:251-253): one thread per
// (sample, hypothesis, pixel)
__global__ __launch_bounds__(256) void epi_coords_of_depths_kernel(const float* depths, const float* poses, const float* K,
                                                                   int B, int D, int h, int w, float* coords) {
  const int hw = h * w;
  const size_t n = (size_t)B * D * hw;
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i >= n) return;
  const int pix = (int)(i % hw), s = (int)((i / hw) % D), b = (int)(i / ((size_t)hw * D));
  const int y = pix / w, x = pix - y * w;
  const float* Kb = K + b * 16;
  const float* T = poses + b * 16;
  const float fx = Kb[0], fy = Kb[5], cx = Kb[2], cy = Kb[6];
  const float Z = depths[i];
  const float X = div_((float)x - cx, fx), Y = div_((float)y - cy, fy);
  const float X0[4] = {Z * X, Z * Y, Z, 1.0f};
  float X1[3];
  for (int r_ = 0; r_ < 3; ++r_) {
    float acc = T[r_ * 4] * X0[0];
    acc = fma_(T[r_ * 4 + 1], X0[1], acc);
    acc = fma_(T[r_ * 4 + 2], X0[2], acc);
    X1[r_] = fma_(T[r_ * 4 + 3], X0[3], acc);
  }
  const float inv = div_(1.0f, X1[2]);
  const float d = inv > 100.0f ? 100.0f : inv;
  const size_t o = ((size_t)b * 2 * D + s) * hw + pix;
  coords[o] = fx * (X1[0] * d) + cx;
  coords[o + (size_t)D * hw] = fy * (X1[1] * d) + cy;
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

// ---------------------------------------------------------------- VJPs of the lookup
// depth2epipolarcoords backward: cotangents of (coords, max_dx, depths) -> d/d depth (per pixel), d/d poses[:3] and
// d/d softplus(delta) (per-workgroup partials, fixed-order second stage; no atomics).  One thread per pixel re-derives
// the forward chain of each hypothesis:  Z = z0 + dx_j 2^l gap,  gap = dd z0 / ratio / r;  X1 = T[:3] (ZX, ZY, Z, 1);
// d = min(1/X1z, 100);  u = fx X1x d + cx,  v = fy X1y d + cy.
struct EpiCoordBwdParams {
  const float* depth; const float* poses; const float* K;
  const float* g_coords; const float* g_max_dx; const float* g_depths;  // the last two nullable
  int B, h, w, r, L, nblk; float dd, ratio;
  float* g_depth; double* partial;  // [B][nblk][13]: d/dT rows 0..2 (12), d/d dd
};

__global__ __launch_bounds__(256) void epi_coords_bwd_kernel(EpiCoordBwdParams p) {
  __shared__ double s_red[13][4];
  const int hw = p.h * p.w, d1 = 2 * p.r + 1, D = p.L * d1;
  const int b = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x;
  const bool live = pix < hw;
  float acc[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) acc[k] = 0.f;
  if (live) {
    const int y = pix / p.w, x = pix - y * p.w, i = b * hw + pix;
    const float* Kb = p.K + b * 16;
    const float* T = p.poses + b * 16;
    const float fx = Kb[0], fy = Kb[5], cx = Kb[2], cy = Kb[6];
    const float z0 = p.depth[i];
    const float X = div_((float)x - cx, fx), Y = div_((float)y - cy, fy);
    const float gap = div_(div_(p.dd * z0, p.ratio), (float)p.r);
    const float dgap_dz0 = div_(div_(p.dd, p.ratio), (float)p.r), dgap_ddd = div_(div_(z0, p.ratio), (float)p.r);
    float g_z0 = 0.f;
    for (int s = 0; s < D; ++s) {
      const int level = s / d1, j = s - level * d1;
      const float cj = (float)(j - p.r) * (float)(1 << level);  // d Z / d gap
      const float Z = z0 + cj * gap;
      const float X0[4] = {Z * X, Z * Y, Z, 1.0f};
      float X1[3];
      for (int r_ = 0; r_ < 3; ++r_) {
        float a = T[r_ * 4] * X0[0];
        a = fma_(T[r_ * 4 + 1], X0[1], a);
        a = fma_(T[r_ * 4 + 2], X0[2], a);
        X1[r_] = fma_(T[r_ * 4 + 3], X0[3], a);
      }
      const float inv = div_(1.0f, X1[2]);
      const bool clamped = inv > 100.0f;
      const float d = clamped ? 100.0f : inv;
      const size_t o = ((size_t)b * 2 * D + s) * hw + pix;
      const float gu = p.g_coords[o], gv = p.g_coords[o + (size_t)D * hw];
      float gX1[3];
      gX1[0] = gu * fx * d;
      gX1[1] = gv * fy * d;
      const float g_d = gu * fx * X1[0] + gv * fy * X1[1];
      gX1[2] = clamped ? 0.f : -g_d * inv * inv;  // torch.clamp(max=100) passes the gradient where 1/Z <= 100
      float gZ = p.g_depths ? p.g_depths[((size_t)b * D + s) * hw + pix] : 0.f;
#pragma unroll
      for (int r_ = 0; r_ < 3; ++r_) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r_ * 4 + c] += gX1[r_] * X0[c];
        gZ += gX1[r_] * (T[r_ * 4] * X + T[r_ * 4 + 1] * Y + T[r_ * 4 + 2]);
      }
      g_z0 += gZ * (1.0f + cj * dgap_dz0);
      acc[12] += gZ * cj * dgap_ddd;
    }
    if (p.g_max_dx) {  // max over the level-0 offsets dx_j * gap: at j = 2r for gap >= 0, at j = 0 otherwise
      const float gm = p.g_max_dx[i], cm = gap >= 0.f ? (float)p.r : -(float)p.r;
      g_z0 += gm * cm * dgap_dz0;
      acc[12] += gm * cm * dgap_ddd;
    }
    p.g_depth[i] = g_z0;
  }
  // 13 sums over the workgroup: wave shuffle, then the four waves in order
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    const double v = wave_sum_d((double)acc[k]);
    if (lane == 0) s_red[k][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < 13)
    p.partial[((size_t)b * p.nblk + blockIdx.x) * 13 + threadIdx.x] =
        ((s_red[threadIdx.x][0] + s_red[threadIdx.x][1]) + s_red[threadIdx.x][2]) + s_red[threadIdx.x][3];
}

// second stage: g_poses (B,16) rows 0..2 (row 3 receives nothing), g_dd (1) -- sums over workgroups / samples in order
__global__ __launch_bounds__(64) void epi_coords_bwd_finish_kernel(const double* partial, int B, int nblk, float* g_poses,
                                                                   float* g_dd) {
  const int t = threadIdx.x;
  if (t < 12)
    for (int b = 0; b < B; ++b) {
      double a = 0.0;
      for (int k = 0; k < nblk; ++k) a += partial[((size_t)b * nblk + k) * 13 + t];
      g_poses[b * 16 + t] = (float)a;
    }
  if (t >= 12 && t < 16)
    for (int b = 0; b < B; ++b) g_poses[b * 16 + t] = 0.f;
  if (t == 16) {
    double a = 0.0;
    for (int b = 0; b < B; ++b)
      for (int k = 0; k < nblk; ++k) a += partial[((size_t)b * nblk + k) * 13 + 12];
    *g_dd = (float)a;
  }
}

// CoordSampler.__call__ backward: cotangent of out (B, L*heads*d1, h, w) -> d/d fmap1, d/d each pyramid level of
// fmap2 (scatter: float atomics, as ATen's grid_sampler backward does -- the one place of this library whose summation
// order is not fixed), d/d coords.  Same decomposition as the forward: lane = pixel, a wavefront owns kEpiG hypotheses of
// one level and walks the channel planes re-deriving the samples.
struct EpiSampleBwdParams {
  const float* fmap1; const float* f2[kEpiMaxLevels];
  const float* coords; const float* g_out;
  int B, C, h, w, L, d1, heads;
  float* g_fmap1; float* g_f2[kEpiMaxLevels]; float* g_coords;  // g_fmap1 / g_f2 zero-initialised by the caller; each nullable
  // experiments only (mal_set_option("epi_probe"), scripts/epi_bwd_probe.py; results are WRONG with any bit set): what the
  // plane kernel's time is made of -- bit 0: taps without divisions / validity tests, bit 1: no LDS adds, bit 2: no gathers
  int probe;
};

__global__ __launch_bounds__(256) void epi_sample_bwd_kernel(EpiSampleBwdParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hw = p.h * p.w, D = p.L * p.d1;
  const int gpl = (p.d1 + kEpiG - 1) / kEpiG;
  const int level = blockIdx.y / gpl, j0 = (blockIdx.y - level * gpl) * kEpiG, b = blockIdx.z;
  const int pix0 = (blockIdx.x * 4 + wv) * 64;
  if (pix0 >= hw) return;
  const bool live = pix0 + lane < hw;
  const int pix = min(pix0 + lane, hw - 1);
  const int hl = p.h >> level, wl = p.w >> level, hwl = hl * wl;
  const float* f2l = p.f2[level];
  float* g_f2l = p.g_f2[level];
  unsigned off[kEpiG][4];
  float wt[kEpiG][4], dwx[kEpiG][4], dwy[kEpiG][4];  // tap weights and their derivatives w.r.t. (ix, iy), zero where padding
  bool act[kEpiG];
#pragma unroll
  for (int g = 0; g < kEpiG; ++g) {
    act[g] = live && j0 + g < p.d1;
    const int s = level * p.d1 + min(j0 + g, p.d1 - 1);
    const size_t o = ((size_t)b * 2 * D + s) * hw + pix;
    const float u = p.coords[o], v = p.coords[o + (size_t)D * hw];
    const float gx = div_(2.0f * (u + 0.5f), (float)p.w) - 1.0f, gy = div_(2.0f * (v + 0.5f), (float)p.h) - 1.0f;
    const float ix = ((gx + 1.0f) * (float)wl - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)hl - 1.0f) / 2.0f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float xc = fminf(fmaxf(x0f, -2.0f), (float)wl + 1.0f), yc = fminf(fmaxf(y0f, -2.0f), (float)hl + 1.0f);
    const bool wild = !(x0f == xc && y0f == yc);
    const int x0 = (int)xc, y0 = (int)yc, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = ix - x0f, ty = iy - y0f, ex = (x0f + 1.0f) - ix, ey = (y0f + 1.0f) - iy;
    const bool vx0 = x0 >= 0 && x0 < wl, vx1 = x1 >= 0 && x1 < wl, vy0 = y0 >= 0 && y0 < hl, vy1 = y1 >= 0 && y1 < hl;
    const int cx0 = min(max(x0, 0), wl - 1), cx1 = min(max(x1, 0), wl - 1), cy0 = min(max(y0, 0), hl - 1), cy1 = min(max(y1, 0), hl - 1);
    off[g][0] = (unsigned)(cy0 * wl + cx0) * 4u; off[g][1] = (unsigned)(cy0 * wl + cx1) * 4u;
    off[g][2] = (unsigned)(cy1 * wl + cx0) * 4u; off[g][3] = (unsigned)(cy1 * wl + cx1) * 4u;
    const bool v00 = !wild && vx0 && vy0, v10 = !wild && vx1 && vy0, v01 = !wild && vx0 && vy1, v11 = !wild && vx1 && vy1;
    wt[g][0] = v00 ? ex * ey : 0.f; wt[g][1] = v10 ? tx * ey : 0.f; wt[g][2] = v01 ? ex * ty : 0.f; wt[g][3] = v11 ? tx * ty : 0.f;
    dwx[g][0] = v00 ? -ey : 0.f; dwx[g][1] = v10 ? ey : 0.f; dwx[g][2] = v01 ? -ty : 0.f; dwx[g][3] = v11 ? ty : 0.f;
    dwy[g][0] = v00 ? -ex : 0.f; dwy[g][1] = v10 ? -tx : 0.f; dwy[g][2] = v01 ? ex : 0.f; dwy[g][3] = v11 ? tx : 0.f;
  }
  const int cg = p.C / p.heads;
  const float inv_cg = 1.0f / (float)cg;
  float g_ix[kEpiG], g_iy[kEpiG];
#pragma unroll
  for (int g = 0; g < kEpiG; ++g) { g_ix[g] = 0.f; g_iy[g] = 0.f; }
  for (int head = 0; head < p.heads; ++head) {
    float go[kEpiG];  // cotangent of this head's output for each hypothesis, / channels per head
#pragma unroll
    for (int g = 0; g < kEpiG; ++g) {
      const int j = min(j0 + g, p.d1 - 1);
      go[g] = act[g] ? p.g_out[((size_t)b * D * p.heads + (size_t)level * p.heads * p.d1 + (size_t)head * p.d1 + j) * hw + pix] * inv_cg
                     : 0.f;
    }
    for (int c = head * cg; c < (head + 1) * cg; ++c) {
      const size_t plane1 = ((size_t)b * p.C + c) * hw;
      const float f1 = p.fmap1[plane1 + pix];
      const char* pl = reinterpret_cast<const char*>(f2l + ((size_t)b * p.C + c) * hwl);
      char* gpl2 = g_f2l ? reinterpret_cast<char*>(g_f2l + ((size_t)b * p.C + c) * hwl) : nullptr;
      float g_f1 = 0.f;
#pragma unroll
      for (int g = 0; g < kEpiG; ++g) {
        const float a = *reinterpret_cast<const float*>(pl + off[g][0]), bb = *reinterpret_cast<const float*>(pl + off[g][1]);
        const float cc = *reinterpret_cast<const float*>(pl + off[g][2]), d = *reinterpret_cast<const float*>(pl + off[g][3]);
        float o = a * wt[g][0];
        o = fma_(bb, wt[g][1], o);
        o = fma_(cc, wt[g][2], o);
        o = fma_(d, wt[g][3], o);
        const float df = f1 - o;
        const float k = go[g] * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));  // d |f1 - o| / d f1, times the cotangent
        g_f1 += k;
        g_ix[g] -= k * (((a * dwx[g][0] + bb * dwx[g][1]) + cc * dwx[g][2]) + d * dwx[g][3]);
        g_iy[g] -= k * (((a * dwy[g][0] + bb * dwy[g][1]) + cc * dwy[g][2]) + d * dwy[g][3]);
        if (gpl2 && act[g] && k != 0.f) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (wt[g][t] != 0.f) atomicAdd(reinterpret_cast<float*>(gpl2 + off[g][t]), -k * wt[g][t]);
        }
      }
      if (p.g_fmap1 && live && g_f1 != 0.f) atomicAdd(p.g_fmap1 + plane1 + pix, g_f1);
    }
  }
  if (p.g_coords) {
    // ix = ((gx+1) wl - 1)/2 with gx = 2 (u + 0.5)/w - 1  =>  d ix / d u = wl / w
    const float sx = (float)wl / (float)p.w, sy = (float)hl / (float)p.h;
#pragma unroll
    for (int g = 0; g < kEpiG; ++g) {
      if (!act[g]) continue;
      const int s = level * p.d1 + j0 + g;
      const size_t o = ((size_t)b * 2 * D + s) * hw + pix;
      p.g_coords[o] = g_ix[g] * sx;
      p.g_coords[o + (size_t)D * hw] = g_iy[g] * sy;
    }
  }
}

// The same cotangents w.r.t. the two feature maps with NO global atomics: a workgroup owns ONE channel plane of a sample
// -- fmap1's plane and every pyramid level's plane of fmap2 --, walks all pixels and all hypotheses of that sample
// (lane = pixel: coordinates and cotangent are coalesced reads, re-read per channel from the L2), adds what each sample
// sends to its four taps into the level planes held in LDS and writes every plane out once.  d/d fmap1 is a per-thread
// register sum.  The LDS planes are 64-bit FIXED-POINT accumulators (ds_add_u64; least significant bit = max|cotangent|
// / channels x 2^-30, so 2^33 full-magnitude hits fit): scripts/lds_atomic_probe.hip measures ds_add_f32 at 0.33
// lane-adds per cycle per CU whatever the addresses -- one lane at a time --, ds_add_u32 at 13 and ds_add_u64 at 7.5-9;
// and integer addition is associative, so this sum does not depend on the order either.  The global-atomic scatter
// above ran at 36 G lane-adds/s (1.6 G of them at DualRefine's size: 44 ms) and float LDS adds at 216 G/s (7.4 ms); it
// remains the path for planes that do not fit the LDS (48x160 + 24x80 + 12x40 accumulators = 79 KB of the 160).
struct LevelTaps { unsigned o[4]; float w[4]; };
MAL_DEV LevelTaps level_taps(float u, float v, int w, int h, int wl, int hl) {
  LevelTaps t;
  const float gx = div_(2.0f * (u + 0.5f), (float)w) - 1.0f, gy = div_(2.0f * (v + 0.5f), (float)h) - 1.0f;
  const float ix = ((gx + 1.0f) * (float)wl - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)hl - 1.0f) / 2.0f;
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float xc = fminf(fmaxf(x0f, -2.0f), (float)wl + 1.0f), yc = fminf(fmaxf(y0f, -2.0f), (float)hl + 1.0f);
  const bool wild = !(x0f == xc && y0f == yc);
  const int x0 = (int)xc, y0 = (int)yc, x1 = x0 + 1, y1 = y0 + 1;
  const float tx = ix - x0f, ty = iy - y0f, ex = (x0f + 1.0f) - ix, ey = (y0f + 1.0f) - iy;
  const bool vx0 = x0 >= 0 && x0 < wl, vx1 = x1 >= 0 && x1 < wl, vy0 = y0 >= 0 && y0 < hl, vy1 = y1 >= 0 && y1 < hl;
  const int cx0 = min(max(x0, 0), wl - 1), cx1 = min(max(x1, 0), wl - 1), cy0 = min(max(y0, 0), hl - 1), cy1 = min(max(y1, 0), hl - 1);
  t.o[0] = (unsigned)(cy0 * wl + cx0); t.o[1] = (unsigned)(cy0 * wl + cx1);
  t.o[2] = (unsigned)(cy1 * wl + cx0); t.o[3] = (unsigned)(cy1 * wl + cx1);
  t.w[0] = (!wild && vx0 && vy0) ? ex * ey : 0.f;
  t.w[1] = (!wild && vx1 && vy0) ? tx * ey : 0.f;
  t.w[2] = (!wild && vx0 && vy1) ? ex * ty : 0.f;
  t.w[3] = (!wild && vx1 && vy1) ? tx * ty : 0.f;
  return t;
}

constexpr int kPlaneThreads = 512;
constexpr float kFixOne = 1073741824.0f;  // 2^30
MAL_DEV void fix_add(unsigned long long* cell, float v, float to_fix) {
  const long long q = (long long)__float2int_rn(v * to_fix);  // |v * to_fix| <= 2^30
  atomicAdd(cell, (unsigned long long)q);
}
MAL_DEV float fix_value(unsigned long long cell, float lsb) { return (float)((double)(long long)cell * (double)lsb); }
// The fixed-point scale comes from the largest |cotangent| (its float bits order like unsigned integers, and NaN / Inf
// order above every finite value): a non-finite scale means a non-finite cotangent somewhere in the sample / plane.  The
// float-atomic path this replaces (and autograd upstream) would carry it into the result; here the whole plane is
// poisoned with NaN rather than silently written as zeros.
MAL_DEV bool fix_scale_bad(float G) { return !(G < __builtin_huge_valf()); }

// gmax[b] = max |x| over sample b's n values (bits of a non-negative float order like unsigned integers)
__global__ __launch_bounds__(256) void absmax_kernel(const float* x, size_t n, unsigned* gmax) {
  __shared__ unsigned s_m[4];
  const float* xb = x + (size_t)blockIdx.y * n;
  unsigned m = 0u;
  for (size_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) m = max(m, __float_as_uint(fabsf(xb[i])));
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_down((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(gmax + blockIdx.y, max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])));
}

// NC channel planes per workgroup (round 4: NC = 2 where 2 x the planes fit the 160 KB of LDS): the coordinates, the cotangent
// and the tap set of a (pixel, hypothesis) pair are read and derived ONCE for NC channels -- the 128-fold re-read of
// coordinates and cotangent that was 45 % of the one-plane kernel (profiles/r03_epi_bwd_probe.txt) halves.
template <int NC, int THREADS>
__global__ __launch_bounds__(THREADS) void epi_sample_bwd_planes_kernel(EpiSampleBwdParams p, const unsigned* gmax) {
  extern __shared__ unsigned long long s_q[];  // NC x the level planes of d/d fmap2[b, c], back to back, fixed point
  const int c0 = blockIdx.x * NC, b = blockIdx.y, tid = threadIdx.x;
  const int hw = p.h * p.w, D = p.L * p.d1;
  int lo[kEpiMaxLevels + 1];
  lo[0] = 0;
  for (int l = 0; l < p.L; ++l) lo[l + 1] = lo[l] + (p.h >> l) * (p.w >> l);
  const int plane = lo[p.L];
  for (int i = tid; i < NC * plane; i += THREADS) s_q[i] = 0ull;
  __syncthreads();
  const int cg = p.C / p.heads;
  const float inv_cg = 1.0f / (float)cg;
  const float G = __uint_as_float(gmax[b]) * inv_cg;            // no contribution exceeds it (tap weights <= 1)
  const bool bad = fix_scale_bad(G);
  const float to_fix = (G > 0.f && !bad) ? div_(kFixOne, G) : 0.f, lsb = (G > 0.f && !bad) ? div_(G, kFixOne) : 0.f;
  for (int pix = tid; pix < hw; pix += THREADS) {
    float f1[NC], g_f1[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { f1[k] = p.fmap1[((size_t)b * p.C + c0 + k) * hw + pix]; g_f1[k] = 0.f; }
    for (int level = 0; level < p.L; ++level) {
      const int hl = p.h >> level, wl = p.w >> level;
      const float* pl[NC];
      int head[NC];
#pragma unroll
      for (int k = 0; k < NC; ++k) { pl[k] = p.f2[level] + ((size_t)b * p.C + c0 + k) * (size_t)(hl * wl); head[k] = (c0 + k) / cg; }
      const bool want = p.g_f2[level] != nullptr;
#pragma unroll 3
      for (int j = 0; j < p.d1; ++j) {
        const int s_ = level * p.d1 + j;
        const size_t o = ((size_t)b * 2 * D + s_) * hw + pix;
        const float u = p.coords[o], v = p.coords[o + (size_t)D * hw];
        LevelTaps t;
        if (p.probe & 1) {  // experiment: a tap set that costs next to nothing
          const unsigned o0 = (unsigned)min(max((int)v, 0), hl - 2) * (unsigned)wl + (unsigned)min(max((int)u, 0), wl - 2);
          t.o[0] = o0; t.o[1] = o0 + 1; t.o[2] = o0 + wl; t.o[3] = o0 + wl + 1;
          t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0.25f;
        } else t = level_taps(u, v, p.w, p.h, wl, hl);
        float go_prev = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          // (two channels of one head share the cotangent: one load)
          const float go = (k > 0 && head[k] == head[k - 1]) ? go_prev :
              p.g_out[((size_t)b * D * p.heads + (size_t)level * p.heads * p.d1 + (size_t)head[k] * p.d1 + j) * hw + pix] * inv_cg;
          go_prev = go;
          float smp;
          if (p.probe & 4) smp = u * t.w[0];  // experiment: no gathers
          else {
            smp = pl[k][t.o[0]] * t.w[0];
            smp = fma_(pl[k][t.o[1]], t.w[1], smp);
            smp = fma_(pl[k][t.o[2]], t.w[2], smp);
            smp = fma_(pl[k][t.o[3]], t.w[3], smp);
          }
          const float df = f1[k] - smp;
          const float kk = go * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
          g_f1[k] += kk;
          if (want && kk != 0.f && !(p.probe & 2)) {
            unsigned long long* sq = s_q + (size_t)k * plane + lo[level];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (t.w[q] != 0.f) fix_add(sq + t.o[q], -kk * t.w[q], to_fix);
          }
        }
      }
    }
    if (p.g_fmap1)
#pragma unroll
      for (int k = 0; k < NC; ++k) p.g_fmap1[((size_t)b * p.C + c0 + k) * hw + pix] += g_f1[k];
  }
  __syncthreads();
  for (int level = 0; level < p.L; ++level) {
    float* out0 = p.g_f2[level];
    if (!out0) continue;
    const int hwl = (p.h >> level) * (p.w >> level);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      float* out = out0 + ((size_t)b * p.C + c0 + k) * hwl;
      for (int i = tid; i < hwl; i += THREADS) out[i] += bad ? __builtin_nanf("") : fix_value(s_q[(size_t)k * plane + lo[level] + i], lsb);
    }
  }
}

// ---------------------------------------------------------------- pose refinement step (forward)
struct EpiGradCoordParams {
  const float* depth; const float* poses; const float* K;
  int B, h, w;
  float* c_p;  // (B,2,1,5,h,w): projection, +x, -x, +y, -y
  float* P2;   // (B,4,h*w): the transformed points
};

// Reprojections.depth2gradcoords, utils.py:219-236
__global__ __launch_bounds__(256) void epi_gradcoords_kernel(EpiGradCoordParams p) {
  const int hw = p.h * p.w;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * hw) return;
  const int b = i / hw, pix = i - b * hw, y = pix / p.w, x = pix - y * p.w;
  const float* Kb = p.K + b * 16;
  const float* T = p.poses + b * 16;
  const float fx = Kb[0], fy = Kb[5], cx = Kb[2], cy = Kb[6];
  const float Z = p.depth[i];
  const float X = div_((float)x - cx, fx), Y = div_((float)y - cy, fy);
  const float X0[4] = {Z * X, Z * Y, Z, 1.0f};
  float X1[4];
  for (int r_ = 0; r_ < 4; ++r_) {
    float acc = T[r_ * 4] * X0[0];
    acc = fma_(T[r_ * 4 + 1], X0[1], acc);
    acc = fma_(T[r_ * 4 + 2], X0[2], acc);
    X1[r_] = fma_(T[r_ * 4 + 3], X0[3], acc);
    p.P2[((size_t)b * 4 + r_) * hw + pix] = X1[r_];
  }
  const float inv = div_(1.0f, X1[2]);
  const float d = inv > 100.0f ? 100.0f : inv;
  const float u = fx * (X1[0] * d) + cx, v = fy * (X1[1] * d) + cy;
  const float du[5] = {0.f, 1.f, -1.f, 0.f, 0.f}, dv[5] = {0.f, 0.f, 0.f, 1.f, -1.f};
  for (int k = 0; k < 5; ++k) {
    p.c_p[(((size_t)b * 2 + 0) * 5 + k) * hw + pix] = u + du[k];
    p.c_p[(((size_t)b * 2 + 1) * 5 + k) * hw + pix] = v + dv[k];
  }
}

struct EpiAlignParams {
  const float* src; const float* tgt;      // (B,C,h,w)
  const float* src_w; const float* tgt_w;  // (B,1,h,w)
  const float* weight;                     // (B,1,h,w) nullable
  const float* K; const float* p2; const float* P2;
  int B, C, h, w, nblk;                    // nblk = workgroups per sample
  double* partial;                         // [B][nblk][27]
  float* H; float* bvec;                   // (B,36), (B,6)
  int robust;                              // --robust_pose_loss (utils.py:344-355)
  // backward
  const float* g_H; const float* g_b;      // (B,36), (B,6)
  float* g_src; float* g_tgt; float* g_src_w; float* g_tgt_w; float* g_weight; float* g_p2; float* g_P2;
  float* coef;  // (B,6,hw) nullable: d/d(Sxx, Sxy, Syy, Srx, Sry, cost) per pixel, for epi_align_bwd_planes_kernel
};

// --robust_pose_loss: scaled_barron(0, 0.1) of the squared residual (losses.py:8-19,41-90: alpha = 0 -> 2 log1p(x/2), first
// derivative 2 / (x + 2) at x = cost / 0.1**2), times the in-image test of the projected centre with a 2 px margin
// (utils.py:409-412).  -> weight, d weight / d cost
constexpr float kBarronA2 = 0.010000000000000002f;
MAL_DEV float robust_weight(float cost, float u, float v, int h, int w, float* d_cost) {
  const float y = div_(cost, kBarronA2);
  const float wl = div_(2.0f, y + 2.0f);
  const bool valid = u >= 2.0f && u <= (float)(w - 3) && v >= 2.0f && v <= (float)(h - 3);
  *d_cost = valid ? -div_(div_(2.0f, (y + 2.0f) * (y + 2.0f)), kBarronA2) : 0.f;
  return valid ? wl : 0.f;
}

struct Tap4 { unsigned o[4]; float w[4]; float tx, ty, ex, ey; bool v[4]; };  // v: tap inside the map (and the position sane)

// bilinear, zero-padded, align_corners=False sample position (u, v) in pixels of an (h,w) map (corr.py:38-39 /
// utils.py:374-379 + grid_sample's unnormalise)
MAL_DEV Tap4 taps_at(float u, float v, int h, int w) {
  Tap4 t;
  const float gx = div_(2.0f * (u + 0.5f), (float)w) - 1.0f, gy = div_(2.0f * (v + 0.5f), (float)h) - 1.0f;
  const float ix = ((gx + 1.0f) * (float)w - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)h - 1.0f) / 2.0f;
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float xc = fminf(fmaxf(x0f, -2.0f), (float)w + 1.0f), yc = fminf(fmaxf(y0f, -2.0f), (float)h + 1.0f);
  const bool wild = !(x0f == xc && y0f == yc);
  const int x0 = (int)xc, y0 = (int)yc, x1 = x0 + 1, y1 = y0 + 1;
  const float tx = ix - x0f, ty = iy - y0f, ex = (x0f + 1.0f) - ix, ey = (y0f + 1.0f) - iy;
  const bool vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
  const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1), cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
  t.o[0] = (unsigned)(cy0 * w + cx0) * 4u; t.o[1] = (unsigned)(cy0 * w + cx1) * 4u;
  t.o[2] = (unsigned)(cy1 * w + cx0) * 4u; t.o[3] = (unsigned)(cy1 * w + cx1) * 4u;
  t.v[0] = !wild && vx0 && vy0; t.v[1] = !wild && vx1 && vy0; t.v[2] = !wild && vx0 && vy1; t.v[3] = !wild && vx1 && vy1;
  t.w[0] = t.v[0] ? ex * ey : 0.f;
  t.w[1] = t.v[1] ? tx * ey : 0.f;
  t.w[2] = t.v[2] ? ex * ty : 0.f;
  t.w[3] = t.v[3] ? tx * ty : 0.f;
  t.tx = tx; t.ty = ty; t.ex = ex; t.ey = ey;
  return t;
}
MAL_DEV float sample4(const char* plane, const Tap4& t) {
  float o = *reinterpret_cast<const float*>(plane + t.o[0]) * t.w[0];
  o = fma_(*reinterpret_cast<const float*>(plane + t.o[1]), t.w[1], o);
  o = fma_(*reinterpret_cast<const float*>(plane + t.o[2]), t.w[2], o);
  return fma_(*reinterpret_cast<const float*>(plane + t.o[3]), t.w[3], o);
}

// PoseUpdate.direct_align up to the solve (utils.py:303-355, without --robust_pose_loss).  The reference materialises
// J = -J_img J_pix (B,hw,C,6) and a 6x6 matrix per pixel; with J_c = -(gx_c a + gy_c b) (a, b: the two rows of the
// projection Jacobian, gx_c, gy_c: the sampled central differences of channel c) the per-pixel sums over the channels are
//   sum_c J_c J_c^T = Sxx a a^T + Sxy (a b^T + b a^T) + Syy b b^T,   sum_c r_c J_c = -(Srx a + Sry b)
// so five channel sums per pixel suffice (lane = pixel, the channel loop gathers 5 positions x 4 taps), the 27 numbers
// of the weighted normal equations are formed once per pixel and reduced over the sample in fixed order.
__global__ __launch_bounds__(256) void epi_align_kernel(EpiAlignParams p) {
  __shared__ double s_red[4][27];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hw = p.h * p.w, b = blockIdx.y;
  const int pix_raw = blockIdx.x * 256 + threadIdx.x;
  const bool live = pix_raw < hw;
  const int pix = min(pix_raw, hw - 1);
  Tap4 t[5];
#pragma unroll
  for (int k = 0; k < 5; ++k)
    t[k] = taps_at(p.p2[(((size_t)b * 2 + 0) * 5 + k) * hw + pix], p.p2[(((size_t)b * 2 + 1) * 5 + k) * hw + pix], p.h, p.w);
  float Sxx = 0.f, Sxy = 0.f, Syy = 0.f, Srx = 0.f, Sry = 0.f, cost = 0.f;
#pragma unroll 2
  for (int c = 0; c < p.C; ++c) {
    const char* pl = reinterpret_cast<const char*>(p.tgt + ((size_t)b * p.C + c) * hw);
    const float f0 = sample4(pl, t[0]);
    const float gx = (sample4(pl, t[1]) - sample4(pl, t[2])) / 2.0f, gy = (sample4(pl, t[3]) - sample4(pl, t[4])) / 2.0f;
    const float r = p.src[((size_t)b * p.C + c) * hw + pix] - f0;
    Sxx = fma_(gx, gx, Sxx); Sxy = fma_(gx, gy, Sxy); Syy = fma_(gy, gy, Syy);
    Srx = fma_(r, gx, Srx); Sry = fma_(r, gy, Sry);
    cost = fma_(r, r, cost);
  }
  float wgt = p.src_w[(size_t)b * hw + pix] * sample4(reinterpret_cast<const char*>(p.tgt_w + (size_t)b * hw), t[0]);
  if (p.weight) wgt *= p.weight[(size_t)b * hw + pix];
  if (p.robust) {
    float unused;
    wgt *= robust_weight(cost, p.p2[(((size_t)b * 2 + 0) * 5) * hw + pix], p.p2[(((size_t)b * 2 + 1) * 5) * hw + pix], p.h, p.w,
                         &unused);
  }
  if (!live) wgt = 0.f;
  const float X = p.P2[((size_t)b * 4 + 0) * hw + pix], Y = p.P2[((size_t)b * 4 + 1) * hw + pix], Z = p.P2[((size_t)b * 4 + 2) * hw + pix];
  const float fx = p.K[b * 16], fy = p.K[b * 16 + 5];
  const float fxz = div_(fx, Z), fyz = div_(fy, Z);
  const float fxxz2 = div_(fxz * X, Z), fyyz2 = div_(fyz * Y, Z);
  const float a[6] = {fxz, 0.f, -fxxz2, -fxxz2 * Y, fx + fxxz2 * X, -fxz * Y};
  const float q[6] = {0.f, fyz, -fyyz2, -fy - fyyz2 * Y, fyyz2 * X, fyz * X};
  float v27[27];
  int n = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j)
      v27[n++] = wgt * (Sxx * (a[i] * a[j]) + Sxy * (a[i] * q[j] + q[i] * a[j]) + Syy * (q[i] * q[j]));
#pragma unroll
  for (int i = 0; i < 6; ++i) v27[21 + i] = wgt * (Srx * a[i] + Sry * q[i]);
#pragma unroll
  for (int i = 0; i < 27; ++i) {
    const double s = wave_sum_d(live ? (double)v27[i] : 0.0);
    if (lane == 0) s_red[wv][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 27)
    p.partial[((size_t)b * p.nblk + blockIdx.x) * 27 + threadIdx.x] =
        (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void epi_align_finish_kernel(EpiAlignParams p) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ double s_v[27];
  if (tid < 27) {
    double acc = 0.0;
    for (int k = 0; k < p.nblk; ++k) acc += p.partial[((size_t)b * p.nblk + k) * 27 + tid];
    s_v[tid] = acc;
  }
  __syncthreads();
  if (tid < 36) {
    const int i = tid / 6, j = tid % 6, lo = min(i, j), hi = max(i, j);
    const int idx = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);  // position of (lo, hi) in the upper triangle, row-major
    p.H[b * 36 + tid] = (float)s_v[idx];
  }
  if (tid < 6) p.bvec[b * 6 + tid] = (float)s_v[21 + tid];
}

// ---------------------------------------------------------------- pose refinement step (backward)
// Reprojections.depth2gradcoords backward: cotangents of c_p (B,2,1,5,h,w) and P2 (B,4,hw; nullable) -> d/d depth, and per
// workgroup the 16 partial sums of d/d pose (fixed order; epi_rows_finish_kernel adds them)
struct EpiGradCoordBwdParams {
  const float* depth; const float* poses; const float* K; const float* g_cp; const float* g_P2;
  int B, h, w, nblk;
  float* g_depth; double* partial;  // [B][nblk][16]
};

__global__ __launch_bounds__(256) void epi_gradcoords_bwd_kernel(EpiGradCoordBwdParams p) {
  __shared__ double s_red[16][4];
  const int hw = p.h * p.w, b = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x;
  const bool live = pix < hw;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  if (live) {
    const int y = pix / p.w, x = pix - y * p.w, i = b * hw + pix;
    const float* Kb = p.K + b * 16;
    const float* T = p.poses + b * 16;
    const float fx = Kb[0], fy = Kb[5], cx = Kb[2], cy = Kb[6];
    const float Z = p.depth[i];
    const float X = div_((float)x - cx, fx), Y = div_((float)y - cy, fy);
    const float X0[4] = {Z * X, Z * Y, Z, 1.0f};
    float X1[3];
#pragma unroll
    for (int r_ = 0; r_ < 3; ++r_) {
      float a = T[r_ * 4] * X0[0];
      a = fma_(T[r_ * 4 + 1], X0[1], a);
      a = fma_(T[r_ * 4 + 2], X0[2], a);
      X1[r_] = fma_(T[r_ * 4 + 3], X0[3], a);
    }
    const float inv = div_(1.0f, X1[2]);
    const bool clamped = inv > 100.0f;
    const float d = clamped ? 100.0f : inv;
    float gu = 0.f, gv = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      gu += p.g_cp[(((size_t)b * 2 + 0) * 5 + k) * hw + pix];
      gv += p.g_cp[(((size_t)b * 2 + 1) * 5 + k) * hw + pix];
    }
    float gX1[4];
    gX1[0] = gu * fx * d;
    gX1[1] = gv * fy * d;
    const float g_d = gu * fx * X1[0] + gv * fy * X1[1];
    gX1[2] = clamped ? 0.f : -g_d * inv * inv;
    gX1[3] = 0.f;
    if (p.g_P2)
#pragma unroll
      for (int r_ = 0; r_ < 4; ++r_) gX1[r_] += p.g_P2[((size_t)b * 4 + r_) * hw + pix];
    float gZ = 0.f;
#pragma unroll
    for (int r_ = 0; r_ < 4; ++r_) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r_ * 4 + c] = gX1[r_] * X0[c];
      gZ += gX1[r_] * (T[r_ * 4] * X + T[r_ * 4 + 1] * Y + T[r_ * 4 + 2]);
    }
    p.g_depth[i] = gZ;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double v = wave_sum_d((double)acc[k]);
    if (lane == 0) s_red[k][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16)
    p.partial[((size_t)b * p.nblk + blockIdx.x) * 16 + threadIdx.x] =
        ((s_red[threadIdx.x][0] + s_red[threadIdx.x][1]) + s_red[threadIdx.x][2]) + s_red[threadIdx.x][3];
}

// out[b][t] = sum over the workgroups (in order) of partial[b][k][t], t < n
__global__ __launch_bounds__(64) void epi_rows_finish_kernel(const double* partial, int nblk, int n, float* out) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= n) return;
  double a = 0.0;
  for (int k = 0; k < nblk; ++k) a += partial[((size_t)b * nblk + k) * n + t];
  out[b * n + t] = (float)a;
}

// PoseUpdate.direct_align backward up to the solve: cotangents of H (B,36) and b (B,6) -> every input of the normal
// equations.  Same decomposition as the forward (lane = pixel, two walks over the channel planes): the first re-derives
// the five channel sums, from which d/d(weight factors), d/d(the sums) and d/d(the two Jacobian rows) follow in registers;
// the second re-derives each channel's samples and sends d/d(sample k) to the four taps of position k (float atomics, as
// ATen's grid_sampler backward: the one unordered sum) and to the position itself through the bilinear slopes.
//   H = sum_pix w (Sxx a a^T + Sxy (a q^T + q a^T) + Syy q q^T),   b = sum_pix w (Srx a + Sry q)
__global__ __launch_bounds__(256) void epi_align_bwd_kernel(EpiAlignParams p) {
  __shared__ float s_G[36], s_gb[6];
  const int hw = p.h * p.w, b = blockIdx.y;
  if (threadIdx.x < 36) {  // Gs = G + G^T: H is built symmetric, both triangles carry the same sum
    const int i = threadIdx.x / 6, j = threadIdx.x % 6;
    s_G[threadIdx.x] = p.g_H[b * 36 + i * 6 + j] + p.g_H[b * 36 + j * 6 + i];
  }
  if (threadIdx.x < 6) s_gb[threadIdx.x] = p.g_b[b * 6 + threadIdx.x];
  __syncthreads();
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= hw) return;
  Tap4 t[5];
  float uu[5], vv[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    uu[k] = p.p2[(((size_t)b * 2 + 0) * 5 + k) * hw + pix];
    vv[k] = p.p2[(((size_t)b * 2 + 1) * 5 + k) * hw + pix];
    t[k] = taps_at(uu[k], vv[k], p.h, p.w);
  }
  float Sxx = 0.f, Sxy = 0.f, Syy = 0.f, Srx = 0.f, Sry = 0.f, cost = 0.f;
#pragma unroll 2
  for (int c = 0; c < p.C; ++c) {
    const char* pl = reinterpret_cast<const char*>(p.tgt + ((size_t)b * p.C + c) * hw);
    const float f0 = sample4(pl, t[0]);
    const float gx = (sample4(pl, t[1]) - sample4(pl, t[2])) / 2.0f, gy = (sample4(pl, t[3]) - sample4(pl, t[4])) / 2.0f;
    const float r = p.src[((size_t)b * p.C + c) * hw + pix] - f0;
    Sxx = fma_(gx, gx, Sxx); Sxy = fma_(gx, gy, Sxy); Syy = fma_(gy, gy, Syy);
    Srx = fma_(r, gx, Srx); Sry = fma_(r, gy, Sry);
    cost = fma_(r, r, cost);
  }
  const char* twp = reinterpret_cast<const char*>(p.tgt_w + (size_t)b * hw);
  const float sw = p.src_w[(size_t)b * hw + pix], tw = sample4(twp, t[0]);
  const float wt = p.weight ? p.weight[(size_t)b * hw + pix] : 1.0f;
  float rw = 1.0f, drw_dcost = 0.f;
  if (p.robust) rw = robust_weight(cost, uu[0], vv[0], p.h, p.w, &drw_dcost);
  const float wgt = ((sw * tw) * wt) * rw;
  const float X = p.P2[((size_t)b * 4 + 0) * hw + pix], Y = p.P2[((size_t)b * 4 + 1) * hw + pix], Z = p.P2[((size_t)b * 4 + 2) * hw + pix];
  const float fx = p.K[b * 16], fy = p.K[b * 16 + 5];
  const float fxz = div_(fx, Z), fyz = div_(fy, Z);
  const float fxxz2 = div_(fxz * X, Z), fyyz2 = div_(fyz * Y, Z);
  const float a[6] = {fxz, 0.f, -fxxz2, -fxxz2 * Y, fx + fxxz2 * X, -fxz * Y};
  const float q[6] = {0.f, fyz, -fyyz2, -fy - fyyz2 * Y, fyyz2 * X, fyz * X};
  float ua[6], uq[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) { s0 = fma_(s_G[i * 6 + j], a[j], s0); s1 = fma_(s_G[i * 6 + j], q[j], s1); }
    ua[i] = s0; uq[i] = s1;
  }
  float a_ua = 0.f, a_uq = 0.f, q_uq = 0.f, gb_a = 0.f, gb_q = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    a_ua = fma_(a[i], ua[i], a_ua); a_uq = fma_(a[i], uq[i], a_uq); q_uq = fma_(q[i], uq[i], q_uq);
    gb_a = fma_(s_gb[i], a[i], gb_a); gb_q = fma_(s_gb[i], q[i], gb_q);
  }
  // d / d (the pixel's weight), d / d (the five channel sums)
  const float dw = (Sxx * (0.5f * a_ua) + Sxy * a_uq + Syy * (0.5f * q_uq)) + (Srx * gb_a + Sry * gb_q);
  const float dSxx = wgt * 0.5f * a_ua, dSxy = wgt * a_uq, dSyy = wgt * 0.5f * q_uq, dSrx = wgt * gb_a, dSry = wgt * gb_q;
  // d / d (the Jacobian rows) -> d / d (X, Y, Z)
  float da[6], dq[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    da[i] = wgt * ((Sxx * ua[i] + Sxy * uq[i]) + Srx * s_gb[i]);
    dq[i] = wgt * ((Sxy * ua[i] + Syy * uq[i]) + Sry * s_gb[i]);
  }
  float d_fxz = da[0] - Y * da[5], d_fxxz2 = (-da[2] - Y * da[3]) + X * da[4];
  float d_fyz = dq[1] + X * dq[5], d_fyyz2 = (-dq[2] - Y * dq[3]) + X * dq[4];
  float dX = fxxz2 * da[4] + (fyyz2 * dq[4] + fyz * dq[5]);
  float dY = (-fxxz2 * da[3] - fxz * da[5]) - fyyz2 * dq[3];
  const float iz = div_(1.0f, Z);
  d_fxz += d_fxxz2 * X * iz; dX += d_fxxz2 * fxz * iz;
  d_fyz += d_fyyz2 * Y * iz; dY += d_fyyz2 * fyz * iz;
  const float dZ = -(d_fxxz2 * fxxz2 + d_fyyz2 * fyyz2 + d_fxz * fxz + d_fyz * fyz) * iz;
  if (p.g_P2) {
    p.g_P2[((size_t)b * 4 + 0) * hw + pix] = dX; p.g_P2[((size_t)b * 4 + 1) * hw + pix] = dY;
    p.g_P2[((size_t)b * 4 + 2) * hw + pix] = dZ; p.g_P2[((size_t)b * 4 + 3) * hw + pix] = 0.f;
  }
  // the factors of the weight
  if (p.g_src_w) p.g_src_w[(size_t)b * hw + pix] = dw * ((tw * wt) * rw);
  if (p.g_weight) p.g_weight[(size_t)b * hw + pix] = dw * ((sw * tw) * rw);
  const float dtw = dw * ((sw * wt) * rw);
  const float dcost = dw * ((sw * tw) * wt) * drw_dcost;
  if (p.coef) {
    const float cf[6] = {dSxx, dSxy, dSyy, dSrx, dSry, dcost};
#pragma unroll
    for (int k = 0; k < 6; ++k) p.coef[((size_t)b * 6 + k) * hw + pix] = cf[k];
  }
  float gu[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, gv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  auto slopes = [&](const char* plane, const Tap4& tp, float* dix, float* diy) {  // of the zero-padded bilinear sample
    const float v00 = tp.v[0] ? *reinterpret_cast<const float*>(plane + tp.o[0]) : 0.f;
    const float v01 = tp.v[1] ? *reinterpret_cast<const float*>(plane + tp.o[1]) : 0.f;
    const float v10 = tp.v[2] ? *reinterpret_cast<const float*>(plane + tp.o[2]) : 0.f;
    const float v11 = tp.v[3] ? *reinterpret_cast<const float*>(plane + tp.o[3]) : 0.f;
    *dix = (v01 - v00) * tp.ey + (v11 - v10) * tp.ty;
    *diy = (v10 - v00) * tp.ex + (v11 - v01) * tp.tx;
  };
  auto scatter = [&](float* plane, const Tap4& tp, float g) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (tp.v[j] && g != 0.f) atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(plane) + tp.o[j]), g * tp.w[j]);
  };
  {
    float dix, diy;
    slopes(twp, t[0], &dix, &diy);
    gu[0] = dtw * dix; gv[0] = dtw * diy;
    if (p.g_tgt_w) scatter(p.g_tgt_w + (size_t)b * hw, t[0], dtw);
  }
  for (int c = 0; c < p.C; ++c) {
    const size_t po = ((size_t)b * p.C + c) * hw;
    const char* pl = reinterpret_cast<const char*>(p.tgt + po);
    const float f0 = sample4(pl, t[0]);
    const float gx = (sample4(pl, t[1]) - sample4(pl, t[2])) / 2.0f, gy = (sample4(pl, t[3]) - sample4(pl, t[4])) / 2.0f;
    const float r = p.src[po + pix] - f0;
    const float dgx = (2.0f * gx * dSxx + gy * dSxy) + r * dSrx, dgy = (2.0f * gy * dSyy + gx * dSxy) + r * dSry;
    const float dr = (gx * dSrx + gy * dSry) + 2.0f * r * dcost;
    if (p.g_src) p.g_src[po + pix] = dr;
    const float df[5] = {-dr, 0.5f * dgx, -0.5f * dgx, 0.5f * dgy, -0.5f * dgy};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float dix, diy;
      slopes(pl, t[k], &dix, &diy);
      gu[k] = fma_(df[k], dix, gu[k]); gv[k] = fma_(df[k], diy, gv[k]);
      if (p.g_tgt) scatter(p.g_tgt + po, t[k], df[k]);
    }
  }
  if (p.g_p2)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      p.g_p2[(((size_t)b * 2 + 0) * 5 + k) * hw + pix] = gu[k];
      p.g_p2[(((size_t)b * 2 + 1) * 5 + k) * hw + pix] = gv[k];
    }
}

// d/d tgt_feat without global atomics (see epi_sample_bwd_planes_kernel): a workgroup owns one channel plane of a
// sample, re-derives for every pixel the five samples of that channel and -- with the per-pixel coefficients the kernel
// above left in `coef` -- what each sends to its four taps, accumulated in LDS (48x160 floats = 30 KB), written once.
__global__ __launch_bounds__(kPlaneThreads) void epi_align_bwd_planes_kernel(EpiAlignParams p) {
  extern __shared__ unsigned long long s_q[];  // fixed point, see epi_sample_bwd_planes_kernel
  __shared__ unsigned s_m[kPlaneThreads / 64];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, hw = p.h * p.w;
  for (int i = tid; i < hw; i += kPlaneThreads) s_q[i] = 0ull;
  const size_t po = ((size_t)b * p.C + c) * hw;
  const char* pl = reinterpret_cast<const char*>(p.tgt + po);
  // two walks over the plane's pixels: the largest |d/d sample| first (the fixed-point scale), then the accumulation
  float to_fix = 0.f, lsb = 0.f;
  bool bad = false;
  unsigned m = 0u;
  for (int pass = 0; pass < 2; ++pass) {
    for (int pix = tid; pix < hw; pix += kPlaneThreads) {
      Tap4 t[5];
#pragma unroll
      for (int k = 0; k < 5; ++k)
        t[k] = taps_at(p.p2[(((size_t)b * 2 + 0) * 5 + k) * hw + pix], p.p2[(((size_t)b * 2 + 1) * 5 + k) * hw + pix], p.h, p.w);
      float cf[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) cf[k] = p.coef[((size_t)b * 6 + k) * hw + pix];
      const float f0 = sample4(pl, t[0]);
      const float gx = (sample4(pl, t[1]) - sample4(pl, t[2])) / 2.0f, gy = (sample4(pl, t[3]) - sample4(pl, t[4])) / 2.0f;
      const float r = p.src[po + pix] - f0;
      const float dgx = (2.0f * gx * cf[0] + gy * cf[1]) + r * cf[3], dgy = (2.0f * gy * cf[2] + gx * cf[1]) + r * cf[4];
      const float dr = (gx * cf[3] + gy * cf[4]) + 2.0f * r * cf[5];
      if (pass == 0) {
        // unsigned max of the bit patterns (not fmaxf, which drops a NaN operand): a non-finite term must win
        m = max(max(m, __float_as_uint(fabsf(dr))), max(__float_as_uint(0.5f * fabsf(dgx)), __float_as_uint(0.5f * fabsf(dgy))));
      } else {
        const float df[5] = {-dr, 0.5f * dgx, -0.5f * dgx, 0.5f * dgy, -0.5f * dgy};
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (t[k].v[j] && df[k] != 0.f) fix_add(s_q + (t[k].o[j] >> 2), df[k] * t[k].w[j], to_fix);
      }
    }
    if (pass == 0) {
      for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_down((int)m, o, 64));
      if ((tid & 63) == 0) s_m[tid >> 6] = m;
      __syncthreads();
      unsigned mm = 0u;
      for (int k = 0; k < kPlaneThreads / 64; ++k) mm = max(mm, s_m[k]);
      const float G = __uint_as_float(mm);
      bad = fix_scale_bad(G);
      to_fix = (G > 0.f && !bad) ? div_(kFixOne, G) : 0.f;
      lsb = (G > 0.f && !bad) ? div_(G, kFixOne) : 0.f;
    }
  }
  __syncthreads();
  for (int i = tid; i < hw; i += kPlaneThreads) p.g_tgt[po + i] += bad ? __builtin_nanf("") : fix_value(s_q[i], lsb);
}

// x = A^-1 rhs as PoseUpdate.direct_align does it (utils.py:357-368): Cholesky, else LU with partial pivoting, else failure.
// `chol` says which succeeded.  Constant loop bounds, fully unrolled: the 6x6 arrays stay in registers.
MAL_DEV bool solve6(const float (&A)[6][6], const float (&rhs)[6], float (&x)[6], bool* chol) {
  float Lm[6][6];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= Lm[j][k] * Lm[j][k];
    ok = ok && (d > 0.f);
    const float ljj = sqrtf(d);
    Lm[j][j] = ljj;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= Lm[i][k] * Lm[j][k];
      Lm[i][j] = v / ljj;
    }
  }
  {
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float v = rhs[i];
#pragma unroll
      for (int k = 0; k < i; ++k) v -= Lm[i][k] * y[k];
      y[i] = v / Lm[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      float v = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) v -= Lm[k][i] * x[k];
      x[i] = v / Lm[i][i];
    }
  }
  bool solved = ok;
#pragma unroll
  for (int i = 0; i < 6; ++i) solved = solved && (x[i] == x[i]);  // NaN in the factor: fall through (utils.py:358-360)
  *chol = solved;
  if (!solved) {  // torch.linalg.solve: LU with partial pivoting (row swaps as selects, so the indices stay static)
    float M[6][7];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = 0; j < 6; ++j) M[i][j] = A[i][j];
      M[i][6] = rhs[i];
    }
    solved = true;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      int piv = c;
      float best = fabsf(M[c][c]);
#pragma unroll
      for (int r_ = c + 1; r_ < 6; ++r_) {
        const float v = fabsf(M[r_][c]);
        if (v > best) { best = v; piv = r_; }
      }
      solved = solved && (best > 0.f);
#pragma unroll
      for (int r_ = c + 1; r_ < 6; ++r_)
        if (piv == r_) {
#pragma unroll
          for (int j = 0; j < 7; ++j) { const float tmp = M[c][j]; M[c][j] = M[r_][j]; M[r_][j] = tmp; }
        }
#pragma unroll
      for (int r_ = c + 1; r_ < 6; ++r_) {
        const float f = M[r_][c] / M[c][c];
#pragma unroll
        for (int j = c; j < 7; ++j) M[r_][j] -= f * M[c][j];
      }
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      float v = M[i][6];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) v -= M[i][k] * x[k];
      x[i] = v / M[i][i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) solved = solved && (x[i] == x[i]);
  }
  return solved;
}

// se3_exp (dualrefine/layers.py:29-55) on any scalar type: float for the forward, Dual6 (value + six partials, forward-mode)
// for the Jacobian the backward needs -- the same code, so the two cannot drift apart
struct Dual6 {
  float v; float d[6];
};
MAL_DEV Dual6 dual_const(float v) { Dual6 r; r.v = v; for (int k = 0; k < 6; ++k) r.d[k] = 0.f; return r; }
MAL_DEV Dual6 dual_var(float v, int i) { Dual6 r = dual_const(v); r.d[i] = 1.0f; return r; }
MAL_DEV Dual6 operator+(const Dual6& a, const Dual6& b) { Dual6 r; r.v = a.v + b.v; for (int k = 0; k < 6; ++k) r.d[k] = a.d[k] + b.d[k]; return r; }
MAL_DEV Dual6 operator-(const Dual6& a, const Dual6& b) { Dual6 r; r.v = a.v - b.v; for (int k = 0; k < 6; ++k) r.d[k] = a.d[k] - b.d[k]; return r; }
MAL_DEV Dual6 operator-(const Dual6& a) { Dual6 r; r.v = -a.v; for (int k = 0; k < 6; ++k) r.d[k] = -a.d[k]; return r; }
MAL_DEV Dual6 operator*(const Dual6& a, const Dual6& b) { Dual6 r; r.v = a.v * b.v; for (int k = 0; k < 6; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k]; return r; }
MAL_DEV Dual6 operator/(const Dual6& a, const Dual6& b) {
  Dual6 r; r.v = a.v / b.v;
  for (int k = 0; k < 6; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) / b.v;
  return r;
}
MAL_DEV Dual6 sqrt_(const Dual6& a) { Dual6 r; r.v = sqrtf(a.v); for (int k = 0; k < 6; ++k) r.d[k] = a.d[k] / (2.0f * r.v); return r; }
MAL_DEV Dual6 sin_(const Dual6& a) { Dual6 r; r.v = sinf(a.v); const float c = cosf(a.v); for (int k = 0; k < 6; ++k) r.d[k] = c * a.d[k]; return r; }
MAL_DEV Dual6 cos_(const Dual6& a) { Dual6 r; r.v = cosf(a.v); const float s = -sinf(a.v); for (int k = 0; k < 6; ++k) r.d[k] = s * a.d[k]; return r; }
MAL_DEV float sqrt_(float a) { return sqrtf(a); }
MAL_DEV float sin_(float a) { return sinf(a); }
MAL_DEV float cos_(float a) { return cosf(a); }
MAL_DEV float lit(float, float v) { return v; }
MAL_DEV Dual6 lit(const Dual6&, float v) { return dual_const(v); }

// T[:3,:4] of exp(x): rotation R (Rodrigues), translation J rho
template <class S>
MAL_DEV void se3_exp_rows(const S (&x)[6], S (&T)[3][4]) {
  const S zero = lit(x[0], 0.f), one = lit(x[0], 1.f);
  const S theta = sqrt_(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
  const S a[3] = {x[3] / theta, x[4] / theta, x[5] / theta};
  const S sk[3][3] = {{zero, -a[2], a[1]}, {a[2], zero, -a[0]}, {-a[1], a[0], zero}};
  const S ct = cos_(theta), st = sin_(theta), sot = st / theta, omc = one - ct, omcot = omc / theta;
  S J[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const S aat = a[i] * a[j];
      if (i == j) { T[i][j] = ct + omc * aat + st * sk[i][j]; J[i][j] = sot + (one - sot) * aat + omcot * sk[i][j]; }
      else        { T[i][j] = omc * aat + st * sk[i][j];      J[i][j] = (one - sot) * aat + omcot * sk[i][j]; }
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) T[i][3] = J[i][0] * x[0] + J[i][1] * x[1] + J[i][2] * x[2];
}

// The rest of PoseUpdate.direct_align (utils.py:357-368) for one sample per thread: the 6x6 solve with upstream's
// fall-backs, se3_exp of the update and new_pose = exp(update) @ pose.  fp32 like torch.
__global__ void epi_align_update_kernel(const float* H, const float* bvec, const float* poses, int B, float* new_poses,
                                        float* update) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float A[6][6], rhs[6], x[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    rhs[i] = bvec[b * 6 + i];
#pragma unroll
    for (int j = 0; j < 6; ++j) A[i][j] = H[b * 36 + i * 6 + j];
  }
  bool chol;
  const bool solved = solve6(A, rhs, x, &chol);
  const float* P = poses + b * 16;
  if (!solved) {  // utils.py:364-365: return poses, poses
    for (int i = 0; i < 16; ++i) new_poses[b * 16 + i] = P[i];
    for (int i = 0; i < 6; ++i) update[b * 6 + i] = 0.f;
    return;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) update[b * 6 + i] = x[i];
  float T[3][4];
  se3_exp_rows<float>(x, T);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += T[i][k] * P[k * 4 + j];
      new_poses[b * 16 + i * 4 + j] = acc;
    }
    new_poses[b * 16 + 12 + j] = P[12 + j];  // row 3 of exp(update) is (0,0,0,1)
  }
}

// Its VJP, one sample per thread: cotangents of (new_poses, update) -> d/dH (B,36), d/db (B,6), d/dposes (B,16).
//   new = E P:  g_E = g_new P^T (rows 0..2; row 3 of E is constant), g_P = E^T g_new
//   E = se3_exp(x): g_x[k] = sum_ij g_E[i][j] dE[i][j]/dx[k] (forward-mode through the same code) + g_update[k]
//   x = H^-1 b:  g_b = H^-T g_x;  g_H = -(g_b x^T), symmetrised on the Cholesky path as torch.linalg.cholesky's backward does
// The failure path of the forward (poses returned unchanged) passes g_new through.
__global__ void epi_align_update_bwd_kernel(const float* H, const float* bvec, const float* poses, const float* g_new,
                                            const float* g_update, int B, float* g_H, float* g_b, float* g_poses) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float A[6][6], rhs[6], x[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    rhs[i] = bvec[b * 6 + i];
#pragma unroll
    for (int j = 0; j < 6; ++j) A[i][j] = H[b * 36 + i * 6 + j];
  }
  bool chol;
  const bool solved = solve6(A, rhs, x, &chol);
  const float* P = poses + b * 16;
  const float* gn = g_new + b * 16;
  if (!solved) {
    for (int i = 0; i < 16; ++i) g_poses[b * 16 + i] = gn[i];
    for (int i = 0; i < 36; ++i) g_H[b * 36 + i] = 0.f;
    for (int i = 0; i < 6; ++i) g_b[b * 6 + i] = 0.f;
    return;
  }
  Dual6 xd[6], Td[3][4];
#pragma unroll
  for (int k = 0; k < 6; ++k) xd[k] = dual_var(x[k], k);
  se3_exp_rows<Dual6>(xd, Td);
  float gx[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) gx[k] = g_update ? g_update[b * 6 + k] : 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gE = 0.f;  // (g_new P^T)[i][j]
#pragma unroll
      for (int k = 0; k < 4; ++k) gE += gn[i * 4 + k] * P[j * 4 + k];
#pragma unroll
      for (int k = 0; k < 6; ++k) gx[k] += gE * Td[i][j].d[k];
    }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = k == 3 ? gn[12 + j] : 0.f;  // row 3 of E is (0,0,0,1)
#pragma unroll
      for (int i = 0; i < 3; ++i) acc += Td[i][k].v * gn[i * 4 + j];
      g_poses[b * 16 + k * 4 + j] = acc;
    }
  float y[6];
  bool chol2;
  solve6(A, gx, y, &chol2);  // H is symmetric: H^-T = H^-1
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    g_b[b * 6 + i] = y[i];
#pragma unroll
    for (int j = 0; j < 6; ++j)
      g_H[b * 36 + i * 6 + j] = chol ? -0.5f * (y[i] * x[j] + x[i] * y[j]) : -(y[i] * x[j]);
  }
}

}  // namespace mal

using namespace mal;

// LDS a workgroup of `kernel` may ask for: the device's per-block limit (160 KB on gfx950), raised from the default 64 KB
static size_t plane_lds_limit(const void* kernel) {
  static size_t limit = 0;
  if (limit == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
      limit = (size_t)v;
    else
      limit = 64 * 1024;
  }
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)limit) != hipSuccess) {
    (void)hipGetLastError();
    return limit < (size_t)64 * 1024 ? limit : (size_t)64 * 1024;  // the default stays in force: larger planes take the atomic path
  }
  return limit;
}


extern "C" int mal_direct_align_update(const float* H, const float* b, const float* poses, int B, float* new_poses,
                                       float* update, void* stream) {
  if (B <= 0) return MAL_ESHAPE;
  if (!H || !b || !poses || !new_poses || !update) return MAL_EINVAL;
  hipLaunchKernelGGL(epi_align_update_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, H, b, poses, B, new_poses,
                     update);
  return launch_status();
}


extern "C" int mal_direct_align_update_bwd(const float* H, const float* b, const float* poses, const float* g_new_poses,
                                           const float* g_update, int B, float* g_H, float* g_b, float* g_poses, void* stream) {
  if (B <= 0) return MAL_ESHAPE;
  if (!H || !b || !poses || !g_new_poses || !g_H || !g_b || !g_poses) return MAL_EINVAL;
  hipLaunchKernelGGL(epi_align_update_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, H, b, poses, g_new_poses,
                     g_update, B, g_H, g_b, g_poses);
  return launch_status();
}

extern "C" size_t mal_epipolar_gradcoords_bwd_workspace_bytes(int B, int h, int w) {
  if (B <= 0 || h <= 0 || w <= 0) return 0;
  return (size_t)B * ((h * w + 255) / 256) * 16 * sizeof(double);
}

extern "C" int mal_epipolar_gradcoords_bwd(const float* depth, const float* poses, const float* K, const float* g_c_p,
                                           const float* g_P2, int B, int h, int w, float* g_depth, float* g_poses, void* ws,
                                           size_t ws_bytes, void* stream) {
  if (B <= 0 || h < 1 || w < 1 || (double)B * 10 * h * w > 2.0e9 / 4) return MAL_ESHAPE;
  if (!depth || !poses || !K || !g_c_p || !g_depth || !g_poses || !ws) return MAL_EINVAL;
  if (ws_bytes < mal_epipolar_gradcoords_bwd_workspace_bytes(B, h, w)) return MAL_EWORKSPACE;
  EpiGradCoordBwdParams p = {depth, poses, K, g_c_p, g_P2, B, h, w, (h * w + 255) / 256, g_depth, (double*)ws};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(epi_gradcoords_bwd_kernel, dim3(p.nblk, B), dim3(256), 0, st, p);
  hipLaunchKernelGGL(epi_rows_finish_kernel, dim3(B), dim3(64), 0, st, p.partial, p.nblk, 16, g_poses);
  return launch_status();
}

extern "C" int mal_epipolar_gradcoords(const float* depth, const float* poses, const float* K, int B, int h, int w, float* c_p,
                                       float* P2, void* stream) {
  if (B <= 0 || h < 1 || w < 1 || (double)B * 10 * h * w > 2.0e9 / 4) return MAL_ESHAPE;
  if (!depth || !poses || !K || !c_p || !P2) return MAL_EINVAL;
  EpiGradCoordParams p = {depth, poses, K, B, h, w, c_p, P2};
  hipLaunchKernelGGL(epi_gradcoords_kernel, dim3((B * h * w + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

extern "C" size_t mal_direct_align_workspace_bytes(int B, int h, int w) {
  if (B <= 0 || h <= 0 || w <= 0) return 0;
  return (size_t)B * ((h * w + 255) / 256) * 27 * sizeof(double);
}

extern "C" size_t mal_direct_align_bwd_workspace_bytes(int B, int h, int w) {
  if (B <= 0 || h <= 0 || w <= 0) return 0;
  return (size_t)B * 6 * h * w * sizeof(float);
}

extern "C" int mal_direct_align_normal_eq_bwd(const float* src_feat, const float* tgt_feat, const float* src_w,
                                              const float* tgt_w, const float* weight, const float* K, const float* p2,
                                              const float* P2, const float* g_H, const float* g_b, int B, int C, int h, int w,
                                              int robust, float* g_src_feat, float* g_tgt_feat, float* g_src_w, float* g_tgt_w,
                                              float* g_weight, float* g_p2, float* g_P2, void* ws, size_t ws_bytes,
                                              void* stream) {
  if (B <= 0 || C < 1 || h < 1 || w < 1 || (double)B * C * h * w > 2.0e9 / 4) return MAL_ESHAPE;
  if (!src_feat || !tgt_feat || !src_w || !tgt_w || !K || !p2 || !P2 || !g_H || !g_b) return MAL_EINVAL;
  if (g_weight && !weight) return MAL_EINVAL;
  // d/d tgt_feat through LDS planes when a plane fits and the caller handed the per-pixel coefficient scratch
  const bool planes = g_epi_bwd_planes && g_tgt_feat && ws && ws_bytes >= mal_direct_align_bwd_workspace_bytes(B, h, w) &&
                      (size_t)h * w * sizeof(unsigned long long) <= plane_lds_limit((const void*)epi_align_bwd_planes_kernel);
  EpiAlignParams p = {};
  p.src = src_feat; p.tgt = tgt_feat; p.src_w = src_w; p.tgt_w = tgt_w; p.weight = weight; p.K = K; p.p2 = p2; p.P2 = P2;
  p.B = B; p.C = C; p.h = h; p.w = w; p.nblk = (h * w + 255) / 256; p.robust = robust ? 1 : 0;
  p.g_H = g_H; p.g_b = g_b; p.g_src = g_src_feat; p.g_tgt = g_tgt_feat; p.g_src_w = g_src_w; p.g_tgt_w = g_tgt_w;
  p.g_weight = g_weight; p.g_p2 = g_p2; p.g_P2 = g_P2;
  hipStream_t st = (hipStream_t)stream;
  if (planes) { p.coef = (float*)ws; p.g_tgt = nullptr; }
  hipLaunchKernelGGL(epi_align_bwd_kernel, dim3(p.nblk, B), dim3(256), 0, st, p);
  if (planes) {
    p.g_tgt = g_tgt_feat;
    hipLaunchKernelGGL(epi_align_bwd_planes_kernel, dim3(C, B), dim3(kPlaneThreads), (size_t)h * w * sizeof(unsigned long long), st, p);
  }
  return launch_status();
}

extern "C" int mal_direct_align_normal_eq(const float* src_feat, const float* tgt_feat, const float* src_w, const float* tgt_w,
                                          const float* weight, const float* K, const float* p2, const float* P2, int B, int C,
                                          int h, int w, int robust, float* H, float* b, void* ws, size_t ws_bytes, void* stream) {
  if (B <= 0 || C < 1 || h < 1 || w < 1 || (double)B * C * h * w > 2.0e9 / 4) return MAL_ESHAPE;
  if (!src_feat || !tgt_feat || !src_w || !tgt_w || !K || !p2 || !P2 || !H || !b || !ws) return MAL_EINVAL;
  if (ws_bytes < mal_direct_align_workspace_bytes(B, h, w)) return MAL_EWORKSPACE;
  EpiAlignParams p = {};
  p.src = src_feat; p.tgt = tgt_feat; p.src_w = src_w; p.tgt_w = tgt_w; p.weight = weight; p.K = K; p.p2 = p2; p.P2 = P2;
  p.robust = robust ? 1 : 0;
  p.B = B; p.C = C; p.h = h; p.w = w; p.nblk = (h * w + 255) / 256; p.partial = (double*)ws; p.H = H; p.bvec = b;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(epi_align_kernel, dim3(p.nblk, B), dim3(256), 0, st, p);
  hipLaunchKernelGGL(epi_align_finish_kernel, dim3(B), dim3(64), 0, st, p);
  return launch_status();
}


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

extern "C" int mal_epipolar_coords_of_depths(const float* depths, const float* poses, const float* K, int B, int D, int h,
                                            int w, float* coords, void* stream) {
  if (B <= 0 || D < 1 || h < 1 || w < 1 || (double)B * 2 * D * h * w > 2.0e9 / 4) return MAL_ESHAPE;
  if (!depths || !poses || !K || !coords) return MAL_EINVAL;
  const size_t n = (size_t)B * D * h * w;
  hipLaunchKernelGGL(epi_coords_of_depths_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, depths,
                     poses, K, B, D, h, w, coords);
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

extern "C" size_t mal_epipolar_coords_bwd_workspace_bytes(int B, int h, int w) {
  if (B <= 0 || h < 1 || w < 1) return 0;
  return align256((size_t)B * ((h * w + 255) / 256) * 13 * sizeof(double));
}

extern "C" int mal_epipolar_coords_bwd(const float* depth, const float* poses, const float* K, const float* g_coords,
                                       const float* g_max_dx, const float* g_depths, int B, int h, int w, int r, int L,
                                       float softplus_delta, float ratio, float* g_depth, float* g_poses, float* g_dd,
                                       void* ws, size_t ws_bytes, void* stream) {
  if (B <= 0 || h < 1 || w < 1 || r < 1 || L < 1 || L > kEpiMaxLevels) return MAL_ESHAPE;
  if ((double)B * 2 * L * (2 * r + 1) * h * w > 2.0e9 / 4) return MAL_ESHAPE;
  if (!depth || !poses || !K || !g_coords || !g_depth || !g_poses || !g_dd || !ws) return MAL_EINVAL;
  if (ws_bytes < mal_epipolar_coords_bwd_workspace_bytes(B, h, w)) return MAL_EWORKSPACE;
  EpiCoordBwdParams p = {};
  p.depth = depth; p.poses = poses; p.K = K; p.g_coords = g_coords; p.g_max_dx = g_max_dx; p.g_depths = g_depths;
  p.B = B; p.h = h; p.w = w; p.r = r; p.L = L; p.nblk = (h * w + 255) / 256; p.dd = softplus_delta; p.ratio = ratio;
  p.g_depth = g_depth; p.partial = (double*)ws;
  hipLaunchKernelGGL(epi_coords_bwd_kernel, dim3(p.nblk, B), dim3(256), 0, (hipStream_t)stream, p);
  hipLaunchKernelGGL(epi_coords_bwd_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p.partial, B, p.nblk, g_poses, g_dd);
  return launch_status();
}

extern "C" size_t mal_coord_sample_l1_bwd_workspace_bytes(int B) { return B > 0 ? align256((size_t)B * sizeof(unsigned)) : 0; }

extern "C" int mal_coord_sample_l1_bwd(const float* fmap1, const float* const* f2_pyramid, const float* coords,
                                       const float* g_out, int B, int C, int h, int w, int L, int d1, int heads,
                                       float* g_fmap1, float* const* g_f2_pyramid, float* g_coords, void* ws, size_t ws_bytes,
                                       void* stream) {
  if (B <= 0 || C < 1 || h < 1 || w < 1 || L < 1 || L > kEpiMaxLevels || d1 < 1 || heads < 1 || C % heads) return MAL_ESHAPE;
  if ((h >> (L - 1)) < 1 || (w >> (L - 1)) < 1) return MAL_ESHAPE;
  if ((double)B * C * h * w > 2.0e9 / 4 || (double)B * 2 * L * d1 * h * w > 2.0e9 / 4) return MAL_ESHAPE;
  if (!fmap1 || !f2_pyramid || !coords || !g_out) return MAL_EINVAL;
  EpiSampleBwdParams p = {};
  p.fmap1 = fmap1; p.coords = coords; p.g_out = g_out; p.B = B; p.C = C; p.h = h; p.w = w; p.L = L; p.d1 = d1; p.heads = heads;
  p.g_fmap1 = g_fmap1; p.g_coords = g_coords; p.probe = g_epi_probe;
  for (int l = 0; l < L; ++l) {
    if (!f2_pyramid[l]) return MAL_EINVAL;
    p.f2[l] = f2_pyramid[l];
    p.g_f2[l] = g_f2_pyramid ? g_f2_pyramid[l] : nullptr;
  }
  hipStream_t st = (hipStream_t)stream;
  size_t plane_floats = 0;
  bool any_f2 = false;
  for (int l = 0; l < L; ++l) { plane_floats += (size_t)(h >> l) * (w >> l); any_f2 = any_f2 || p.g_f2[l]; }
  const size_t lds = plane_floats * sizeof(unsigned long long);
  const bool planes = g_epi_bwd_planes && (g_fmap1 || any_f2) && ws && ws_bytes >= (size_t)B * sizeof(unsigned) &&
                      lds <= plane_lds_limit((const void*)epi_sample_bwd_planes_kernel<1, kPlaneThreads>);
  // two channel planes per workgroup where both fit (option "epi_bwd_planes" 2, default; 1: one plane, round 3)
  const bool planes2 = planes && g_epi_bwd_planes >= 2 && (C % 2) == 0 &&
                       2 * lds <= plane_lds_limit((const void*)epi_sample_bwd_planes_kernel<2, 1024>);
  if (planes) {
    // d/d fmap1, d/d fmap2 levels: one workgroup per (sample, channel), LDS accumulation; d/d coords: the sweep below
    unsigned* gmax = (unsigned*)ws;
    (void)hipMemsetAsync(gmax, 0, (size_t)B * sizeof(unsigned), st);
    const size_t n = (size_t)L * heads * d1 * h * w;
    hipLaunchKernelGGL(absmax_kernel, dim3(64, B), dim3(256), 0, st, g_out, n, gmax);
    if (planes2) hipLaunchKernelGGL((epi_sample_bwd_planes_kernel<2, 1024>), dim3(C / 2, B), dim3(1024), 2 * lds, st, p, gmax);
    else hipLaunchKernelGGL((epi_sample_bwd_planes_kernel<1, kPlaneThreads>), dim3(C, B), dim3(kPlaneThreads), lds, st, p, gmax);
    if (!g_coords) return launch_status();
    p.g_fmap1 = nullptr;
    for (int l = 0; l < L; ++l) p.g_f2[l] = nullptr;
  }
  hipLaunchKernelGGL(epi_sample_bwd_kernel, dim3((h * w + 255) / 256, L * ((d1 + kEpiG - 1) / kEpiG), B), dim3(256), 0, st, p);
  return launch_status();
}

// N3: ManyDepth's cost volume as MAL's student encoder builds it, forward only (upstream runs
// it under torch.no_grad()): manydepth/networks/resnet_encoder.py:152-233 `match_features` and the lines of
// `ResnetEncoderMatching.forward` that consume it (:296-312).
//
// Upstream loops over the batch in Python and, per sample and lookup frame, replicates the 64-channel feature
// map 96 times, back-projects / projects 96 x h x w points through ~10 ATen launches, grid_samples 189 MB and
// reduces it again.  Here: features channel-last (B,h,w,64), one wavefront per output pixel, two launches.
//
//   costvol_match_kernel   phase 1: lane = depth bin.  Each lane back-projects the pixel to ITS bin's depth,
//                          projects it into the lookup frame (layers.py:163-199), unnormalises (zeros padding,
//                          align_corners=True), and keeps the four tap offsets (clamped), the four weights
//                          (zeroed for taps outside the image) and the border mask of resnet_encoder.py:199-211.
//                          phase 2: lane = channel.  For every bin the tap data is broadcast from its lane, the
//                          four taps are four coalesced 256-byte loads, |warped - current| is summed over the
//                          wave, and the bin's lane accumulates mean * mask and the hit count over the frames.
//                          Result: cost / (counts + 1e-7) per (bin, pixel).
//   costvol_finish_kernel  per pixel over the bins: missing = (cost == 0), missing bins take the pixel's maximum
//                          (:221-226), confidence = every bin seen (:255-262, 299-300), lowest_cost = 1 / depth
//                          of the first minimum with zeros read as 100 (:303-307), volume x confidence (:311).
#include "mal_common.h"
#include "mal_device.h"

namespace mal {

constexpr int kCvC = 64;  // feature channels = lanes (ResNet-18 stage 1, num_ch_enc[1])

struct CostVolParams {
  const float* cur;    // (B,h,w,64)
  const float* look;   // (B,F,h,w,64)
  const float* K; const float* invK;  // (B,16) at the matching resolution
  const float* poses;  // (B,F,16)
  const float* bins;   // (D)
  int B, F, D, h, w; float eps; int set_missing_to_max;
  float* cost;         // (B,D,h,w) normalised cost, then (finish) the returned volume
  float* missing;      // (B,D,h,w) nullable
  float* masked;       // (B,D,h,w) nullable: volume x confidence
  float* lowest_cost;  // (B,h,w) nullable
  float* confidence;   // (B,h,w) nullable
};

struct TapSet { int o[4]; float w[4]; float edge; };

// the reference's chain for one (pixel, depth): layers.py:163-168 (X = depth * ray), :184-199 (project),
// grid_sample's unnormalise with zeros padding, resnet_encoder.py:199-205 (border mask on the sampling position)
MAL_DEV TapSet taps_for(const float* P, const float* ray, float depth, float eps, int w, int h) {
  TapSet t;
  const float X[3] = {depth * ray[0], depth * ray[1], depth * ray[2]};
  float c[3];
  for (int i = 0; i < 3; ++i) {
    float acc = P[4 * i] * X[0];
    acc = fma_(P[4 * i + 1], X[1], acc);
    acc = fma_(P[4 * i + 2], X[2], acc);
    c[i] = fma_(P[4 * i + 3], 1.0f, acc);
  }
  const float zp = c[2] + eps;
  const float u = div_safe_(c[0], zp), v = div_safe_(c[1], zp);
  const float wm1 = (float)(w - 1), hm1 = (float)(h - 1);
  const float gx = (div_(u, wm1) - 0.5f) * 2.0f, gy = (div_(v, hm1) - 0.5f) * 2.0f;
  const float ix = (gx + 1.0f) * (wm1 * 0.5f), iy = (gy + 1.0f) * (hm1 * 0.5f);
  const float xv = (gx / 2.0f + 0.5f) * wm1, yv = (gy / 2.0f + 0.5f) * hm1;
  t.edge = (xv >= 2.0f && xv <= (float)(w - 2) && yv >= 2.0f && yv <= (float)(h - 2)) ? 1.0f : 0.0f;
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float tx = ix - x0f, ex = 1.0f - tx, ty = iy - y0f, ey = 1.0f - ty;
  // keep the int conversion defined for wild projections: such taps are outside anyway
  const float xc = fminf(fmaxf(x0f, -2.0f), (float)w + 1.0f), yc = fminf(fmaxf(y0f, -2.0f), (float)h + 1.0f);
  const int x0 = (int)xc, y0 = (int)yc, x1 = x0 + 1, y1 = y0 + 1;
  const bool wild = !(x0f == xc && y0f == yc);  // also true for NaN
  const bool vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
  const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1), cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
  t.o[0] = cy0 * w + cx0; t.o[1] = cy0 * w + cx1; t.o[2] = cy1 * w + cx0; t.o[3] = cy1 * w + cx1;
  t.w[0] = (!wild && vx0 && vy0) ? ey * ex : 0.f;
  t.w[1] = (!wild && vx1 && vy0) ? ey * tx : 0.f;
  t.w[2] = (!wild && vx0 && vy1) ? ty * ex : 0.f;
  t.w[3] = (!wild && vx1 && vy1) ? ty * tx : 0.f;
  return t;
}

// 64-lane sum with DPP only (no LDS crossbar): pairs, quads, half rows, rows, then the two row broadcasts;
// the total lands in lane 63
template <int CTRL, int ROW_MASK>
MAL_DEV float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
MAL_DEV float wave_sum_dpp(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror: every lane holds its row's sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1, 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2, 3
  return v;
}

MAL_DEV float bcast(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
MAL_DEV int bcasti(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

__global__ __launch_bounds__(256) void costvol_match_kernel(CostVolParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = p.h, w = p.w, hw = h * w;
  const int xg = blockIdx.x * 4 + wv, y = blockIdx.y, b = blockIdx.z;
  if (xg >= w) return;
  const int pix = y * w + xg;
  const int rounds = (p.D + 63) / 64;
  float* out = p.cost + (size_t)b * p.D * hw + pix;
  const bool inner = y >= 2 && y < h - 2 && xg >= 2 && xg < w - 2;  // resnet_encoder.py:208-210
  if (!inner) {  // every difference is masked: cost 0 / (0 + 1e-7) = 0
    for (int k = 0; k < rounds; ++k)
      if (k * 64 + lane < p.D) out[(size_t)(k * 64 + lane) * hw] = 0.f;
    return;
  }
  const float curv = p.cur[((size_t)b * hw + pix) * kCvC + lane];
  float ray[3], ik[9];
  for (int e = 0; e < 9; ++e) ik[e] = p.invK[b * 16 + (e / 3) * 4 + (e % 3)];
  ray_of(ik, (float)xg, (float)y, ray);
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, cnt[4] = {0.f, 0.f, 0.f, 0.f};  // bins lane, lane+64, ... (D <= 256)
  for (int f = 0; f < p.F; ++f) {
    const float* T = p.poses + ((size_t)b * p.F + f) * 16;
    float tsum = 0.f;
    for (int i = 0; i < 16; ++i) tsum += T[i];
    if (tsum == 0.f) continue;  // a missing lookup frame (:176-178); wave-uniform
    float P[12];
    compose_P(p.K + b * 16, T, P);
    const float* lf = p.look + (((size_t)b * p.F + f) * hw) * kCvC + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k >= rounds) break;
      const int dmine = k * 64 + lane;
      const TapSet t = taps_for(P, ray, p.bins[min(dmine, p.D - 1)], p.eps, w, h);  // phase 1: lane = bin
      const int nb = min(64, p.D - k * 64);
      for (int j = 0; j < nb; ++j) {                                                   // phase 2: lane = channel
        if (bcast(t.edge, j) == 0.f) continue;  // masked out (wave-uniform): the difference would be multiplied by 0
        const int o0 = bcasti(t.o[0], j), o1 = bcasti(t.o[1], j), o2 = bcasti(t.o[2], j), o3 = bcasti(t.o[3], j);
        const float w0 = bcast(t.w[0], j), w1 = bcast(t.w[1], j), w2 = bcast(t.w[2], j), w3 = bcast(t.w[3], j);
        const float a = lf[(size_t)o0 * kCvC], bb = lf[(size_t)o1 * kCvC], c = lf[(size_t)o2 * kCvC], d = lf[(size_t)o3 * kCvC];
        float o = a * w0;
        o = fma_(bb, w1, o);
        o = fma_(c, w2, o);
        o = fma_(d, w3, o);
        const float s = wave_sum_dpp(fabsf(o - curv));        // lane 63 holds the sum
        const float diff = bcast(s, 63) * (1.0f / (float)kCvC) * bcast(t.edge, j);
        if (lane == j) { acc[k] += diff; cnt[k] += diff > 0.f ? 1.0f : 0.f; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < rounds && k * 64 + lane < p.D) out[(size_t)(k * 64 + lane) * hw] = div_(acc[k], cnt[k] + 1e-7f);
}

// Second formulation (default): features stay PLANAR (B,C,h,w) as the encoder produces them -- no relayout -- and
// lane = pixel: a wavefront owns 64 consecutive pixels and kCvG consecutive depth bins.  Per lookup frame every lane
// derives its taps for its bins once (phase 1 of the first formulation, without the broadcast), then walks the 64
// channel planes: per (bin, channel) four dword loads whose lanes fall in adjacent addresses (neighbouring pixels
// sample neighbouring positions) and that move only a little from bin to bin (the epipolar line), so they hit the
// L1 -- and 6 VALU operations, with no cross-lane reduction at all (the channel sum is a per-lane running sum).
// bins per wavefront, (4 offsets + 4 weights) registers each.  Measured at B=12 48x160 D=96: 2..6 bins 0.50 ms, 8 0.52,
// 12 0.88, 16 1.28 (spills); forcing three or four waves per SIMD through the launch bounds is slower (0.65-0.74 ms:
// the compiler's wide version keeps more loads in flight); loading a row's two taps as one unaligned 8-byte pair is
// slower too (0.58 ms).
constexpr int kCvG = 6;

__global__ __launch_bounds__(256) void costvol_match_px_kernel(CostVolParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = p.h, w = p.w, hw = h * w;
  const int pix0 = (blockIdx.x * 4 + wv) * 64, grp = blockIdx.y, b = blockIdx.z;
  if (pix0 >= hw) return;
  const bool live = pix0 + lane < hw;
  const int pix = min(pix0 + lane, hw - 1);
  const int y = pix / w, x = pix - y * w;
  const bool inner = y >= 2 && y < h - 2 && x >= 2 && x < w - 2;  // resnet_encoder.py:208-210
  const int d0 = grp * kCvG;
  float ray[3], ik[9];
  for (int e = 0; e < 9; ++e) ik[e] = p.invK[b * 16 + (e / 3) * 4 + (e % 3)];
  ray_of(ik, (float)x, (float)y, ray);
  float acc[kCvG], cnt[kCvG];
#pragma unroll
  for (int j = 0; j < kCvG; ++j) { acc[j] = 0.f; cnt[j] = 0.f; }
  const float* cf = p.cur + (size_t)b * kCvC * hw;
  for (int f = 0; f < p.F; ++f) {
    const float* T = p.poses + ((size_t)b * p.F + f) * 16;
    float tsum = 0.f;
    for (int i = 0; i < 16; ++i) tsum += T[i];
    if (tsum == 0.f) continue;  // a missing lookup frame (:176-178); wave-uniform
    float P[12];
    compose_P(p.K + b * 16, T, P);
    unsigned off[kCvG][4];
    float wt[kCvG][4], edge[kCvG], sum[kCvG];
    bool any = false;
#pragma unroll
    for (int j = 0; j < kCvG; ++j) {
      const TapSet t = taps_for(P, ray, p.bins[min(d0 + j, p.D - 1)], p.eps, w, h);
#pragma unroll
      for (int k = 0; k < 4; ++k) { off[j][k] = (unsigned)t.o[k] * 4u; wt[j][k] = t.w[k]; }
      edge[j] = inner ? t.edge : 0.f;
      sum[j] = 0.f;
      any = any || edge[j] != 0.f;
    }
    if (__ballot(any) == 0ull) continue;  // every (pixel, bin) of this wave is masked out: all differences are x 0
    const float* lf = p.look + ((size_t)b * p.F + f) * kCvC * hw;
    for (int ch = 0; ch < kCvC; ++ch) {
      const char* pl = reinterpret_cast<const char*>(lf + (size_t)ch * hw);
      const float cv = cf[(size_t)ch * hw + pix];
#pragma unroll
      for (int j = 0; j < kCvG; ++j) {
        const float a = *reinterpret_cast<const float*>(pl + off[j][0]), bb = *reinterpret_cast<const float*>(pl + off[j][1]);
        const float c = *reinterpret_cast<const float*>(pl + off[j][2]), d = *reinterpret_cast<const float*>(pl + off[j][3]);
        float o = a * wt[j][0];
        o = fma_(bb, wt[j][1], o);
        o = fma_(c, wt[j][2], o);
        o = fma_(d, wt[j][3], o);
        sum[j] += fabsf(o - cv);
      }
    }
#pragma unroll
    for (int j = 0; j < kCvG; ++j) {
      const float diff = sum[j] * (1.0f / (float)kCvC) * edge[j];
      acc[j] += diff;
      cnt[j] += diff > 0.f ? 1.0f : 0.f;
    }
  }
  float* out = p.cost + (size_t)b * p.D * hw + pix;
#pragma unroll
  for (int j = 0; j < kCvG; ++j)
    if (live && d0 + j < p.D) out[(size_t)(d0 + j) * hw] = div_(acc[j], cnt[j] + 1e-7f);  // 0 / 1e-7 = 0 outside the inner region
}

// 64 pixels per workgroup, the bins split over its four wavefronts (lane = pixel: every access is a coalesced row
// piece of one bin plane); the pixel's maximum / hit count / first minimum are combined through LDS in bin order
__global__ __launch_bounds__(256) void costvol_finish_kernel(CostVolParams p) {
  __shared__ float s_mx[4][64], s_best[4][64];
  __shared__ int s_seen[4][64], s_arg[4][64];
  const int hw = p.h * p.w, lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < p.B * hw;
  const int ii = live ? i : p.B * hw - 1;
  const int b = ii / hw, pix = ii - b * hw;
  const int per = (p.D + 3) / 4, d_lo = min(part * per, p.D), d_hi = min(d_lo + per, p.D);
  float* c = p.cost + (size_t)b * p.D * hw + pix;
  float mx = -INFINITY;
  int seen = 0;
#pragma unroll 8
  for (int d = d_lo; d < d_hi; ++d) {
    const float v = c[(size_t)d * hw];
    mx = fmaxf(mx, v);
    seen += v > 0.f ? 1 : 0;  // (cost * (1 - missing) > 0): a missing bin is exactly 0
  }
  s_mx[part][lane] = mx; s_seen[part][lane] = seen;
  __syncthreads();
  mx = fmaxf(fmaxf(s_mx[0][lane], s_mx[1][lane]), fmaxf(s_mx[2][lane], s_mx[3][lane]));
  seen = (s_seen[0][lane] + s_seen[1][lane]) + (s_seen[2][lane] + s_seen[3][lane]);
  const float conf = seen == p.D ? 1.0f : 0.0f;
  float best = INFINITY;
  int arg = 0;
#pragma unroll 4
  for (int d = d_lo; d < d_hi; ++d) {
    const float v = c[(size_t)d * hw];
    const float miss = v == 0.f ? 1.0f : 0.0f;
    const float filled = p.set_missing_to_max ? v * (1.0f - miss) + mx * miss : v;
    const float viz = filled == 0.f ? 100.0f : filled;
    if (viz < best) { best = viz; arg = d; }  // first minimum, as torch.min
    if (live) {
      const size_t o = (size_t)b * p.D * hw + (size_t)d * hw + pix;
      c[(size_t)d * hw] = filled;
      if (p.missing) p.missing[o] = miss;
      if (p.masked) p.masked[o] = filled * conf;
    }
  }
  s_best[part][lane] = best; s_arg[part][lane] = arg;
  __syncthreads();
  if (part == 0 && live) {
    for (int k = 1; k < 4; ++k)
      if (s_best[k][lane] < best) { best = s_best[k][lane]; arg = s_arg[k][lane]; }  // strict: the earlier bin wins a tie
    if (p.confidence) p.confidence[i] = conf;
    if (p.lowest_cost) p.lowest_cost[i] = div_(1.0f, p.bins[arg]);
  }
}

}  // namespace mal

using namespace mal;

namespace mal { opt_t g_costvol_impl{1}; }  // 1 = planar features, lane = pixel (default); 0 = channel-last, lane = channel

extern "C" int mal_costvol_channel_last(void) { return g_costvol_impl == 0; }

extern "C" int mal_cost_volume(const float* current_feats, const float* lookup_feats, const float* poses, const float* K,
                               const float* inv_K, const float* depth_bins, int B, int F, int C, int D, int h, int w,
                               float eps, int set_missing_to_max, float* cost_volume, float* missing_mask,
                               float* masked_cost_volume, float* lowest_cost, float* confidence_mask, void* stream) {
  if (B <= 0 || F <= 0 || D <= 0 || h < 5 || w < 5) return MAL_ESHAPE;
  if (C != kCvC || D > 256) return MAL_ESHAPE;  // lanes = channels; four bins per lane at most
  if ((double)B * D * h * w > 2.0e9 / 4 || (double)B * F * h * w * C > 2.0e9 / 4) return MAL_ESHAPE;
  if (!current_feats || !lookup_feats || !poses || !K || !inv_K || !depth_bins || !cost_volume) return MAL_EINVAL;
  CostVolParams p = {};
  p.cur = current_feats; p.look = lookup_feats; p.poses = poses; p.K = K; p.invK = inv_K; p.bins = depth_bins;
  p.B = B; p.F = F; p.D = D; p.h = h; p.w = w; p.eps = eps; p.set_missing_to_max = set_missing_to_max;
  p.cost = cost_volume; p.missing = missing_mask; p.masked = masked_cost_volume; p.lowest_cost = lowest_cost;
  p.confidence = confidence_mask;
  hipStream_t st = (hipStream_t)stream;
  if (g_costvol_impl == 0)
    hipLaunchKernelGGL(costvol_match_kernel, dim3((w + 3) / 4, h, B), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(costvol_match_px_kernel, dim3((h * w + 255) / 256, (D + kCvG - 1) / kCvG, B), dim3(256), 0, st, p);
  hipLaunchKernelGGL(costvol_finish_kernel, dim3((B * h * w + 63) / 64), dim3(256), 0, st, p);
  return launch_status();
}

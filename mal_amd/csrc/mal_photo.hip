// Photometric operators on MATERIALISED images (the path taken when the warped frames must
// exist as tensors: the module-level drop-ins, and the --temporal hint whose "syn" candidates
// are built from the warped images, manydepth/dyn_utils.py:127-168).
//
//   mal_ssim / _bwd                 SSIM.forward                      manydepth/layers.py:243-257
//   mal_reprojection_loss / _bwd    compute_reprojection_loss         manydepth/loss_utils.py:46-55
//   mal_photo_fwd / _bwd            min over candidates + automask    manydepth/loss_utils.py:79-113,146-199
//   mal_smooth_loss                 get_smooth_loss (+ mean norm)     manydepth/layers.py:210-223, loss_utils.py:119-121
//   mal_distil_epilogue             consistency + distillation        manydepth/loss_utils.py:193-254
//   mal_axpy_maps / mal_finish_scalars / mal_sum_f64   gradient assembly without host syncs
//
// These are one-thread-per-pixel kernels that read their 3x3 windows straight from global
// memory (the windows of neighbouring lanes overlap, so L1/L2 absorb the re-reads); the
// LDS-tiled, single-launch formulation of the same arithmetic is mal_pass.hip.
#include "mal_common.h"
#include "mal_device.h"

namespace mal {

// window sums of two planes around (gy,gx) with reflection padding
MAL_DEV void window_sums(const float* __restrict__ x, const float* __restrict__ y, int gy, int gx, int H, int W,
                         float* sx, float* sy, float* sxx, float* syy, float* sxy) {
  int ro[3], co[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    ro[d] = reflect1(gy + d - 1, H) * W;
    co[d] = reflect1(gx + d - 1, W);
  }
  float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float xv = x[ro[i] + co[j]], yv = y[ro[i] + co[j]];
      a += xv; b += yv; aa += xv * xv; bb += yv * yv; ab += xv * yv;
    }
  *sx = a; *sy = b; *sxx = aa; *syy = bb; *sxy = ab;
}

MAL_DEV float ssim_at(const float* x, const float* y, int gy, int gx, int H, int W, SsimStats* st) {
  float sx, sy, sxx, syy, sxy;
  window_sums(x, y, gy, gx, H, W, &sx, &sy, &sxx, &syy, &sxy);
  return ssim_from_sums(sx, sy, sxx, syy, sxy, st);
}

MAL_DEV float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// multiplicity of neighbour offset d (-1,0,1) of pixel q along an axis of length n: the
// reflected border tap of the neighbour lands on q once more (adjoint of ReflectionPad2d(1)).
MAL_DEV float refl_mult(int q, int d, int n) {
  const int pq = q + d;
  if (pq < 0 || pq >= n) return 0.f;
  if (d == -1 && q == 1) return 2.f;
  if (d == 1 && q == n - 2) return 2.f;
  return 1.f;
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* x, const float* y, int planes, int H, int W,
                                                       float* out) {
  const int HW = H * W;
  const size_t n = (size_t)planes * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int pl = (int)(i / HW), pix = (int)(i - (size_t)pl * HW);
    const int gy = pix / W, gx = pix - gy * W;
    out[i] = clamp01(ssim_at(x + (size_t)pl * HW, y + (size_t)pl * HW, gy, gx, H, W, nullptr));
  }
}

// gradient of sum(g_out * SSIM(x,y)) wrt x and y: gather over the 3x3 neighbours of q
MAL_DEV void ssim_gather(const float* xp, const float* yp, const float* gp /*nullable: plane of upstream*/,
                         float gconst, int gy, int gx, int H, int W, float* gx_out, float* gy_out) {
  const float xq = xp[gy * W + gx], yq = yp[gy * W + gx];
  float ax = 0.f, ay = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const float wgt = refl_mult(gy, dy, H) * refl_mult(gx, dx, W);
      if (wgt == 0.f) continue;
      const int py = gy + dy, px = gx + dx;
      const float up = gp ? gp[py * W + px] * gconst : gconst;
      if (up == 0.f) continue;
      SsimStats st;
      ssim_at(xp, yp, py, px, H, W, &st);
      if (st.gate == 0.f) continue;
      const SsimGrad g = ssim_partials(st);
      const float kk = -0.5f * up * wgt * (1.0f / 9.0f);
      ax += kk * (g.dmx + 2.0f * xq * g.dsxx + yq * g.dsxy);
      ay += kk * (g.dmy + 2.0f * yq * g.dsyy + xq * g.dsxy);
    }
  *gx_out = ax;
  *gy_out = ay;
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* x, const float* y, const float* g_out, int planes,
                                                       int H, int W, float* g_x, float* g_y) {
  const int HW = H * W;
  const size_t n = (size_t)planes * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int pl = (int)(i / HW), pix = (int)(i - (size_t)pl * HW);
    const int gy = pix / W, gx = pix - gy * W;
    float a, b;
    ssim_gather(x + (size_t)pl * HW, y + (size_t)pl * HW, g_out + (size_t)pl * HW, 1.0f, gy, gx, H, W, &a, &b);
    if (g_x) g_x[i] = a;
    if (g_y) g_y[i] = b;
  }
}

// r(pred, target) at one pixel: 0.85 * mean_c SSIM + 0.15 * mean_c |t - p|
MAL_DEV float reproj_at(const float* pred_b, const float* tgt_b, int C, int gy, int gx, int H, int W, bool no_ssim) {
  const int HW = H * W, pix = gy * W + gx;
  float ssum = 0.f, lsum = 0.f;
  for (int ch = 0; ch < C; ++ch) {
    const float l1 = fabsf(tgt_b[ch * HW + pix] - pred_b[ch * HW + pix]);
    lsum = ch == 0 ? l1 : lsum + l1;
    if (!no_ssim) {
      const float v = clamp01(ssim_at(pred_b + (size_t)ch * HW, tgt_b + (size_t)ch * HW, gy, gx, H, W, nullptr));
      ssum = ch == 0 ? v : ssum + v;
    }
  }
  const float cf = (float)C;
  const float l1m = C == 3 ? div3_(lsum) : div_(lsum, cf);
  if (no_ssim) return l1m;
  const float sm = C == 3 ? div3_(ssum) : div_(ssum, cf);
  return 0.85f * sm + 0.15f * l1m;
}

__global__ __launch_bounds__(256) void reproj_fwd_kernel(const float* pred, const float* target, int B, int C, int H,
                                                         int W, int no_ssim, float* out) {
  const int HW = H * W;
  const size_t n = (size_t)B * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW), pix = (int)(i - (size_t)b * HW);
    const int gy = pix / W, gx = pix - gy * W;
    out[i] = reproj_at(pred + (size_t)b * C * HW, target + (size_t)b * C * HW, C, gy, gx, H, W, no_ssim != 0);
  }
}

// gradient of sum_px up(px) * r(pred,target)(px) wrt pred / target at (b, ch, q)
__global__ __launch_bounds__(256) void reproj_bwd_kernel(const float* pred, const float* target, const float* g_out,
                                                         int B, int C, int H, int W, int no_ssim, float* g_pred,
                                                         float* g_target) {
  const int HW = H * W;
  const size_t n = (size_t)B * C * HW;
  const float invC = 1.0f / (float)C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int pl = (int)(i / HW), pix = (int)(i - (size_t)pl * HW);
    const int b = pl / C;
    const int gy = pix / W, gx = pix - gy * W;
    const float* up = g_out + (size_t)b * HW;
    const float* xp = pred + (size_t)pl * HW;
    const float* yp = target + (size_t)pl * HW;
    float a = 0.f, bq = 0.f;
    const float l1w = no_ssim ? invC : 0.15f * invC;
    if (!no_ssim) ssim_gather(xp, yp, up, 0.85f * invC, gy, gx, H, W, &a, &bq);
    const float s = sgn(xp[pix] - yp[pix]) * l1w * up[pix];
    if (g_pred) g_pred[i] = a + s;
    if (g_target) g_target[i] = bq - s;
  }
}

// ---------------------------------------------------------------- photo fwd / bwd
// marching formulation (mal_photo_march.hip), the default for SSIM + min
extern opt_t g_photo_impl;
int photo_march_fwd(const float* target, const float* const* cand, int n_cand, const float* ident, const float* noise,
                    const float* ext_mask, int B, int H, int W, int automask, float* min_reproj, uint8_t* argmin,
                    float* weight_out, double* block_sums, int* ntasks_out, hipStream_t st);
int photo_march_bwd(const float* target, const float* const* cand, int n_cand, const uint8_t* argmin, const float* weight,
                    const float* scale, const double* sums, int B, int H, int W, float* const* g_cand, hipStream_t st);

struct PhotoParams {
  const float* target; const float* cand[MAL_MAX_CAND]; int n_cand;
  const float* ident; const float* noise; const float* ext_mask;
  int B, H, W, flags;
  float* min_reproj; uint8_t* argmin_u8; float* weight_out; double* block_sums;
  // bwd
  const uint8_t* argmin_in; const float* weight_in; const float* scale; const double* sums;
  float* g_cand[MAL_MAX_CAND];
};

__global__ __launch_bounds__(256) void photo_fwd_kernel(PhotoParams p) {
  __shared__ double s_red[4][2];
  const int HW = p.H * p.W;
  const size_t n = (size_t)p.B * HW;
  const bool no_ssim = p.flags & MAL_F_NO_SSIM, avg = p.flags & MAL_F_AVG, automask = p.flags & MAL_F_AUTOMASK;
  double acc_rw = 0.0, acc_w = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW), pix = (int)(i - (size_t)b * HW);
    const int gy = pix / p.W, gx = pix - gy * p.W;
    const float* tb = p.target + (size_t)b * 3 * HW;
    float rp = 0.f;
    int win = 0;
    for (int c = 0; c < p.n_cand; ++c) {
      const float r = reproj_at(p.cand[c] + (size_t)b * 3 * HW, tb, 3, gy, gx, p.H, p.W, no_ssim);
      if (avg) rp = c == 0 ? r : rp + r;
      else if (c == 0 || r < rp) { rp = r; win = c; }
    }
    if (avg) { rp = div_(rp, (float)p.n_cand); win = 255; }
    float w = 1.0f;
    if (automask) {
      float idn = p.ident[i];
      if (p.noise) idn += p.noise[i] * 0.00001f;
      w = (rp <= idn) ? 1.0f : 0.0f;
    }
    if (p.ext_mask) w *= p.ext_mask[i];
    if (p.min_reproj) p.min_reproj[i] = rp;
    if (p.argmin_u8) p.argmin_u8[i] = (uint8_t)win;
    if (p.weight_out) p.weight_out[i] = w;
    acc_rw += (double)(rp * w);
    acc_w += (double)w;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double r0 = wave_sum_d(acc_rw), r1 = wave_sum_d(acc_w);
  if (lane == 0) { s_red[wv][0] = r0; s_red[wv][1] = r1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int j = threadIdx.x;
    p.block_sums[(size_t)blockIdx.x * 2 + j] = (s_red[0][j] + s_red[1][j]) + (s_red[2][j] + s_red[3][j]);
  }
}

__global__ __launch_bounds__(256) void photo_bwd_kernel(PhotoParams p) {
  const int HW = p.H * p.W;
  const size_t n = (size_t)p.B * 3 * HW;
  const bool no_ssim = p.flags & MAL_F_NO_SSIM, avg = p.flags & MAL_F_AVG;
  float sc = p.scale ? *p.scale : 1.0f;
  if (p.sums) sc = (float)((double)sc / (p.sums[1] + 1e-7));
  if (avg) sc = sc / (float)p.n_cand;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int pl = (int)(i / HW), pix = (int)(i - (size_t)pl * HW);
    const int b = pl / 3;
    const int gy = pix / p.W, gx = pix - gy * p.W;
    const float* yp = p.target + (size_t)pl * HW;
    const float* wgt = p.weight_in + (size_t)b * HW;
    const uint8_t* am = p.argmin_in + (size_t)b * HW;
    const float yq = yp[pix];
    for (int c = 0; c < p.n_cand; ++c) {
      if (!p.g_cand[c]) continue;
      const float* xp = p.cand[c] + (size_t)pl * HW;
      const float xq = xp[pix];
      float a = 0.f;
      if (!no_ssim) {
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const float m = refl_mult(gy, dy, p.H) * refl_mult(gx, dx, p.W);
            if (m == 0.f) continue;
            const int q = (gy + dy) * p.W + gx + dx;
            if (!avg && am[q] != c) continue;
            const float up = wgt[q];
            if (up == 0.f) continue;
            SsimStats st;
            ssim_at(xp, yp, gy + dy, gx + dx, p.H, p.W, &st);
            if (st.gate == 0.f) continue;
            const SsimGrad g = ssim_partials(st);
            const float kk = -0.5f * (0.85f / 3.0f) * up * m * (1.0f / 9.0f);
            a += kk * (g.dmx + 2.0f * xq * g.dsxx + yq * g.dsxy);
          }
      }
      if (avg || am[pix] == c) a += sgn(xq - yq) * (no_ssim ? (1.0f / 3.0f) : (0.15f / 3.0f)) * wgt[pix];
      p.g_cand[c][i] = a * sc;
    }
  }
}

// out[j] = sum_i partial[i*stride + j], j < nvals (one workgroup)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* partial, int nparts, int stride, int nvals,
                                                              double* out) {
  __shared__ double s[256];
  const int tid = threadIdx.x;
  for (int j = 0; j < nvals; ++j) {
    double acc = 0.0;
    for (int i = tid; i < nparts; i += 256) acc += partial[(size_t)i * stride + j];
    s[tid] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
      if (tid < k) s[tid] += s[tid + k];
      __syncthreads();
    }
    if (tid == 0) out[j] = s[0];
    __syncthreads();
  }
}

// ---------------------------------------------------------------- smoothness
// stage 1: per-sample sum of disp (partials[b][chunk])
__global__ __launch_bounds__(256) void plane_sum_kernel(const float* x, int HW, int chunks, double* partial) {
  __shared__ double s[4];
  const int b = blockIdx.x / chunks, ck = blockIdx.x % chunks;
  double acc = 0.0;
  for (int i = ck * 256 + threadIdx.x; i < HW; i += chunks * 256) acc += (double)x[(size_t)b * HW + i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

int smooth_march(const float* disp, const float* img, int B, int H, int W, int normalise, float* gn, double* partial,
                 double* stats, double* sums, double* loss_out, hipStream_t st);

struct SmoothParams {
  const float* disp; const float* img; int B, C, H, W; int normalise; int chunks;
  const double* mean_partial;  // [B][chunks] sums of disp
  double* partial;             // [B*chunks][4]: sum_x, sum_y, dot(gn, disp)
  double* stats;               // [2B]: mean_b, then corr_b = dot_b / (HW (mean_b+eps)^2)
  float* gn;                   // d loss / d normalised disp (nullable)
};

MAL_DEV float sample_mean(const SmoothParams& p, int b, int HW) { return (float)p.stats[b]; }

// one wave per sample.  stage 1: mean_b from the plane partials.  stage 2: corr_b, and (block B)
// the loss = sum_x/Nx + sum_y/Ny from all smooth partials.
__global__ __launch_bounds__(64) void smooth_mid_kernel(SmoothParams p, int stage, double* loss_out) {
  const int b = blockIdx.x, lane = threadIdx.x, HW = p.H * p.W;
  if (stage == 1) {
    double acc = 0.0;
    for (int k = lane; k < p.chunks; k += 64) acc += p.mean_partial[b * p.chunks + k];
    acc = wave_sum_d(acc);
    if (lane == 0) p.stats[b] = acc / (double)HW;
    return;
  }
  if (b < p.B) {
    double dot = 0.0;
    for (int k = lane; k < p.chunks; k += 64) dot += p.partial[(size_t)(b * p.chunks + k) * 4 + 2];
    dot = wave_sum_d(dot);
    if (lane == 0) {
      const double m = (double)((float)p.stats[b] + 1e-7f);
      p.stats[p.B + b] = dot / ((double)HW * m * m);
    }
  } else {
    double sx = 0.0, sy = 0.0;
    for (int i = lane; i < p.B * p.chunks; i += 64) { sx += p.partial[(size_t)i * 4]; sy += p.partial[(size_t)i * 4 + 1]; }
    sx = wave_sum_d(sx); sy = wave_sum_d(sy);
    if (lane == 0) loss_out[0] = sx / ((double)p.B * p.H * (p.W - 1)) + sy / ((double)p.B * (p.H - 1) * p.W);
  }
}

__global__ __launch_bounds__(256) void smooth_kernel(SmoothParams p) {
  __shared__ double s[4][3];
  const int HW = p.H * p.W;
  const int b = blockIdx.x / p.chunks, ck = blockIdx.x % p.chunks;
  const float inv = p.normalise ? div_(1.0f, sample_mean(p, b, HW) + 1e-7f) : 1.0f;
  const float* d = p.disp + (size_t)b * HW;
  const float* im = p.img + (size_t)b * p.C * HW;
  const float nx = 1.0f / ((float)p.B * (float)p.H * (float)(p.W - 1));
  const float ny = 1.0f / ((float)p.B * (float)(p.H - 1) * (float)p.W);
  const float invC = 1.0f / (float)p.C;
  double ax = 0.0, ay = 0.0, adot = 0.0;
  for (int pix = ck * 256 + threadIdx.x; pix < HW; pix += p.chunks * 256) {
    const int gy = pix / p.W, gx = pix - gy * p.W;
    const float dq = d[pix];
    const float nq = p.normalise ? dq * inv : dq;
    // edge weights between q and its right / lower / left / upper neighbour
    auto edge = [&](int other) {
      float e = 0.f;
      for (int ch = 0; ch < p.C; ++ch) {
        const float v = fabsf(im[ch * HW + pix] - im[ch * HW + other]);
        e = ch == 0 ? v : e + v;
      }
      return expf(-(e * invC));
    };
    float g = 0.f;
    if (gx + 1 < p.W) {
      const float nn = p.normalise ? d[pix + 1] * inv : d[pix + 1];
      const float w = edge(pix + 1), df = nq - nn;
      ax += (double)(fabsf(df) * w);
      g += sgn(df) * w * nx;
    }
    if (gx > 0) {
      const float nn = p.normalise ? d[pix - 1] * inv : d[pix - 1];
      g -= sgn(nn - nq) * edge(pix - 1) * nx;
    }
    if (gy + 1 < p.H) {
      const float nn = p.normalise ? d[pix + p.W] * inv : d[pix + p.W];
      const float w = edge(pix + p.W), df = nq - nn;
      ay += (double)(fabsf(df) * w);
      g += sgn(df) * w * ny;
    }
    if (gy > 0) {
      const float nn = p.normalise ? d[pix - p.W] * inv : d[pix - p.W];
      g -= sgn(nn - nq) * edge(pix - p.W) * ny;
    }
    if (p.gn) p.gn[(size_t)b * HW + pix] = g;
    adot += (double)g * (double)dq;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  ax = wave_sum_d(ax); ay = wave_sum_d(ay); adot = wave_sum_d(adot);
  if (lane == 0) { s[wv][0] = ax; s[wv][1] = ay; s[wv][2] = adot; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int j = threadIdx.x;
    p.partial[(size_t)blockIdx.x * 4 + j] = (s[0][j] + s[1][j]) + (s[2][j] + s[3][j]);
  }
}

// g_disp = gn/(mean+eps) - corr_b
__global__ __launch_bounds__(256) void smooth_finish_kernel(SmoothParams p, float* g_disp) {
  const int HW = p.H * p.W;
  const int b = blockIdx.x / p.chunks, ck = blockIdx.x % p.chunks;
  float inv = 1.0f, corr = 0.0f;
  if (p.normalise) {
    inv = div_(1.0f, sample_mean(p, b, HW) + 1e-7f);
    corr = (float)p.stats[p.B + b];
  }
  for (int pix = ck * 256 + threadIdx.x; pix < HW; pix += p.chunks * 256) {
    const size_t gi = (size_t)b * HW + pix;
    g_disp[gi] = g_disp[gi] * inv - corr;
  }
}

// ---------------------------------------------------------------- distillation epilogue
struct DistilParams {
  const float* multi_depth; const float* mono_depth; const float* multi_reproj; const float* mono_reproj;
  const float* ens_reproj; const float* ext_mask; size_t n; int flags;
  double* block_sums; float* g_multi_cons; float* g_multi_distil; float* g_mono_distil; float* cons_target;
  // --learn_ens (loss_utils.py:240-241): the ensemble depth is a map of its own (disp_to_depth of the learnt head's
  // output) instead of (mono + multi) / 2, and receives the gradient of the pixels it wins
  const float* ens_depth; float* g_ens_depth;
};

__global__ __launch_bounds__(256) void distil_kernel(DistilParams p) {
  __shared__ double s_red[4][2];
  double acc_c = 0.0, acc_d = 0.0;
  const bool dual = p.flags & MAL_F_DUAL_DISTIL;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < p.n; i += (size_t)gridDim.x * blockDim.x) {
    const float dm = p.multi_depth[i], dmono = p.mono_depth[i];
    const float m = p.ext_mask ? p.ext_mask[i] : 1.0f, cm = 1.0f - m, mm = 1.0f - cm;
    const float dc = dm - dmono;
    acc_c += (double)(fabsf(dc) * cm);
    int idx = 0;
    float best = p.mono_reproj[i];
    if (p.ens_reproj) {
      const float r_ens = p.ens_reproj[i];
      if (r_ens < best) { best = r_ens; idx = 1; }
    }
    if (p.multi_reproj[i] < best) idx = 2;
    const float ens = p.ens_depth ? p.ens_depth[i] : (dmono + dm) / 2.0f;
    const float target = idx == 0 ? dmono : (idx == 2 ? dm : ens);
    const float dd = target - dm;
    acc_d += (double)(fabsf(dd) * mm);
    if (p.cons_target) p.cons_target[i] = div_(1.0f, dmono * cm + dm * (1.0f - cm));
    if (p.g_multi_cons) p.g_multi_cons[i] = sgn(dc) * cm;
    const float d_ens = p.ens_depth ? -1.0f : -0.5f;  // d (ens - dm) / d dm
    if (p.g_multi_distil) p.g_multi_distil[i] = sgn(dd) * (idx == 0 ? -1.0f : (idx == 2 ? 0.0f : d_ens)) * mm;
    if (p.g_ens_depth) p.g_ens_depth[i] = idx == 1 ? sgn(dd) * mm : 0.0f;
    if (p.g_mono_distil) p.g_mono_distil[i] = (dual && idx == 0) ? sgn(dd) * mm : 0.0f;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double r0 = wave_sum_d(acc_c), r1 = wave_sum_d(acc_d);
  if (lane == 0) { s_red[wv][0] = r0; s_red[wv][1] = r1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int j = threadIdx.x;
    p.block_sums[(size_t)blockIdx.x * 2 + j] = (s_red[0][j] + s_red[1][j]) + (s_red[2][j] + s_red[3][j]);
  }
}

// ---------------------------------------------------------------- assembly helpers
struct AxpyParams {
  int n_terms; const float* maps[6]; const float* scale[6]; const double* denom[6]; float mult[6]; float eps[6];
  size_t n; float* out; int accumulate;
};

__global__ __launch_bounds__(256) void axpy_kernel(AxpyParams p) {
  float coef[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    coef[k] = 0.f;
    if (k < p.n_terms) {
      double c = (double)p.mult[k];
      if (p.scale[k]) c *= (double)*p.scale[k];
      if (p.denom[k]) c /= (*p.denom[k] + (double)p.eps[k]);
      coef[k] = (float)c;
    }
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < p.n; i += (size_t)gridDim.x * blockDim.x) {
    float acc = p.accumulate ? p.out[i] : 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (k < p.n_terms) acc = fma_(coef[k], p.maps[k][i], acc);
    p.out[i] = acc;
  }
}

__global__ void finish_scalars_kernel(const double* num, const double* den, float eps, float mult, int n, float* out) {
  const int i = threadIdx.x;
  if (i < n) {
    double v = num[i] * (double)mult;
    if (den) v /= (den[i] + (double)eps);
    out[i] = (float)v;
  }
}

__global__ __launch_bounds__(256) void sum_partial_kernel(const float* x, size_t n, double* partial) {
  __shared__ double s[4];
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += (double)x[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

inline int ew_grid2(size_t n, int cap) {
  size_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > (size_t)cap ? (size_t)cap : g));
}

}  // namespace mal

using namespace mal;

extern "C" int mal_ssim(const float* x, const float* y, int B, int C, int H, int W, float* out, void* stream) {
  int rc = check_shape(B * C, H, W);
  if (rc) return rc;
  if (!x || !y || !out) return MAL_EINVAL;
  hipLaunchKernelGGL(ssim_fwd_kernel, dim3(ew_grid2((size_t)B * C * H * W, 4096)), dim3(256), 0, (hipStream_t)stream, x,
                     y, B * C, H, W, out);
  return launch_status();
}

extern "C" int mal_ssim_bwd(const float* x, const float* y, const float* g_out, int B, int C, int H, int W, float* g_x,
                            float* g_y, void* stream) {
  int rc = check_shape(B * C, H, W);
  if (rc) return rc;
  if (!x || !y || !g_out) return MAL_EINVAL;
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3(ew_grid2((size_t)B * C * H * W, 4096)), dim3(256), 0, (hipStream_t)stream, x,
                     y, g_out, B * C, H, W, g_x, g_y);
  return launch_status();
}

extern "C" int mal_reprojection_loss(const float* pred, const float* target, int B, int C, int H, int W, int no_ssim,
                                     float* out, void* stream) {
  int rc = check_shape(B * C, H, W);
  if (rc) return rc;
  if (!pred || !target || !out || C <= 0) return MAL_EINVAL;
  hipLaunchKernelGGL(reproj_fwd_kernel, dim3(ew_grid2((size_t)B * H * W, 4096)), dim3(256), 0, (hipStream_t)stream,
                     pred, target, B, C, H, W, no_ssim, out);
  return launch_status();
}

extern "C" int mal_reprojection_loss_bwd(const float* pred, const float* target, const float* g_out, int B, int C,
                                         int H, int W, int no_ssim, float* g_pred, float* g_target, void* stream) {
  int rc = check_shape(B * C, H, W);
  if (rc) return rc;
  if (!pred || !target || !g_out || C <= 0) return MAL_EINVAL;
  hipLaunchKernelGGL(reproj_bwd_kernel, dim3(ew_grid2((size_t)B * C * H * W, 4096)), dim3(256), 0, (hipStream_t)stream,
                     pred, target, g_out, B, C, H, W, no_ssim, g_pred, g_target);
  return launch_status();
}

extern "C" int mal_photo_fwd(const float* target, const float* const* cand, int n_cand, const float* ident,
                             const float* noise, const float* ext_mask, int B, int H, int W, int flags,
                             float* min_reproj, uint8_t* argmin_u8, float* weight_out, double* sums, void* ws,
                             size_t ws_bytes, void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (!target || !cand || n_cand < 1 || n_cand > MAL_MAX_CAND || !sums || !ws) return MAL_EINVAL;
  if ((flags & MAL_F_AUTOMASK) && !ident) return MAL_EINVAL;
  Workspace w = carve(ws, B, H, W);
  if (ws_bytes < w.bytes) return MAL_EWORKSPACE;
  for (int c = 0; c < n_cand; ++c) if (!cand[c]) return MAL_EINVAL;
  if (g_photo_impl == 1 && !(flags & (MAL_F_NO_SSIM | MAL_F_AVG)) && (n_cand <= 2 || (min_reproj && argmin_u8))) {
    int ntasks = 0;  // marching kernels (mal_photo_march.hip), two candidates per launch
    rc = photo_march_fwd(target, cand, n_cand, ident, noise, ext_mask, B, H, W, (flags & MAL_F_AUTOMASK) ? 1 : 0,
                         min_reproj, argmin_u8, weight_out, w.block_sums, &ntasks, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w.block_sums, ntasks, 2, 2, sums);
    return launch_status();
  }
  PhotoParams p = {};
  p.target = target; p.n_cand = n_cand;
  for (int c = 0; c < n_cand; ++c) p.cand[c] = cand[c];
  p.ident = ident; p.noise = noise; p.ext_mask = ext_mask; p.B = B; p.H = H; p.W = W; p.flags = flags;
  p.min_reproj = min_reproj; p.argmin_u8 = argmin_u8; p.weight_out = weight_out; p.block_sums = w.scratch;
  const int grid = ew_grid2((size_t)B * H * W, 2048);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(photo_fwd_kernel, dim3(grid), dim3(256), 0, st, p);
  rc = launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, w.scratch, grid, 2, 2, sums);
  return launch_status();
}

extern "C" int mal_photo_bwd(const float* target, const float* const* cand, int n_cand, const uint8_t* argmin_u8,
                             const float* weight, const float* scale, const double* sums, int B, int H, int W,
                             int flags, float* const* g_cand, void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (!target || !cand || n_cand < 1 || n_cand > MAL_MAX_CAND || !argmin_u8 || !weight || !g_cand) return MAL_EINVAL;
  for (int c = 0; c < n_cand; ++c) if (!cand[c]) return MAL_EINVAL;
  if (g_photo_impl == 1 && !(flags & (MAL_F_NO_SSIM | MAL_F_AVG)))
    return photo_march_bwd(target, cand, n_cand, argmin_u8, weight, scale, sums, B, H, W, g_cand, (hipStream_t)stream);
  PhotoParams p = {};
  p.target = target; p.n_cand = n_cand;
  for (int c = 0; c < n_cand; ++c) { if (!cand[c]) return MAL_EINVAL; p.cand[c] = cand[c]; p.g_cand[c] = g_cand[c]; }
  p.B = B; p.H = H; p.W = W; p.flags = flags; p.argmin_in = argmin_u8; p.weight_in = weight; p.scale = scale;
  p.sums = sums;
  hipLaunchKernelGGL(photo_bwd_kernel, dim3(ew_grid2((size_t)B * 3 * H * W, 4096)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

extern "C" int mal_smooth_loss(const float* disp, const float* img, int B, int C, int H, int W, int normalise,
                               double* loss_out, float* g_disp, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (!disp || !img || !loss_out || !ws || C <= 0) return MAL_EINVAL;
  Workspace w = carve(ws, B, H, W);
  if (ws_bytes < w.bytes) return MAL_EWORKSPACE;
  if (g_photo_impl == 1 && C == 3 && 4 * B <= 4096) {
    // marching sweep (mal_photo_march.hip): 2-3 launches instead of 5.  scratch: stats [2B], sums [2B]
    double* stats = w.scratch;
    double* sums = w.scratch + 2 * B;
    hipStream_t st = (hipStream_t)stream;
    rc = smooth_march(disp, img, B, H, W, normalise, g_disp, w.block_sums, stats, sums, loss_out, st);
    if (rc || !g_disp || !normalise) return rc;
    SmoothParams q = {disp, img, B, C, H, W, normalise, 64, nullptr, nullptr, stats, g_disp};
    hipLaunchKernelGGL(smooth_finish_kernel, dim3(B * 64), dim3(256), 0, st, q, g_disp);  // g = gn/(mean+eps) - corr
    return launch_status();
  }
  int chunks = (H * W + 1023) / 1024;
  const int cap = (4096 - 2 * B) / 5 / B;  // scratch holds 4096 doubles: B*chunks*(1+4) + 2B
  if (chunks > cap) chunks = cap;
  if (chunks < 1) return MAL_ESHAPE;
  SmoothParams p = {disp, img, B, C, H, W, normalise, chunks, w.scratch, w.scratch + (size_t)B * chunks,
                    w.scratch + (size_t)B * chunks * 5, g_disp};
  hipStream_t st = (hipStream_t)stream;
  if (normalise) {
    hipLaunchKernelGGL(plane_sum_kernel, dim3(B * chunks), dim3(256), 0, st, disp, H * W, chunks, w.scratch);
    hipLaunchKernelGGL(smooth_mid_kernel, dim3(B), dim3(64), 0, st, p, 1, loss_out);
    rc = launch_status();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(smooth_kernel, dim3(B * chunks), dim3(256), 0, st, p);
  hipLaunchKernelGGL(smooth_mid_kernel, dim3(B + 1), dim3(64), 0, st, p, 2, loss_out);
  rc = launch_status();
  if (rc || !g_disp) return rc;
  if (normalise) hipLaunchKernelGGL(smooth_finish_kernel, dim3(B * chunks), dim3(256), 0, st, p, g_disp);
  return launch_status();
}

static int distil_epilogue_impl(const float* multi_depth, const float* mono_depth, const float* multi_reproj,
                                const float* mono_reproj, const float* ens_reproj, const float* ext_mask, int B, int H, int W,
                                int flags, double* sums, float* g_multi_cons, float* g_multi_distil, float* g_mono_distil,
                                float* consistency_target, const float* ens_depth, float* g_ens_depth, void* ws,
                                size_t ws_bytes, void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (!multi_depth || !mono_depth || !multi_reproj || !mono_reproj || !sums || !ws) return MAL_EINVAL;
  if (ens_depth && (!ens_reproj || (flags & MAL_F_DUAL_DISTIL))) return MAL_EINVAL;  // the learnt ensemble is a third candidate
  if (g_ens_depth && !ens_depth) return MAL_EINVAL;
  Workspace w = carve(ws, B, H, W);
  if (ws_bytes < w.bytes) return MAL_EWORKSPACE;
  DistilParams p = {multi_depth, mono_depth, multi_reproj, mono_reproj, ens_reproj, ext_mask, (size_t)B * H * W, flags,
                    w.scratch, g_multi_cons, g_multi_distil, g_mono_distil, consistency_target, ens_depth, g_ens_depth};
  const int grid = ew_grid2(p.n, 2048);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(distil_kernel, dim3(grid), dim3(256), 0, st, p);
  rc = launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, w.scratch, grid, 2, 2, sums + 2);
  return launch_status();
}

extern "C" int mal_distil_epilogue(const float* multi_depth, const float* mono_depth, const float* multi_reproj,
                                   const float* mono_reproj, const float* ens_reproj, const float* ext_mask, int B,
                                   int H, int W, int flags, double* sums, float* g_multi_cons, float* g_multi_distil,
                                   float* g_mono_distil, float* consistency_target, void* ws, size_t ws_bytes,
                                   void* stream) {
  return distil_epilogue_impl(multi_depth, mono_depth, multi_reproj, mono_reproj, ens_reproj, ext_mask, B, H, W, flags, sums,
                              g_multi_cons, g_multi_distil, g_mono_distil, consistency_target, nullptr, nullptr, ws, ws_bytes,
                              stream);
}

extern "C" int mal_distil_epilogue_learned(const float* multi_depth, const float* mono_depth, const float* ens_depth,
                                           const float* multi_reproj, const float* mono_reproj, const float* ens_reproj,
                                           const float* ext_mask, int B, int H, int W, double* sums, float* g_multi_cons,
                                           float* g_multi_distil, float* g_ens_depth, float* consistency_target, void* ws,
                                           size_t ws_bytes, void* stream) {
  if (!ens_depth) return MAL_EINVAL;
  return distil_epilogue_impl(multi_depth, mono_depth, multi_reproj, mono_reproj, ens_reproj, ext_mask, B, H, W, 0, sums,
                              g_multi_cons, g_multi_distil, nullptr, consistency_target, ens_depth, g_ens_depth, ws, ws_bytes,
                              stream);
}

extern "C" int mal_axpy_maps(int n_terms, const float* const* maps, const float* const* scale,
                             const double* const* denom, const float* mult, const float* denom_eps, size_t n,
                             float* out, int accumulate, void* stream) {
  if (n_terms < 1 || n_terms > 6 || !maps || !out || n == 0) return MAL_EINVAL;
  AxpyParams p = {};
  p.n_terms = n_terms; p.n = n; p.out = out; p.accumulate = accumulate;
  for (int k = 0; k < n_terms; ++k) {
    if (!maps[k]) return MAL_EINVAL;
    p.maps[k] = maps[k];
    p.scale[k] = scale ? scale[k] : nullptr;
    p.denom[k] = denom ? denom[k] : nullptr;
    p.mult[k] = mult ? mult[k] : 1.0f;
    p.eps[k] = denom_eps ? denom_eps[k] : 0.0f;
  }
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid2(n, 2048)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

extern "C" int mal_finish_scalars(const double* num, const double* den, float eps, float mult, int n, float* out,
                                  void* stream) {
  if (!num || !out || n < 1 || n > 64) return MAL_EINVAL;
  hipLaunchKernelGGL(finish_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, num, den, eps, mult, n, out);
  return launch_status();
}

extern "C" int mal_sum_f64(const float* x, size_t n, double* out, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !out || !ws || n == 0) return MAL_EINVAL;
  if (ws_bytes < 2048 * sizeof(double)) return MAL_EWORKSPACE;
  const int grid = ew_grid2(n, 1024);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sum_partial_kernel, dim3(grid), dim3(256), 0, st, x, n, (double*)ws);
  int rc = launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, grid, 1, 1, out);
  return launch_status();
}

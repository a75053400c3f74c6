// The fused pass: depth -> back-projection -> pose/projection -> bilinear border warp of both
// source frames -> 3x3 SSIM + L1 -> per-pixel min over candidates -> automask / external mask
// -> loss sums, AND in the same launch the gradient of those sums with respect to the
// disparity map and the two poses (plus the student's consistency / distillation epilogue).
//
// Replaces, for one pass of manydepth/trainer.py:573-612:
//   Trainer.generate_images_pred            manydepth/trainer.py:1078-1125
//   compute_mono_losses / compute_main_losses (reprojection part)   manydepth/loss_utils.py:57-113,131-199
//   Trainer.generate_images_pred_ensemble   manydepth/trainer.py:1172-1207
//   the distillation / consistency terms    manydepth/loss_utils.py:193-254
//
// Layout.  One 256-thread workgroup produces a 64x16 tile of one sample.  LDS holds the
// target and the two warped sources on the tile grown by 2 pixels (68x20, "region"): SSIM at
// a pixel needs the 3x3 window, and the gradient at a pixel needs the SSIM terms of its 3x3
// neighbours (grown by 1, 66x18, "stats").  Out-of-image window taps are the reflected
// in-image taps (ReflectionPad2d(1), layers.py:238), which always lie inside the region.
// Thread t owns tile pixels t, t+256, t+512, t+768 (wave w -> rows w, w+4, w+8, w+12), so
// every global read/write of a plane is a coalesced 256-byte row segment; the 336 halo
// positions of the region and the 164 ring positions of the stats area are spread over the
// threads as a fifth/sixth step.
//
// Backward without a second pass: the loss is sum(rp*w)/(sum(w)+1e-7); its gradient is the
// gradient of sum(rp*w) times a scalar known only after the global reduction, so the kernel
// writes the unnormalised map and the caller's assembly kernel (mal_axpy_maps) applies the
// scalar.  Per stats pixel the three SSIM partials (d/d mu_x, d/d E[x^2], d/d E[xy]) of the
// winning candidate are put in LDS one colour channel at a time (6 planes, 28 KB); a tile
// pixel then gathers its 3x3 neighbourhood (the adjoint of the reflect-padded box filter:
// border neighbours count twice) and chains through the bilinear taps, the perspective
// divide, the pose and the depth.
#include "mal_common.h"
#include <atomic>
#include "mal_device.h"

namespace mal {

extern std::atomic<hipEvent_t> g_prof_start, g_prof_stop;  // mal_api.hip: one-shot timing hooks

#ifdef MAL_EXPERIMENTS  // the LDS-tiled first formulation (option "pass_impl" 0): not in the default build
constexpr int RW = kTW + 4, RH = kTH + 4, RN = RW * RH;  // 68 x 20 = 1360
constexpr int SW = kTW + 2, SH = kTH + 2, SN = SW * SH;  // 66 x 18 = 1188
constexpr int kHalo = RN - kTW * kTH;                    // 336
constexpr int kRing = SN - kTW * kTH;                    // 164

struct PassParams {
  const float* disp; const float* disp2; const float* K; const float* invK;
  const float* T[2]; const float* src[2];
  const float* target; const float* ident; const float* noise; const float* ext_mask; const float* sample_scale;
  const float* mono_depth; const float* mono_reproj; const float* ens_reproj;
  int B, H, W; float min_disp, range, eps; int convention;
  float* min_reproj; float* g_reproj; float* g_cons; float* g_distil; float* cons_target; float* depth_out;
  double* block_sums; float* block_gP;
  int tiles_x, tiles_y, nblocks, per_xcd;
  int packed;  // src[f] are (B,H,W,kTexel) copies made by mal_pack_texels (MAL_F_SRC_PACKED)
};

struct Own {          // what a tile pixel's owner keeps from phase 1 for the chain rule
  float dxdu[2][3];   // d warped[f][ch] / d u  (border clip folded in)
  float dxdv[2][3];
  float u[2], v[2], rz[2];
};

// halo index (0..335) -> region coordinates
MAL_DEV void halo_pos(int h, int* ry, int* rx) {
  if (h < 2 * RW) { *ry = h / RW; *rx = h % RW; }
  else if (h < 4 * RW) { int k = h - 2 * RW; *ry = kTH + 2 + k / RW; *rx = k % RW; }
  else { int k = h - 4 * RW; *ry = 2 + (k >> 2); int c = k & 3; *rx = c < 2 ? c : kTW + c; }
}
// ring index (0..163) -> stats coordinates
MAL_DEV void ring_pos(int r, int* sy, int* sx) {
  if (r < SW) { *sy = 0; *sx = r; }
  else if (r < 2 * SW) { *sy = SH - 1; *sx = r - SW; }
  else { int k = r - 2 * SW; *sy = 1 + (k >> 1); *sx = (k & 1) ? SW - 1 : 0; }
}

template <bool GRAD, bool AUTOMASK, bool POSE, bool EPI>
__global__ __launch_bounds__(kThreads, 2) void pass_kernel(PassParams p) {
  __shared__ float s_tgt[3][RN];
  __shared__ float s_wp[2][3][RN];
  __shared__ float s_coef[GRAD ? 6 : 1][GRAD ? SN : 1];
  __shared__ float s_P[2][12];
  __shared__ float s_ik[9];
  __shared__ double s_red[4][4];
  __shared__ float s_redP[POSE ? 4 : 1][24];

  // XCD-aware order: dispatch ids i, i+8, i+16.. share an XCD (and its L2); give each XCD a
  // contiguous run of tiles so halo re-reads and the source rows stay in one L2.
  const int id = blockIdx.x;
  const int blk = (id & 7) * p.per_xcd + (id >> 3);
  if (blk >= p.nblocks) return;
  const int tiles = p.tiles_x * p.tiles_y;
  const int b = blk / tiles;
  const int tt = blk - b * tiles;
  const int tyi = tt / p.tiles_x, txi = tt - tyi * p.tiles_x;
  const int x0 = txi * kTW, y0 = tyi * kTH;
  const int H = p.H, W = p.W, HW = H * W;
  const int tid = threadIdx.x;

  if (tid < 24) {
    int f = tid / 12, e = tid % 12, i = e >> 2, j = e & 3;
    const float* K = p.K + b * 16;
    const float* T = p.T[f] + b * 16;
    float acc = K[i * 4 + 0] * T[0 * 4 + j];
    acc = fma_(K[i * 4 + 1], T[1 * 4 + j], acc);
    acc = fma_(K[i * 4 + 2], T[2 * 4 + j], acc);
    acc = fma_(K[i * 4 + 3], T[3 * 4 + j], acc);
    s_P[f][e] = acc;
  } else if (tid >= 32 && tid < 41) {
    int e = tid - 32;
    s_ik[e] = p.invK[b * 16 + (e / 3) * 4 + (e % 3)];
  }
  __syncthreads();

  const float* disp_b = p.disp + (size_t)b * HW;
  const float* disp2_b = p.disp2 ? p.disp2 + (size_t)b * HW : nullptr;
  // ensemble pass: the disparity is the mean of teacher and student (trainer.py:598)
  auto disp_at = [&](int pix) { return disp2_b ? (disp_b[pix] + disp2_b[pix]) / 2.0f : disp_b[pix]; };

  // ---- phase 1: warp every in-image region position into LDS --------------------------
  Own own[4];
  auto warp_pos = [&](int ry, int rx, Own* o) {
    const int gy = y0 - 2 + ry, gx = x0 - 2 + rx;
    if (gy < 0 || gy >= H || gx < 0 || gx >= W) return;
    const int ridx = ry * RW + rx, pix = gy * W + gx;
    const float depth = depth_of(disp_at(pix), p.min_disp, p.range);
    float ray[3], X[3];
    ray_of(s_ik, (float)gx, (float)gy, ray);
    X[0] = depth * ray[0]; X[1] = depth * ray[1]; X[2] = depth * ray[2];
    float ty3[3];
    load_px3(p.target, p.packed & 2, b, HW, pix, ty3);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) s_tgt[ch][ridx] = ty3[ch];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      Sample s = project_pixel(s_P[f], X, p.eps, W, H, p.convention);
      Taps t = make_taps(s.ix, s.iy, W, H);
      float ta[3], tb[3], tc[3], td[3];
      load_taps(p.src[f], p.packed & 1, b, HW, t, ta, tb, tc, td);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float a = ta[ch], bb = tb[ch], c = tc[ch], d = td[ch];
        s_wp[f][ch][ridx] = blend(t, a, bb, c, d);
        if (GRAD && o) {
          float dx, dy;
          blend_grad(t, a, bb, c, d, &dx, &dy);
          o->dxdu[f][ch] = dx * s.mx;
          o->dxdv[f][ch] = dy * s.my;
        }
      }
      if (GRAD && o) { o->u[f] = s.u; o->v[f] = s.v; o->rz[f] = s.rz; }
    }
  };
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int ty = k * 4 + (tid >> 6), tx = tid & 63;
    warp_pos(ty + 2, tx + 2, GRAD ? &own[k] : nullptr);
  }
  for (int h = tid; h < kHalo; h += kThreads) {
    int ry, rx;
    halo_pos(h, &ry, &rx);
    warp_pos(ry, rx, nullptr);
  }
  __syncthreads();

  // ---- phase 2: SSIM + L1 per stats pixel, min over the two candidates, weights ---------
  float coef[5][3][3];  // [slot][channel][alpha, beta, gamma] of the winning candidate
  int tag[5];           // winner (0/1), or -1 when the slot is outside the image
  float wq[4], rpq[4];  // weight and min reprojection of the owned tile pixels
  double acc_rw = 0.0, acc_w = 0.0, acc_cons = 0.0, acc_dist = 0.0;

  auto stats_px = [&](int sy, int sx, int slot, bool owned) {
    const int gy = y0 - 1 + sy, gx = x0 - 1 + sx;
    tag[slot] = -1;
    if (gy < 0 || gy >= H || gx < 0 || gx >= W) return;
    int ro[3], co[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      ro[d] = (reflect1(gy + d - 1, H) - (y0 - 2)) * RW;
      co[d] = reflect1(gx + d - 1, W) - (x0 - 2);
    }
    const int cidx = ro[1] + co[1];
    float ssum[2] = {0.f, 0.f}, lsum[2] = {0.f, 0.f};
    SsimStats st[2][3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float yv[9];
      float sy_ = 0.f, syy = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float t = s_tgt[ch][ro[i] + co[j]];
          yv[i * 3 + j] = t;
          sy_ += t;
          syy += t * t;
        }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float sx_ = 0.f, sxx = 0.f, sxy = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            float x = s_wp[c][ch][ro[i] + co[j]];
            sx_ += x;
            sxx += x * x;
            sxy += x * yv[i * 3 + j];
          }
        float v = ssim_from_sums(sx_, sy_, sxx, syy, sxy, &st[c][ch]);
        float vc = clamp01(v);
        ssum[c] = (ch == 0) ? vc : ssum[c] + vc;
        float l1 = fabsf(yv[4] - s_wp[c][ch][cidx]);
        lsum[c] = (ch == 0) ? l1 : lsum[c] + l1;
      }
    }
    float r0 = 0.85f * div3_(ssum[0]) + 0.15f * div3_(lsum[0]);
    float r1 = 0.85f * div3_(ssum[1]) + 0.15f * div3_(lsum[1]);
    int win = (r1 < r0) ? 1 : 0;
    float rp = win ? r1 : r0;
    const int pix = gy * W + gx;
    float w = 1.0f;
    if (AUTOMASK) {
      float idn = p.ident[(size_t)b * HW + pix];
      if (p.noise) idn += p.noise[(size_t)b * HW + pix] * 0.00001f;
      w = (rp <= idn) ? 1.0f : 0.0f;
    }
    if (p.ext_mask) w *= p.ext_mask[(size_t)b * HW + pix];
    if (p.sample_scale) w *= p.sample_scale[b];
    tag[slot] = win;
    if (owned) {
      wq[slot] = w;
      rpq[slot] = rp;
      acc_rw += (double)(rp * w);
      acc_w += (double)w;
      if (p.min_reproj) p.min_reproj[(size_t)b * HW + pix] = rp;
    }
    if (GRAD) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const SsimStats& s = win ? st[1][ch] : st[0][ch];
        SsimGrad g = ssim_partials(s);
        // w * 0.85/3 (channel mean) * -1/2 (v = (1-S)/2) * gate * 1/9 (box mean)
        float kk = -w * (0.85f / 3.0f) * 0.5f * s.gate * (1.0f / 9.0f);
        coef[slot][ch][0] = kk * g.dmx;
        coef[slot][ch][1] = kk * 2.0f * g.dsxx;
        coef[slot][ch][2] = kk * g.dsxy;
      }
    }
  };
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int ty = k * 4 + (tid >> 6), tx = tid & 63;
    wq[k] = 0.f; rpq[k] = 0.f;
    stats_px(ty + 1, tx + 1, k, true);
  }
  if (GRAD) {
    tag[4] = -1;
    if (tid < kRing) {
      int sy, sx;
      ring_pos(tid, &sy, &sx);
      stats_px(sy, sx, 4, false);
    }
  }

  // ---- phase 3: gather the SSIM partials back to each tile pixel, chain to disp / pose --
  float gu[4][2], gv[4][2];
  if (GRAD) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { gu[k][0] = gu[k][1] = gv[k][0] = gv[k][1] = 0.f; }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      __syncthreads();  // previous channel's gathers (and phase 2's window reads) are done
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        int sidx;
        if (k < 4) sidx = (k * 4 + (tid >> 6) + 1) * SW + (tid & 63) + 1;
        else {
          if (tid >= kRing) break;
          int sy, sx;
          ring_pos(tid, &sy, &sx);
          sidx = sy * SW + sx;
        }
        float a = 0.f, bq = 0.f, c = 0.f;
        if (tag[k] >= 0) { a = coef[k][ch][0]; bq = coef[k][ch][1]; c = coef[k][ch][2]; }
        bool w1 = tag[k] == 1;
        s_coef[0][sidx] = w1 ? 0.f : a; s_coef[1][sidx] = w1 ? 0.f : bq; s_coef[2][sidx] = w1 ? 0.f : c;
        s_coef[3][sidx] = w1 ? a : 0.f; s_coef[4][sidx] = w1 ? bq : 0.f; s_coef[5][sidx] = w1 ? c : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ty = k * 4 + (tid >> 6), tx = tid & 63;
        const int gy = y0 + ty, gx = x0 + tx;
        if (gy >= H || gx >= W) continue;
        // multiplicity of neighbour (dy,dx): 2 where the reflected border tap lands on this pixel
        float wy[3] = {gy == 1 ? 2.f : 1.f, 1.f, gy == H - 2 ? 2.f : 1.f};
        float wx[3] = {gx == 1 ? 2.f : 1.f, 1.f, gx == W - 2 ? 2.f : 1.f};
        float S[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int sidx = (ty + i) * SW + tx + j;
            const float wgt = wy[i] * wx[j];
#pragma unroll
            for (int m = 0; m < 6; ++m) S[m] = fma_(wgt, s_coef[m][sidx], S[m]);
          }
        const int ridx = (ty + 2) * RW + tx + 2;
        const float yq = s_tgt[ch][ridx];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float xq = s_wp[c][ch][ridx];
          float g = S[3 * c] + S[3 * c + 1] * xq + S[3 * c + 2] * yq;
          if (tag[k] == c) {
            float df = xq - yq;
            float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            g += wq[k] * (0.15f / 3.0f) * sg;
          }
          gu[k][c] = fma_(g, own[k].dxdu[c][ch], gu[k][c]);
          gv[k][c] = fma_(g, own[k].dxdv[c][ch], gv[k][c]);
        }
      }
    }
  }

  // ---- per tile pixel: chain rule to disparity and pose, epilogue terms -----------------
  float gP[POSE ? 24 : 1];
  if (POSE)
#pragma unroll
    for (int i = 0; i < 24; ++i) gP[i] = 0.f;
  if (GRAD || EPI || p.depth_out) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ty = k * 4 + (tid >> 6), tx = tid & 63;
      const int gy = y0 + ty, gx = x0 + tx;
      if (gy >= H || gx >= W) continue;
      const int pix = gy * W + gx;
      const size_t gi = (size_t)b * HW + pix;
      const float depth = depth_of(disp_at(pix), p.min_disp, p.range);
      const float ddepth = -(depth * depth) * p.range;  // d depth / d disp
      if (p.depth_out) p.depth_out[gi] = depth;
      if (GRAD) {
        float ray[3], X[4];
        ray_of(s_ik, (float)gx, (float)gy, ray);
        X[0] = depth * ray[0]; X[1] = depth * ray[1]; X[2] = depth * ray[2]; X[3] = 1.0f;
        float gX[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const float rz = own[k].rz[f];
          float a[3];
          a[0] = gu[k][f] * rz;
          a[1] = gv[k][f] * rz;
          a[2] = -(gu[k][f] * own[k].u[f] + gv[k][f] * own[k].v[f]) * rz;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) gX[j] = fma_(a[i], s_P[f][i * 4 + j], gX[j]);
            if (POSE)
#pragma unroll
              for (int j = 0; j < 4; ++j) gP[f * 12 + i * 4 + j] = fma_(a[i], X[j], gP[f * 12 + i * 4 + j]);
          }
        }
        float gdepth = gX[0] * ray[0] + gX[1] * ray[1] + gX[2] * ray[2];
        p.g_reproj[gi] = gdepth * ddepth;
      }
      if (EPI) {
        const float dm = depth, dmono = p.mono_depth[gi];
        const float m = wq[k], cm = 1.0f - m;
        const float dc = dm - dmono;
        acc_cons += (double)(fabsf(dc) * cm);
        const float r_mono = p.mono_reproj[gi], r_multi = rpq[k];
        int idx = 0;
        float best = r_mono;
        if (p.ens_reproj) {
          float r_ens = p.ens_reproj[gi];
          if (r_ens < best) { best = r_ens; idx = 1; }
          if (r_multi < best) idx = 2;
        } else {
          if (r_multi < best) idx = 2;
        }
        const float ens = (dmono + dm) / 2.0f;
        const float target = idx == 0 ? dmono : (idx == 2 ? dm : ens);
        const float dd = target - dm;
        const float mm = 1.0f - cm;
        acc_dist += (double)(fabsf(dd) * mm);
        if (p.cons_target) p.cons_target[gi] = div_(1.0f, dmono * cm + dm * (1.0f - cm));
        if (GRAD) {
          float sc = dc > 0.f ? 1.f : (dc < 0.f ? -1.f : 0.f);
          p.g_cons[gi] = sc * cm * ddepth;
          float sd = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
          float dtd = idx == 0 ? -1.0f : (idx == 2 ? 0.0f : -0.5f);  // d(target - dm)/d dm
          p.g_distil[gi] = sd * dtd * mm * ddepth;
        }
      }
    }
  }

  // ---- block reduction of the loss sums and the pose gradient ---------------------------
  const int lane = tid & 63, wv = tid >> 6;
  double r0 = wave_sum_d(acc_rw), r1 = wave_sum_d(acc_w);
  double r2 = EPI ? wave_sum_d(acc_cons) : 0.0, r3 = EPI ? wave_sum_d(acc_dist) : 0.0;
  if (lane == 0) { s_red[wv][0] = r0; s_red[wv][1] = r1; s_red[wv][2] = r2; s_red[wv][3] = r3; }
  if (POSE) {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      float v = wave_sum(gP[i]);
      if (lane == 0) s_redP[wv][i] = v;
    }
  }
  __syncthreads();
  if (tid < 4) {
    double v = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    p.block_sums[(size_t)blk * 8 + tid] = v;
  }
  if (POSE && tid >= 64 && tid < 88) {
    int i = tid - 64;
    p.block_gP[(size_t)blk * 24 + i] = (s_redP[0][i] + s_redP[1][i]) + (s_redP[2][i] + s_redP[3][i]);
  }
}

#endif  // MAL_EXPERIMENTS

// blocks 0..7: sums[j] = sum over workgroups (fixed order; j >= nsums -> 0); blocks 8..8+B-1: g_T[f][b] = K_b^T [gP_fb ; 0]
__global__ __launch_bounds__(256) void pass_finalize_kernel(const double* block_sums, const float* block_gP,
                                                            const float* K, int nblocks, int tiles, int B,
                                                            double* sums, float* gT0, float* gT1, int nsums,
                                                            BoundaryFold fold, int fold_first) {
  __shared__ double s_part[256];
  __shared__ double s_gP[24];
  const int tid = threadIdx.x;
  if (fold.bnd && (int)blockIdx.x >= fold_first) {
    // one-row halo of the marching gradient pass: add the neighbouring task's scratch row to a segment's first / last row
    const int i = blockIdx.x - fold_first, which = i & 1, bs = i >> 1, seg = bs % fold.segs, b = bs / fold.segs;
    if ((which == 0 && seg == 0) || (which == 1 && seg == fold.segs - 1)) return;
    const int y = which == 0 ? seg * fold.rows : min(seg * fold.rows + fold.rows, fold.H) - 1;
    float* g = fold.g + ((size_t)b * fold.H + y) * fold.W;
    const float* r = fold.bnd + ((size_t)(b * fold.segs + seg) * 2 + which) * fold.W;
    for (int x = tid; x < fold.W; x += 256) g[x] += r[x];
    return;
  }
  if (blockIdx.x < 8) {
    const int j = blockIdx.x;
    double acc = 0.0;
    if (j < nsums)
      for (int i = tid; i < nblocks; i += 256) acc += block_sums[(size_t)i * 8 + j];
    s_part[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) s_part[tid] += s_part[tid + s];
      __syncthreads();
    }
    if (tid == 0) sums[j] = s_part[0];
    return;
  }
  const int b = blockIdx.x - 8;
  if (!gT0 || b >= B) return;
  // 24 sums of `tiles` partials: 8 lanes per value, then an 8-lane tree (fixed order)
  const int v = tid >> 3, sub = tid & 7;
  double acc = 0.0;
  if (v < 24)
    for (int t = sub; t < tiles; t += 8) acc += (double)block_gP[((size_t)b * tiles + t) * 24 + v];
  s_part[tid] = acc;
  __syncthreads();
  if (v < 24 && sub == 0) {
    double a = 0.0;
    for (int k = 0; k < 8; ++k) a += s_part[tid + k];
    s_gP[v] = a;
  }
  __syncthreads();
  if (tid < 32) {
    int f = tid >> 4, e = tid & 15, k = e >> 2, j = e & 3;
    const float* Kb = K + b * 16;
    double a = 0.0;
    for (int i = 0; i < 3; ++i) a += (double)Kb[i * 4 + k] * s_gP[f * 12 + i * 4 + j];
    float* out = f ? gT1 : gT0;
    out[b * 16 + e] = (float)a;
  }
}

int launch_pass_finalize(const double* block_sums, const float* block_gP, const float* K, int nblocks,
                         int blocks_per_sample, int B, double* sums, float* gT0, float* gT1, hipStream_t st,
                         int nsums, const BoundaryFold* fold) {
  const int base = gT0 ? 8 + B : 8;
  BoundaryFold f = {};
  if (fold && fold->bnd && fold->g) f = *fold;
  const int extra = f.bnd ? f.B * f.segs * 2 : 0;
  hipLaunchKernelGGL(pass_finalize_kernel, dim3(base + extra), dim3(256), 0, st, block_sums, block_gP, K, nblocks,
                     blocks_per_sample, B, sums, gT0, gT1, nsums, f, base);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MAL_OK : MAL_ELAUNCH;
}

}  // namespace mal

using namespace mal;

#ifdef MAL_EXPERIMENTS
// The LDS-tiled formulation (first version); mal_pass_fused (mal_march.hip) dispatches here when
// mal_set_option("pass_impl", 0).
extern "C" int mal_pass_fused_tiled(const float* disp, const float* disp2, const float* K, const float* inv_K,
                              const float* const* T, const float* const* src, const float* target,
                              const float* ident, const float* noise, const float* ext_mask,
                              const float* sample_scale, const float* mono_depth,
                              const float* mono_reproj, const float* ens_reproj, int B, int H, int W, int F,
                              float min_depth, float max_depth, float eps, int convention, int flags,
                              float* min_reproj, double* sums, float* g_reproj, float* g_cons, float* g_distil,
                              float* const* g_T, float* consistency_target, float* depth_out, void* ws,
                              size_t ws_bytes, void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (F != 2 || !disp || !K || !inv_K || !T || !src || !T[0] || !T[1] || !src[0] || !src[1] || !target || !sums || !ws)
    return MAL_EINVAL;
  if (flags & (MAL_F_NO_SSIM | MAL_F_AVG | MAL_F_DUAL_DISTIL)) return MAL_EINVAL;  // materialising path only
  const bool grad = flags & MAL_F_GRAD, automask = flags & MAL_F_AUTOMASK, pose = flags & MAL_F_POSE_GRAD,
             epi = flags & MAL_F_EPILOGUE;
  if (automask && !ident) return MAL_EINVAL;
  if (grad && !g_reproj) return MAL_EINVAL;
  if (pose && (!grad || !g_T || !g_T[0] || !g_T[1])) return MAL_EINVAL;
  if (epi && (!mono_depth || !mono_reproj || (grad && (!g_cons || !g_distil)))) return MAL_EINVAL;
  if (convention != 0 && convention != 1) return MAL_EINVAL;
  Workspace w = carve(ws, B, H, W);
  if (ws_bytes < w.bytes) return MAL_EWORKSPACE;

  TileGrid tg = tile_grid(B, H, W);
  PassParams p;
  p.disp = disp; p.disp2 = disp2; p.sample_scale = sample_scale; p.K = K; p.invK = inv_K; p.T[0] = T[0]; p.T[1] = T[1]; p.src[0] = src[0]; p.src[1] = src[1];
  p.target = target; p.ident = ident; p.noise = noise; p.ext_mask = ext_mask; p.mono_depth = mono_depth;
  p.mono_reproj = mono_reproj; p.ens_reproj = ens_reproj; p.B = B; p.H = H; p.W = W;
  p.min_disp = (float)(1.0 / (double)max_depth);
  p.range = (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth);
  p.eps = eps; p.convention = convention;
  p.min_reproj = min_reproj; p.g_reproj = g_reproj; p.g_cons = g_cons; p.g_distil = g_distil;
  p.cons_target = consistency_target; p.depth_out = depth_out; p.block_sums = w.block_sums; p.block_gP = w.block_gP;
  p.tiles_x = tg.tiles_x; p.tiles_y = tg.tiles_y; p.nblocks = tg.blocks();
  p.packed = ((flags & MAL_F_SRC_PACKED) ? 1 : 0) | ((flags & MAL_F_TGT_PACKED) ? 2 : 0);
  p.per_xcd = (p.nblocks + 7) / 8;
  dim3 grid(p.per_xcd * 8), block(kThreads);
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t ev0 = g_prof_start.exchange(nullptr), ev1 = g_prof_stop.exchange(nullptr);
  if (ev0) (void)hipEventRecord(ev0, st);
#define MAL_LAUNCH(G, A, P, E) hipLaunchKernelGGL((pass_kernel<G, A, P, E>), grid, block, 0, st, p)
  if (!grad) {
    if (automask) { if (epi) MAL_LAUNCH(false, true, false, true); else MAL_LAUNCH(false, true, false, false); }
    else          { if (epi) MAL_LAUNCH(false, false, false, true); else MAL_LAUNCH(false, false, false, false); }
  } else if (pose) {
    if (automask) { if (epi) MAL_LAUNCH(true, true, true, true); else MAL_LAUNCH(true, true, true, false); }
    else          { if (epi) MAL_LAUNCH(true, false, true, true); else MAL_LAUNCH(true, false, true, false); }
  } else {
    if (automask) { if (epi) MAL_LAUNCH(true, true, false, true); else MAL_LAUNCH(true, true, false, false); }
    else          { if (epi) MAL_LAUNCH(true, false, false, true); else MAL_LAUNCH(true, false, false, false); }
  }
#undef MAL_LAUNCH
  if (ev1) (void)hipEventRecord(ev1, st);
  rc = launch_status();
  if (rc) return rc;
  return launch_pass_finalize(w.block_sums, w.block_gP, K, p.nblocks, tg.tiles_x * tg.tiles_y, B, sums,
                              pose ? g_T[0] : nullptr, pose ? g_T[1] : nullptr, st);
}
#endif  // MAL_EXPERIMENTS

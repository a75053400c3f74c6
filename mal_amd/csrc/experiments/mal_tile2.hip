// The fused pass, LDS-tiled formulation tuned for occupancy ("pass_impl" = 2).
//
// Same tiling as mal_pass.hip (64x16 output pixels per workgroup, target + both warped sources on
// the tile grown by 2 px in LDS, SSIM partials exchanged through 6 LDS planes one colour channel
// at a time), but laid out so that TWICE as many wavefronts are resident per CU:
//   * 512 threads per workgroup, 2 tile pixels per thread (wave w owns rows w and w+8): half the
//     per-thread state, <= 128 VGPRs => 4 waves/SIMD, and two workgroups (2 x 77 KB LDS) per CU
//     so one workgroup's barrier phases overlap the other's;
//   * the pose/intrinsics products live in scalar registers (computed by 33 lanes, broadcast with
//     v_readlane), not in LDS or VGPRs;
//   * the chain rule to the disparity is folded at warp time into 6 numbers per pixel
//     (d warped[f][ch] / d disp); only the teacher pass (pose gradient) keeps d/du, d/dv;
//   * SSIM is evaluated on the window sums (numerator and denominator scaled by 81^2): no
//     divisions by 9, one reciprocal per candidate-channel.
// Window sums keep ATen's row-major order (layers.py:243-257 via AvgPool2d).
#ifdef MAL_EXPERIMENTS  // the 512-thread LDS-tiled formulation (option "pass_impl" 2): not in the default build
#include "mal_common.h"
#include <atomic>
#include "mal_device.h"

namespace mal {

extern std::atomic<hipEvent_t> g_prof_start, g_prof_stop;

namespace t2 {

constexpr int TW = 64, TH = 16;
constexpr int RW = TW + 4, RH = TH + 4, RN = RW * RH;  // 68 x 20
constexpr int SW = TW + 2, SH = TH + 2, SN = SW * SH;  // 66 x 18
constexpr int kHalo = RN - TW * TH;                    // 336
constexpr int kRing = SN - TW * TH;                    // 164
constexpr float kC1s = 81.0f * 0.0001f, kC2s = 81.0f * 0.0009f;

struct Params {
  const float* disp; const float* disp2; const float* K; const float* invK;
  const float* T[2]; const float* src[2];
  const float* target; const float* ident; const float* noise; const float* ext_mask; const float* sample_scale;
  const float* mono_depth; const float* mono_reproj; const float* ens_reproj;
  int B, H, W; float min_disp, range, eps; int convention;
  float* min_reproj; float* g_reproj; float* g_cons; float* g_distil; float* cons_target; float* depth_out;
  double* block_sums; float* block_gP;
  int tiles_x, tiles_y, nblocks, per_xcd;
  int packed;
};

MAL_DEV void halo_pos(int h, int* ry, int* rx) {
  if (h < 2 * RW) { *ry = h / RW; *rx = h % RW; }
  else if (h < 4 * RW) { int k = h - 2 * RW; *ry = TH + 2 + k / RW; *rx = k % RW; }
  else { int k = h - 4 * RW; *ry = 2 + (k >> 2); int c = k & 3; *rx = c < 2 ? c : TW + c; }
}
MAL_DEV void ring_pos(int r, int* sy, int* sx) {
  if (r < SW) { *sy = 0; *sx = r; }
  else if (r < 2 * SW) { *sy = SH - 1; *sx = r - SW; }
  else { int k = r - 2 * SW; *sy = 1 + (k >> 1); *sx = (k & 1) ? SW - 1 : 0; }
}

// un-clamped (1 - S)/2 from the 3x3 window sums; with GRAD the partials of S wrt sum x, sum x^2, sum xy
template <bool GRAD>
MAL_DEV float ssim_sums(float sx, float sy, float sxx, float syy, float sxy, float* dsx, float* dsxx, float* dsxy) {
  const float pxy = sx * sy;
  const float n1 = 2.0f * pxy + kC1s;
  const float n2 = 2.0f * (9.0f * sxy - pxy) + kC2s;
  const float sx2 = sx * sx, sy2 = sy * sy;
  const float d1 = sx2 + sy2 + kC1s;
  const float d2 = (9.0f * sxx - sx2) + (9.0f * syy - sy2) + kC2s;
  const float n = n1 * n2, d = d1 * d2;
  float rd = __builtin_amdgcn_rcpf(d);
  rd = fma_(fma_(-d, rd, 1.0f), rd, rd);
  const float S = n * rd;
  if (GRAD) {
    *dsx = (2.0f * sy * (n2 - n1) - S * (2.0f * sx * (d2 - d1))) * rd;
    *dsxx = -9.0f * S * d1 * rd;
    *dsxy = 18.0f * n1 * rd;
  }
  return (1.0f - S) * 0.5f;
}

template <bool POSE>
struct Own {
  float e[2][3];                                    // d warped[f][ch] / d disp
  float du[POSE ? 2 : 1][3], dv[POSE ? 2 : 1][3];   // d warped / d u, d v (border clip folded in)
  float u[POSE ? 2 : 1], v[POSE ? 2 : 1], rz[POSE ? 2 : 1];
};

template <int NT, bool GRAD, bool AUTOMASK, bool POSE, bool EPI>
__global__ __launch_bounds__(NT, NT / 128) void tile2_kernel(Params p) {
  constexpr int NPX = TW * TH / NT;   // tile pixels per thread
  constexpr int ROWSTEP = NT / 64;    // rows between a thread's pixels
  constexpr int NW = NT / 64;
  __shared__ float s_tgt[3][RN];
  __shared__ float s_wp[2][3][RN];
  __shared__ float s_coef[GRAD ? 6 : 1][GRAD ? SN : 1];
  __shared__ double s_red[NW][4];
  __shared__ float s_redP[POSE ? NW : 1][24];

  const int id = blockIdx.x;
  const int blk = (id & 7) * p.per_xcd + (id >> 3);
  if (blk >= p.nblocks) return;
  const int tiles = p.tiles_x * p.tiles_y;
  const int b = blk / tiles;
  const int tt = blk - b * tiles;
  const int tyi = tt / p.tiles_x, txi = tt - tyi * p.tiles_x;
  const int x0 = txi * TW, y0 = tyi * TH;
  const int H = p.H, W = p.W, HW = H * W;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

  // P = (K T)[:3,:] per frame and inv_K[:3,:3] in scalar registers
  float P[2][12], ik[9];
  {
    float mine = 0.f;
    if (lane < 24) {
      const int f = lane / 12, e = lane % 12, i = e >> 2, j = e & 3;
      const float* K = p.K + b * 16;
      const float* T = p.T[f] + b * 16;
      float acc = K[i * 4 + 0] * T[0 * 4 + j];
      acc = fma_(K[i * 4 + 1], T[1 * 4 + j], acc);
      acc = fma_(K[i * 4 + 2], T[2 * 4 + j], acc);
      mine = fma_(K[i * 4 + 3], T[3 * 4 + j], acc);
    } else if (lane >= 32 && lane < 41) {
      const int e = lane - 32;
      mine = p.invK[b * 16 + (e / 3) * 4 + (e % 3)];
    }
    const int bits = __builtin_bit_cast(int, mine);
#pragma unroll
    for (int e = 0; e < 24; ++e) P[e / 12][e % 12] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, e));
#pragma unroll
    for (int e = 0; e < 9; ++e) ik[e] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 32 + e));
  }

  const float* disp_b = p.disp + (size_t)b * HW;
  const float* disp2_b = p.disp2 ? p.disp2 + (size_t)b * HW : nullptr;
  const size_t map_b = (size_t)b * HW;

  // ---- phase 1: warp every in-image region position into LDS ------------------------------
  Own<POSE> own[NPX];
  float Xq[POSE ? NPX : 1][3];
  auto warp_pos = [&](int ry, int rx, int k /* owned slot or -1 */) {
    const int gy = y0 - 2 + ry, gx = x0 - 2 + rx;
    if (gy < 0 || gy >= H || gx < 0 || gx >= W) return;
    const int ridx = ry * RW + rx, pix = gy * W + gx;
    float dv_ = disp_b[pix];
    if (disp2_b) dv_ = (dv_ + disp2_b[pix]) / 2.0f;
    const float depth = depth_of(dv_, p.min_disp, p.range);
    float ray[3], X[3];
    ray_of(ik, (float)gx, (float)gy, ray);
    X[0] = depth * ray[0]; X[1] = depth * ray[1]; X[2] = depth * ray[2];
    const float ddepth = -(depth * depth) * p.range;
    float ty3[3];
    load_px3(p.target, p.packed & 2, b, HW, pix, ty3);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) s_tgt[ch][ridx] = ty3[ch];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      Sample s = project_pixel(P[f], X, p.eps, W, H, p.convention);
      Taps t = make_taps(s.ix, s.iy, W, H);
      float ta[3], tb[3], tc[3], td[3];
      load_taps(p.src[f], p.packed & 1, b, HW, t, ta, tb, tc, td);
      float du_dd = 0.f, dv_dd = 0.f;
      if (GRAD && k >= 0) {
        const float c0 = P[f][0] * ray[0] + P[f][1] * ray[1] + P[f][2] * ray[2];
        const float c1 = P[f][4] * ray[0] + P[f][5] * ray[1] + P[f][6] * ray[2];
        const float c2 = P[f][8] * ray[0] + P[f][9] * ray[1] + P[f][10] * ray[2];
        du_dd = (c0 - s.u * c2) * s.rz * ddepth * s.mx;
        dv_dd = (c1 - s.v * c2) * s.rz * ddepth * s.my;
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float a = ta[ch], bb = tb[ch], c = tc[ch], d = td[ch];
        s_wp[f][ch][ridx] = blend(t, a, bb, c, d);
        if (GRAD && k >= 0) {
          float dx, dy;
          blend_grad(t, a, bb, c, d, &dx, &dy);
          own[k].e[f][ch] = dx * du_dd + dy * dv_dd;
          if (POSE) { own[k].du[f][ch] = dx * s.mx; own[k].dv[f][ch] = dy * s.my; }
        }
      }
      if (GRAD && POSE && k >= 0) { own[k].u[f] = s.u; own[k].v[f] = s.v; own[k].rz[f] = s.rz; }
    }
    if (GRAD && POSE && k >= 0) { Xq[k][0] = X[0]; Xq[k][1] = X[1]; Xq[k][2] = X[2]; }
  };
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    if (GRAD) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) own[k].e[f][ch] = 0.f;
    }
    warp_pos(k * ROWSTEP + wv + 2, lane + 2, k);
  }
  for (int h = tid; h < kHalo; h += NT) {
    int ry, rx;
    halo_pos(h, &ry, &rx);
    warp_pos(ry, rx, -1);
  }
  __syncthreads();

  // ---- phase 2: SSIM + L1 per stats pixel, min over the two candidates, weights ------------
  float coef[NPX + 1][GRAD ? 9 : 1];
  int tag[NPX + 1];
  float wq[NPX], rpq[NPX];
  double acc_rw = 0.0, acc_w = 0.0, acc_cons = 0.0, acc_dist = 0.0;
  const float sscale = p.sample_scale ? p.sample_scale[b] : 1.0f;

  auto stats_px = [&](int sy, int sx, int slot, bool owned) {
    const int gy = y0 - 1 + sy, gx = x0 - 1 + sx;
    tag[slot] = -1;
    if (GRAD)
#pragma unroll
      for (int i = 0; i < 9; ++i) coef[slot][i] = 0.f;
    if (gy < 0 || gy >= H || gx < 0 || gx >= W) return;
    int ro[3], co[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      ro[d] = (reflect1(gy + d - 1, H) - (y0 - 2)) * RW;
      co[d] = reflect1(gx + d - 1, W) - (x0 - 2);
    }
    const int cidx = ro[1] + co[1];
    float ssum[2], lsum[2];
    float S[2][3][3], Sy[3][2];  // window sums kept for the winner's partials
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float yv[9];
      float sy_ = 0.f, syy = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float t = s_tgt[ch][ro[i] + co[j]];
          yv[i * 3 + j] = t;
          sy_ += t;
          syy += t * t;
        }
      Sy[ch][0] = sy_; Sy[ch][1] = syy;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float sx_ = 0.f, sxx = 0.f, sxy = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float x = s_wp[c][ch][ro[i] + co[j]];
            sx_ += x;
            sxx += x * x;
            sxy += x * yv[i * 3 + j];
          }
        S[c][ch][0] = sx_; S[c][ch][1] = sxx; S[c][ch][2] = sxy;
        const float vc = clamp01(ssim_sums<false>(sx_, sy_, sxx, syy, sxy, nullptr, nullptr, nullptr));
        ssum[c] = ch == 0 ? vc : ssum[c] + vc;
        const float l1 = fabsf(yv[4] - s_wp[c][ch][cidx]);
        lsum[c] = ch == 0 ? l1 : lsum[c] + l1;
      }
    }
    const float r0 = 0.85f * div3_(ssum[0]) + 0.15f * div3_(lsum[0]);
    const float r1 = 0.85f * div3_(ssum[1]) + 0.15f * div3_(lsum[1]);
    const int win = (r1 < r0) ? 1 : 0;
    const float rp = win ? r1 : r0;
    const size_t gi = map_b + (size_t)gy * W + gx;
    float w = 1.0f;
    if (AUTOMASK) {
      float idn = p.ident[gi];
      if (p.noise) idn += p.noise[gi] * 0.00001f;
      w = (rp <= idn) ? 1.0f : 0.0f;
    }
    if (p.ext_mask) w *= p.ext_mask[gi];
    w *= sscale;
    tag[slot] = win;
    if (owned) {
      wq[slot] = w;
      rpq[slot] = rp;
      acc_rw += (double)(rp * w);
      acc_w += (double)w;
      if (p.min_reproj) p.min_reproj[gi] = rp;
    }
    if (GRAD) {
      const float kk = -w * (0.85f / 3.0f) * 0.5f;  // channel mean, v = (1-S)/2; the 1/9 lives in the sums
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float a, bq, cq;
        const float v = ssim_sums<true>(win ? S[1][ch][0] : S[0][ch][0], Sy[ch][0], win ? S[1][ch][1] : S[0][ch][1],
                                        Sy[ch][1], win ? S[1][ch][2] : S[0][ch][2], &a, &bq, &cq);
        const float g = (v >= 0.0f && v <= 1.0f) ? kk : 0.0f;  // torch.clamp passes gradient on [0,1]
        coef[slot][ch * 3 + 0] = g * a;
        coef[slot][ch * 3 + 1] = g * 2.0f * bq;
        coef[slot][ch * 3 + 2] = g * cq;
      }
    }
  };
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    wq[k] = 0.f; rpq[k] = 0.f;
    stats_px(k * ROWSTEP + wv + 1, lane + 1, k, true);
  }
  tag[NPX] = -1;
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < 9; ++i) coef[NPX][i] = 0.f;
    if (tid < kRing) {
      int sy, sx;
      ring_pos(tid, &sy, &sx);
      stats_px(sy, sx, NPX, false);
    }
  }

  // ---- phase 3: gather the SSIM partials back to each tile pixel ---------------------------
  float gdisp[NPX], gu[POSE ? NPX : 1][2], gv[POSE ? NPX : 1][2];
  if (GRAD) {
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      gdisp[k] = 0.f;
      if (POSE) { gu[k][0] = gu[k][1] = gv[k][0] = gv[k][1] = 0.f; }
    }
    int ring_sidx = 0;
    if (tid < kRing) {
      int sy, sx;
      ring_pos(tid, &sy, &sx);
      ring_sidx = sy * SW + sx;
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      __syncthreads();  // previous channel's gathers (and phase 2's window reads) are done
#pragma unroll
      for (int k = 0; k <= NPX; ++k) {
        if (k == NPX && tid >= kRing) break;
        const int sidx = k < NPX ? (k * ROWSTEP + wv + 1) * SW + lane + 1 : ring_sidx;
        const bool w1 = tag[k] == 1;
        const float a = coef[k][ch * 3], bq = coef[k][ch * 3 + 1], c = coef[k][ch * 3 + 2];
        s_coef[0][sidx] = w1 ? 0.f : a; s_coef[1][sidx] = w1 ? 0.f : bq; s_coef[2][sidx] = w1 ? 0.f : c;
        s_coef[3][sidx] = w1 ? a : 0.f; s_coef[4][sidx] = w1 ? bq : 0.f; s_coef[5][sidx] = w1 ? c : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NPX; ++k) {
        const int ty = k * ROWSTEP + wv, tx = lane;
        const int gy = y0 + ty, gx = x0 + tx;
        if (gy >= H || gx >= W) continue;
        const float wy[3] = {gy == 1 ? 2.f : 1.f, 1.f, gy == H - 2 ? 2.f : 1.f};
        const float wx[3] = {gx == 1 ? 2.f : 1.f, 1.f, gx == W - 2 ? 2.f : 1.f};
        float Sg[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int sidx = (ty + i) * SW + tx + j;
            const float wgt = wy[i] * wx[j];
#pragma unroll
            for (int m = 0; m < 6; ++m) Sg[m] = fma_(wgt, s_coef[m][sidx], Sg[m]);
          }
        const int ridx = (ty + 2) * RW + tx + 2;
        const float yq = s_tgt[ch][ridx];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float xq = s_wp[c][ch][ridx];
          float g = Sg[3 * c] + Sg[3 * c + 1] * xq + Sg[3 * c + 2] * yq;
          if (tag[k] == c) {
            const float df = xq - yq;
            g += wq[k] * (0.15f / 3.0f) * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
          }
          gdisp[k] = fma_(g, own[k].e[c][ch], gdisp[k]);
          if (POSE) {
            gu[k][c] = fma_(g, own[k].du[c][ch], gu[k][c]);
            gv[k][c] = fma_(g, own[k].dv[c][ch], gv[k][c]);
          }
        }
      }
    }
  }

  // ---- per tile pixel: outputs, pose-gradient partials, epilogue terms ----------------------
  float gP[POSE ? 24 : 1];
  if (POSE)
#pragma unroll
    for (int i = 0; i < 24; ++i) gP[i] = 0.f;
  if (GRAD || EPI || p.depth_out) {
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int gy = y0 + k * ROWSTEP + wv, gx = x0 + lane;
      if (gy >= H || gx >= W) continue;
      const int pix = gy * W + gx;
      const size_t gi = map_b + pix;
      if (GRAD) p.g_reproj[gi] = gdisp[k];
      if (POSE) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const float rz = own[k].rz[f];
          const float a[3] = {gu[k][f] * rz, gv[k][f] * rz, -(gu[k][f] * own[k].u[f] + gv[k][f] * own[k].v[f]) * rz};
#pragma unroll
          for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) gP[f * 12 + i * 4 + j] = fma_(a[i], Xq[k][j], gP[f * 12 + i * 4 + j]);
            gP[f * 12 + i * 4 + 3] += a[i];
          }
        }
      }
      if (EPI || p.depth_out) {
        const float dm = depth_of(disp_b[pix], p.min_disp, p.range);
        if (p.depth_out) p.depth_out[gi] = dm;
        if (EPI) {
          const float ddepth = -(dm * dm) * p.range;
          const float dmono = p.mono_depth[gi];
          const float m = wq[k], cm = 1.0f - m, mm = 1.0f - cm;
          const float dc = dm - dmono;
          acc_cons += (double)(fabsf(dc) * cm);
          int idx = 0;
          float best = p.mono_reproj[gi];
          if (p.ens_reproj) {
            const float r_ens = p.ens_reproj[gi];
            if (r_ens < best) { best = r_ens; idx = 1; }
          }
          if (rpq[k] < best) idx = 2;
          const float ens = (dmono + dm) / 2.0f;
          const float target = idx == 0 ? dmono : (idx == 2 ? dm : ens);
          const float dd = target - dm;
          acc_dist += (double)(fabsf(dd) * mm);
          if (p.cons_target) p.cons_target[gi] = div_(1.0f, dmono * cm + dm * (1.0f - cm));
          if (GRAD) {
            const float sc = dc > 0.f ? 1.f : (dc < 0.f ? -1.f : 0.f);
            const float sd = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
            p.g_cons[gi] = sc * cm * ddepth;
            p.g_distil[gi] = sd * (idx == 0 ? -1.0f : (idx == 2 ? 0.0f : -0.5f)) * mm * ddepth;
          }
        }
      }
    }
  }

  // ---- block reduction of the loss sums and the pose gradient -------------------------------
  const double r0 = wave_sum_d(acc_rw), r1 = wave_sum_d(acc_w);
  const double r2 = EPI ? wave_sum_d(acc_cons) : 0.0, r3 = EPI ? wave_sum_d(acc_dist) : 0.0;
  if (lane == 0) { s_red[wv][0] = r0; s_red[wv][1] = r1; s_red[wv][2] = r2; s_red[wv][3] = r3; }
  if (POSE) {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      const float v = wave_sum(gP[i]);
      if (lane == 0) s_redP[wv][i] = v;
    }
  }
  __syncthreads();
  if (tid < 4) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += s_red[w][tid];
    p.block_sums[(size_t)blk * 8 + tid] = v;
  }
  if (POSE && tid >= 64 && tid < 88) {
    const int i = tid - 64;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += s_redP[w][i];
    p.block_gP[(size_t)blk * 24 + i] = v;
  }
}

}  // namespace t2
}  // namespace mal

using namespace mal;

extern "C" int mal_pass_fused_tile2(const float* disp, const float* disp2, const float* K, const float* inv_K,
                                    const float* const* T, const float* const* src, const float* target,
                                    const float* ident, const float* noise, const float* ext_mask,
                                    const float* sample_scale, const float* mono_depth, const float* mono_reproj,
                                    const float* ens_reproj, int B, int H, int W, int F, float min_depth,
                                    float max_depth, float eps, int convention, int flags, float* min_reproj,
                                    double* sums, float* g_reproj, float* g_cons, float* g_distil, float* const* g_T,
                                    float* consistency_target, float* depth_out, void* ws, size_t ws_bytes,
                                    void* stream) {
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
  t2::Params p;
  p.disp = disp; p.disp2 = disp2; p.sample_scale = sample_scale; p.K = K; p.invK = inv_K;
  p.T[0] = T[0]; p.T[1] = T[1]; p.src[0] = src[0]; p.src[1] = src[1];
  p.target = target; p.ident = ident; p.noise = noise; p.ext_mask = ext_mask; p.mono_depth = mono_depth;
  p.mono_reproj = mono_reproj; p.ens_reproj = ens_reproj; p.B = B; p.H = H; p.W = W;
  p.min_disp = (float)(1.0 / (double)max_depth);
  p.range = (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth);
  p.eps = eps; p.convention = convention;
  p.min_reproj = min_reproj; p.g_reproj = g_reproj; p.g_cons = g_cons; p.g_distil = g_distil;
  p.cons_target = consistency_target; p.depth_out = depth_out; p.block_sums = w.block_sums; p.block_gP = w.block_gP;
  p.tiles_x = (W + t2::TW - 1) / t2::TW; p.tiles_y = (H + t2::TH - 1) / t2::TH;
  p.nblocks = p.tiles_x * p.tiles_y * B;
  p.per_xcd = (p.nblocks + 7) / 8;
  p.packed = ((flags & MAL_F_SRC_PACKED) ? 1 : 0) | ((flags & MAL_F_TGT_PACKED) ? 2 : 0);
  constexpr int NT = 512;
  dim3 grid(p.per_xcd * 8), block(NT);
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t ev0 = g_prof_start.exchange(nullptr), ev1 = g_prof_stop.exchange(nullptr);
  if (ev0) (void)hipEventRecord(ev0, st);
#define MAL_LAUNCH(G, A, P, E) hipLaunchKernelGGL((t2::tile2_kernel<NT, G, A, P, E>), grid, block, 0, st, p)
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
  return launch_pass_finalize(w.block_sums, w.block_gP, K, p.nblocks, p.tiles_x * p.tiles_y, B, sums,
                              pose ? g_T[0] : nullptr, pose ? g_T[1] : nullptr, st);
}
#endif  // MAL_EXPERIMENTS

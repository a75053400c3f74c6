// The fused pass, second formulation: register-resident marching (no LDS, no barriers).
//
// One wavefront owns a 64-column strip of one sample and walks down a segment of rows.
//   * lane <-> image column: every plane access (disp, target, masks, outputs) is one coalesced
//     256-byte row piece; the bilinear taps of adjacent lanes fall in adjacent addresses.
//   * horizontal neighbours come from DPP wave shifts (v_add_f32_dpp wave_shr/wave_shl): the 3x3
//     box statistics are separable -- a 3-tap horizontal sum by two DPP adds, then a 3-row
//     vertical sum kept as two running partial sums in registers.
//   * the backward needs, per pixel, the SSIM partials of its 3x3 neighbours: the same
//     separable trick on the 18 partial planes (2 candidates x 3 channels x {d/d sum x,
//     d/d sum x^2, d/d sum xy}), one row behind the statistics; the output row is two rows
//     behind the row being warped.  Reflection padding = lanes/rows outside the image warp the
//     reflected pixel; its adjoint = border neighbours count twice.
//   * strips overlap by the halo (2 columns each side with gradients, 1 without): 60 (62) useful
//     columns per wave; segments overlap by 2 (1) rows each side.
// The SSIM is evaluated on window SUMS (numerator and denominator scaled by 81^2) so no
// division by 9 is needed; with the separable association this differs from ATen's row-major
// order by fp32 reassociation only (tests bound it).
//
// Same entry point and semantics as mal_pass.hip (the LDS-tiled formulation, kept for A/B);
// `mal_set_option("pass_impl", 0|1)` picks.  Replaces for one pass of manydepth/trainer.py:573-612:
// generate_images_pred (:1078-1125), the reprojection/min/automask part of compute_mono_losses
// / compute_main_losses (loss_utils.py:57-113,131-199), generate_images_pred_ensemble
// (:1172-1207) and the consistency/distillation terms (loss_utils.py:193-254).
#include "mal_march.h"
#include "mal_device.h"

namespace mal {

extern hipEvent_t g_prof_start, g_prof_stop;

MAL_DEV float dpp_shr1(float v) {  // lane i <- lane i-1 (lane 0 <- 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
MAL_DEV float dpp_shl1(float v) {  // lane i <- lane i+1 (lane 63 <- 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
MAL_DEV float hsum3(float v) { return (dpp_shr1(v) + v) + dpp_shl1(v); }

constexpr float kC1s = 81.0f * 0.0001f;  // 81 * C1
constexpr float kC2s = 81.0f * 0.0009f;  // 81 * C2

// SSIM from 3x3 window sums; returns the un-clamped (1 - S)/2 and the partials of S wrt the
// window sums of x (the candidate): dS/d(sum x), dS/d(sum x^2), dS/d(sum xy).
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

typedef __attribute__((address_space(4))) const float cfloat;  // constant address space: uniform loads are s_load
constexpr int kCamFloats = 40;  // P[2][12], inv_K 3x3, 7 pad

// the pointer is made opaque per call so the loads stay where they are written (not hoisted out of the
// row loop and spilled)
MAL_DEV void load_cam(const cfloat* cam, float (&P)[2][12], float (&ik)[9]) {
  asm volatile("" : "+s"(cam));
#pragma unroll
  for (int e = 0; e < 24; ++e) P[e / 12][e % 12] = cam[e];
#pragma unroll
  for (int e = 0; e < 9; ++e) ik[e] = cam[24 + e];
}

// one block per sample: P_f = (K T_f)[:3,:] with ATen's bmm association, inv_K[:3,:3]
__global__ void cam_setup_kernel(const float* K, const float* T0, const float* T1, const float* invK, float* cam) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float mine = 0.f;
  if (lane < 24) {
    const int f = lane / 12, e = lane % 12, i = e >> 2, j = e & 3;
    const float* Kb = K + b * 16;
    const float* T = (f ? T1 : T0) + b * 16;
    float acc = Kb[i * 4 + 0] * T[0 * 4 + j];
    acc = fma_(Kb[i * 4 + 1], T[1 * 4 + j], acc);
    acc = fma_(Kb[i * 4 + 2], T[2 * 4 + j], acc);
    mine = fma_(Kb[i * 4 + 3], T[3 * 4 + j], acc);
  } else if (lane < 33) {
    const int e = lane - 24;
    mine = invK[b * 16 + (e / 3) * 4 + (e % 3)];
  }
  if (lane < kCamFloats) cam[b * kCamFloats + lane] = mine;
}

struct WarpRow {        // one pixel of a warped row
  float x[2][3];        // warped candidates
  float y[3];           // target
};
struct DerivRow {       // chain-rule data of that pixel
  float e[2][3];        // d x[f][ch] / d disp  (through u, v, depth; border clip folded in)
  float du[2][3], dv[2][3];  // POSE only: d x / d u, d x / d v
  float rz[2], u[2], v[2];   // POSE only
  float X[3];                // POSE only
};
struct PixInfo { float rp, w; int win; };

// warp one pixel (both frames); DERIV also returns the chain-rule data
template <bool DERIV, bool POSE, class PT>
MAL_DEV void warp_px(const PT& p, const float (&P)[2][12], const float (&ik)[9], int b,
                     int gyr, int gxr, float dispv, WarpRow& w, DerivRow& d) {
  const int W = p.W, H = p.H, HW = H * W, pix = gyr * W + gxr;
  const float depth = depth_of(dispv, p.min_disp, p.range);
  float ray[3], X[3];
  ray_of(ik, (float)gxr, (float)gyr, ray);
  X[0] = depth * ray[0]; X[1] = depth * ray[1]; X[2] = depth * ray[2];
  const float ddepth = -(depth * depth) * p.range;
  load_px3(p.target, p.packed & 2, b, HW, pix, w.y);
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    Sample s = project_pixel(P[f], X, p.eps, W, H, p.convention);
    Taps t = make_taps(s.ix, s.iy, W, H);
    if (p.debug & 1) { t.o00 = t.o01 = t.o10 = t.o11 = pix; }
    float ta[3], tb[3], tc[3], td[3];
    load_taps(p.src[f], p.packed & 1, b, HW, t, ta, tb, tc, td);
    float du_ddisp = 0.f, dv_ddisp = 0.f;
    if (DERIV) {
      // d c_i / d depth = P_i[:3] . ray ;  u = c0/z', v = c1/z'
      const float c0 = P[f][0] * ray[0] + P[f][1] * ray[1] + P[f][2] * ray[2];
      const float c1 = P[f][4] * ray[0] + P[f][5] * ray[1] + P[f][6] * ray[2];
      const float c2 = P[f][8] * ray[0] + P[f][9] * ray[1] + P[f][10] * ray[2];
      du_ddisp = (c0 - s.u * c2) * s.rz * ddepth * s.mx;
      dv_ddisp = (c1 - s.v * c2) * s.rz * ddepth * s.my;
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float a = ta[ch], bb = tb[ch], c = tc[ch], dd = td[ch];
      w.x[f][ch] = blend(t, a, bb, c, dd);
      if (DERIV) {
        float dx, dy;
        blend_grad(t, a, bb, c, dd, &dx, &dy);
        d.e[f][ch] = dx * du_ddisp + dy * dv_ddisp;
        if (POSE) { d.du[f][ch] = dx * s.mx; d.dv[f][ch] = dy * s.my; }
      }
    }
    if (DERIV && POSE) { d.rz[f] = s.rz; d.u[f] = s.u; d.v[f] = s.v; }
  }
  if (DERIV && POSE) { d.X[0] = X[0]; d.X[1] = X[1]; d.X[2] = X[2]; }
}

template <bool GRAD, bool AUTOMASK, bool POSE, bool EPI>
__global__ __launch_bounds__(64, 2) void march_kernel(MarchParams p_kernarg) {
  constexpr int HALO = GRAD ? 2 : 1;
  constexpr int CW = 64 - 2 * HALO;
  // The ~30 pointers and sizes of the parameter block do not fit in scalar registers next to the loop
  // state; rather than let them spill (v_readlane reloads in the row loop) every iteration re-reads the
  // fields it uses from the kernarg segment with scalar loads, through a pointer made opaque per iteration.
  typedef __attribute__((address_space(4))) const MarchParams CParams;
  CParams* const kp0 = (CParams*)__builtin_amdgcn_kernarg_segment_ptr();
  CParams& p = *kp0;

  const int id = blockIdx.x;
  const int task = (id & 7) * p.per_xcd + (id >> 3);
  if (task >= p.ntasks) return;
  const int per_b = p.strips * p.segs;
  const int b = task / per_b;
  const int tt = task - b * per_b;
  const int seg = tt / p.strips, strip = tt - seg * p.strips;  // strips of one segment are consecutive tasks
  const int H = p.H, W = p.W, HW = H * W;
  const int lane = threadIdx.x;
  const int y_lo = seg * p.rows, y_hi = min(y_lo + p.rows, H);

  // P = (K T)[:3,:] of both frames and inv_K[:3,:3] (cam_setup_kernel wrote them, 40 floats per sample).
  // They are fetched with scalar loads where they are used, every iteration, rather than held across
  // the loop: 33 resident scalars overflowed the SGPR file and the spill traffic (v_readlane + hazard
  // nops, ~230 instructions per row) cost more than five s_load per row through the scalar cache.
  const cfloat* cam_b = (const cfloat*)(p.cam + (size_t)b * kCamFloats);

  // ---- lane geometry
  const int gx = strip * CW - HALO + lane;
  const bool in_x = gx >= 0 && gx < W;
  const int gxr = min(max(reflect1(gx, W), 0), W - 1);
  const bool out_x = in_x && lane >= HALO && lane < 64 - HALO;
  // adjoint of the horizontal reflection: what this lane's partials count for when they are
  // shifted to the right neighbour (it is column 0 feeding column 1) / to the left neighbour
  const float sL = gx == 0 ? 2.0f : 1.0f, sR = gx == W - 1 ? 2.0f : 1.0f;
  const bool strip_border = (strip == 0) || (strip * CW - HALO + 63 >= W - 1);  // wave-uniform

  const float* disp_b = p.disp + (size_t)b * HW;
  const float* disp2_b = p.disp2 ? p.disp2 + (size_t)b * HW : nullptr;
  const size_t map_b = (size_t)b * HW;
  const float sscale = p.sample_scale ? p.sample_scale[b] : 1.0f;

  // ---- running state
  float hsA[24], hsB[24];        // partial vertical sums of the horizontal sums: centre r-1 (top+mid), centre r (top)
  float hcA[GRAD ? 18 : 1], hcB[GRAD ? 18 : 1];  // same for the 18 partial planes: output row c-1, c
  WarpRow w1;                    // row r-1 (raw values, for the L1 term of the centre row)
  PixInfo pi1;                   // row c-1 = r-2 (decided one iteration ago)
  float gP[POSE ? 24 : 1];
  float acc_rw = 0.f, acc_w = 0.f, acc_cons = 0.f, acc_dist = 0.f;  // per-lane partials (<= rows terms each)
  // in-sweep smoothness state: normalised / raw disparity of the previous row, its pending gradient
  const bool smooth = GRAD && p.smooth_mean != nullptr;
  const float sm_inv = smooth ? div_(1.0f, (float)p.smooth_mean[b] + 1e-7f) : 0.f;
  const float sm_nx = 1.0f / ((float)p.B * (float)H * (float)(W - 1)), sm_ny = 1.0f / ((float)p.B * (float)(H - 1) * (float)W);
  float sm_n1 = 0.f, sm_d1 = 0.f, sm_g1 = 0.f;
  float acc_sx = 0.f, acc_sy = 0.f, acc_sd = 0.f;  // per-lane partials over <= rows pixels: fp32, widened at the wave sum
#pragma unroll
  for (int i = 0; i < 24; ++i) { hsA[i] = 0.f; hsB[i] = 0.f; }
  if (GRAD)
#pragma unroll
    for (int i = 0; i < 18; ++i) { hcA[i] = 0.f; hcB[i] = 0.f; }
  if (POSE)
#pragma unroll
    for (int i = 0; i < 24; ++i) gP[i] = 0.f;
  pi1.rp = 0.f; pi1.w = 0.f; pi1.win = 0;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) { w1.x[0][ch] = w1.x[1][ch] = 0.f; w1.y[ch] = 0.f; }

  // rows r = y_lo-HALO .. y_hi-1+HALO are warped (reflected when outside the image; row -2 is never
  // needed); the statistics row is c = r-1, the gradient row q = r-2
  // what the gradient row (two iterations behind the warped row) needs of its pixel -- the warped
  // values, the target and the chain-rule numbers -- waits in a 3-slot per-lane LDS ring (no bank
  // conflicts: lane-private dwords; no barrier: one wavefront) instead of ~50 registers or a re-warp
  constexpr int RING = GRAD ? (POSE ? 21 : 15) : 1;
  __shared__ float s_ring[GRAD ? 3 : 1][RING][64];
  int it = 0;
  const int r_first = max(y_lo - HALO, -1), r_last = y_hi - 1 + HALO;
  for (int r = r_first; r <= r_last; ++r, ++it) {
    CParams* kp = kp0;
    asm volatile("" : "+s"(kp));
    CParams& p = *kp;
    // ---- the small per-pixel operands of this iteration's later stages are requested first, so their
    // latency overlaps the warp (loads sitting behind a wave-uniform branch are not hoisted by the compiler)
    const size_t gic = map_b + (size_t)min(max(r - 1, 0), H - 1) * W + gxr;  // statistics row c = r-1
    float ld_ident = 0.f, ld_noise = 0.f, ld_ext = 1.f, ld_mono = 0.f, ld_cost = 1.f;
    if (AUTOMASK) { ld_ident = p.ident[gic]; if (p.noise) ld_noise = p.noise[gic]; }
    if (p.ext_mask) {
      ld_ext = p.ext_mask[gic];
      if (p.lowest_cost) { ld_mono = p.mono_disp[gic]; ld_cost = p.lowest_cost[gic]; }
    }
    const int qe = GRAD ? r - 2 : r - 1;                                     // epilogue row
    const size_t giq = map_b + (size_t)min(max(qe, 0), H - 1) * W + gxr;
    float le_disp = 0.f, le_mono = 0.f, le_mr = 0.f, le_er = 0.f;
    if (EPI) {
      le_disp = disp_b[giq - map_b];
      le_mono = p.mono_disp ? p.mono_disp[giq] : p.mono_depth[giq];
      le_mr = p.mono_reproj[giq];
      if (p.ens_reproj) le_er = p.ens_reproj[giq];
    }
    // ================= stage W: warp row r (reflected if outside the image) ==================
    const int gyr = min(max(reflect1(r, H), 0), H - 1);
    WarpRow w0;
    float dv_ = disp_b[gyr * W + gxr];
    {
      if (disp2_b) dv_ = (dv_ + disp2_b[gyr * W + gxr]) / 2.0f;
      DerivRow d0;
      float P[2][12], ik[9];
      load_cam(cam_b, P, ik);
      warp_px<GRAD, POSE>(p, P, ik, b, gyr, gxr, dv_, w0, d0);
      if (GRAD) {
        float (*slot)[64] = s_ring[it % 3];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            slot[f * 3 + ch][lane] = w0.x[f][ch];
            if (POSE) { slot[9 + f * 3 + ch][lane] = d0.du[f][ch]; slot[15 + f * 3 + ch][lane] = d0.dv[f][ch]; }
            else slot[9 + f * 3 + ch][lane] = d0.e[f][ch];
          }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) slot[6 + ch][lane] = w0.y[ch];
      }
    }

    // ================= smoothness of row r (edges to the right and up), finishing row r-1 ======
    if (smooth) {
      auto sgnf = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
      const bool row_ok = r >= 0 && r < H;
      const float n0 = dv_ * sm_inv;
      // right edge (r,x)-(r,x+1): exists for in-image x with x+1 < W
      const float nR = dpp_shl1(n0);
      const float eR = (fabsf(w0.y[0] - dpp_shl1(w0.y[0])) + fabsf(w0.y[1] - dpp_shl1(w0.y[1]))) +
                       fabsf(w0.y[2] - dpp_shl1(w0.y[2]));
      const bool hx = row_ok && in_x && gx + 1 < W;
      const float wxr = hx ? expf(-(eR * (1.0f / 3.0f))) : 0.f;
      const float dfx = n0 - nR;
      const float sx = sgnf(dfx) * wxr * sm_nx;
      // up edge (r-1,x)-(r,x): exists when both rows are image rows
      const bool vy = row_ok && r >= 1 && in_x;
      const float eU = (fabsf(w1.y[0] - w0.y[0]) + fabsf(w1.y[1] - w0.y[1])) + fabsf(w1.y[2] - w0.y[2]);
      const float wyu_ = vy ? expf(-(eU * (1.0f / 3.0f))) : 0.f;
      const float dfy = sm_n1 - n0;
      const float sy = sgnf(dfy) * wyu_ * sm_ny;
      const int qs = r - 1;  // row finished now: its down edge is this up edge
      if (qs >= y_lo && qs < y_hi && out_x) {
        const float g = sm_g1 + sy;
        p.smooth_gn[map_b + (size_t)qs * W + gxr] = g;
        acc_sy += fabsf(dfy) * wyu_;
        acc_sd += g * sm_d1;
      }
      if (r >= y_lo && r < y_hi && out_x) acc_sx += fabsf(dfx) * wxr;
      sm_g1 = (sx - dpp_shr1(sx)) - sy;
      sm_n1 = n0;
      sm_d1 = dv_;
    }

    // ================= stage H: horizontal 3-sums of row r ====================================
    float h[24];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float y = w0.y[ch];
      h[18 + ch] = hsum3(y);
      h[21 + ch] = hsum3(y * y);
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float x = w0.x[f][ch];
        h[f * 9 + ch * 3 + 0] = hsum3(x);
        h[f * 9 + ch * 3 + 1] = hsum3(x * x);
        h[f * 9 + ch * 3 + 2] = hsum3(x * y);
      }
    }

    // ================= stage S: statistics of centre row c = r-1 ==============================
    const int c = r - 1;
    const bool c_valid = c >= 0 && c < H && c >= y_lo - (HALO - 1) && c <= y_hi - 1 + (HALO - 1);
    PixInfo pi0;
    pi0.rp = 0.f; pi0.w = 0.f; pi0.win = 0;
    float coef[GRAD ? 9 : 1];
    if (GRAD)
#pragma unroll
      for (int i = 0; i < 9; ++i) coef[i] = 0.f;
    if (c_valid) {  // wave-uniform
      float ssum[2], lsum[2];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float sy = hsA[18 + ch] + h[18 + ch], syy = hsA[21 + ch] + h[21 + ch];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const float sx = hsA[f * 9 + ch * 3] + h[f * 9 + ch * 3];
          const float sxx = hsA[f * 9 + ch * 3 + 1] + h[f * 9 + ch * 3 + 1];
          const float sxy = hsA[f * 9 + ch * 3 + 2] + h[f * 9 + ch * 3 + 2];
          const float vc = clamp01(ssim_sums<false>(sx, sy, sxx, syy, sxy, nullptr, nullptr, nullptr));
          ssum[f] = ch == 0 ? vc : ssum[f] + vc;
          const float l1 = fabsf(w1.y[ch] - w1.x[f][ch]);
          lsum[f] = ch == 0 ? l1 : lsum[f] + l1;
        }
      }
      const float r0 = 0.85f * div3_(ssum[0]) + 0.15f * div3_(lsum[0]);
      const float r1 = 0.85f * div3_(ssum[1]) + 0.15f * div3_(lsum[1]);
      pi0.win = (r1 < r0) ? 1 : 0;
      pi0.rp = pi0.win ? r1 : r0;
      float w = 1.0f;
      const size_t gi = map_b + (size_t)c * W + gxr;
      if (AUTOMASK) {
        float idn = ld_ident;
        if (p.noise) idn += ld_noise * 0.00001f;
        w = (pi0.rp <= idn) ? 1.0f : 0.0f;
      }
      if (p.ext_mask) {
        float em = ld_ext;
        if (p.lowest_cost) {  // consistency_mask *= compute_matching_mask (trainer.py:592-593,1066-1076)
          const float mono = depth_of(ld_mono, p.min_disp, p.range);
          const float matching = div_safe_(1.0f, ld_cost);
          const bool ok = (div_safe_(matching - mono, mono) < 1.0f) && (div_safe_(mono - matching, matching) < 1.0f);
          em = ok ? em : em * 0.0f;
          if (p.cmask_out && out_x && c >= y_lo && c < y_hi) p.cmask_out[gi] = em;
        }
        w *= em;
      }
      w *= sscale;
      if (!in_x) w = 0.f;  // not a pixel: contributes nothing (its statistics only served as halo)
      pi0.w = w;
      if (out_x && c >= y_lo && c < y_hi) {
        if (p.min_reproj) p.min_reproj[gi] = pi0.rp;
        acc_rw += pi0.rp * w;
        acc_w += w;
      }
      if (GRAD) {
        // partials of the WINNING candidate only (its window sums are re-formed from the running sums)
        // w * 0.85/3 (channel mean) * -1/2 (v = (1-S)/2); the box mean's 1/9 lives in the sums
        const float kk = -w * (0.85f / 3.0f) * 0.5f;
        const bool w1_ = pi0.win != 0;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float sy = hsA[18 + ch] + h[18 + ch], syy = hsA[21 + ch] + h[21 + ch];
          const float sx = (w1_ ? hsA[9 + ch * 3] : hsA[ch * 3]) + (w1_ ? h[9 + ch * 3] : h[ch * 3]);
          const float sxx = (w1_ ? hsA[9 + ch * 3 + 1] : hsA[ch * 3 + 1]) + (w1_ ? h[9 + ch * 3 + 1] : h[ch * 3 + 1]);
          const float sxy = (w1_ ? hsA[9 + ch * 3 + 2] : hsA[ch * 3 + 2]) + (w1_ ? h[9 + ch * 3 + 2] : h[ch * 3 + 2]);
          float a, bq, cq;
          const float v = ssim_sums<true>(sx, sy, sxx, syy, sxy, &a, &bq, &cq);
          const float g = (v >= 0.0f && v <= 1.0f) ? kk : 0.0f;
          coef[ch * 3 + 0] = g * a;
          coef[ch * 3 + 1] = g * 2.0f * bq;  // d(sum x^2)/dx = 2x
          coef[ch * 3 + 2] = g * cq;
        }
      }
    }

    if (GRAD) {
      // ================= stage HC: horizontal sums of the partial planes of row c ==============
      float hc[18];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const float v = (pi0.win == f) ? coef[i] : 0.f;
          float l = v, rr = v;
          if (strip_border) { l = v * sL; rr = v * sR; }
          hc[f * 9 + i] = (dpp_shr1(l) + v) + dpp_shl1(rr);
        }
      // ================= stage G: output row q = c-1 = r-2 ======================================
      const int q = r - 2;
      if (q >= y_lo && q < y_hi) {  // wave-uniform; q is always inside the image
        WarpRow wq;
        DerivRow dq;
        {
          float (*slot)[64] = s_ring[(it + 1) % 3];  // written two iterations ago
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
              wq.x[f][ch] = slot[f * 3 + ch][lane];
              if (POSE) { dq.du[f][ch] = slot[9 + f * 3 + ch][lane]; dq.dv[f][ch] = slot[15 + f * 3 + ch][lane]; }
              else dq.e[f][ch] = slot[9 + f * 3 + ch][lane];
            }
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) wq.y[ch] = slot[6 + ch][lane];
        }
        float alq[2] = {0.f, 0.f}, beq[2] = {0.f, 0.f};
        if (POSE) {  // the projection of row q re-derived from its disparity (no gathers)
          const float depth = depth_of(disp_b[q * W + gxr], p.min_disp, p.range);
          const float ddepth = -(depth * depth) * p.range;
          float ray[3], P[2][12], ik[9];
          load_cam(cam_b, P, ik);
          ray_of(ik, (float)gxr, (float)q, ray);
          dq.X[0] = depth * ray[0]; dq.X[1] = depth * ray[1]; dq.X[2] = depth * ray[2];
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const Sample sm = project_pixel(P[f], dq.X, p.eps, W, H, p.convention);
            dq.rz[f] = sm.rz; dq.u[f] = sm.u; dq.v[f] = sm.v;
            const float c0 = P[f][0] * ray[0] + P[f][1] * ray[1] + P[f][2] * ray[2];
            const float c1 = P[f][4] * ray[0] + P[f][5] * ray[1] + P[f][6] * ray[2];
            const float c2 = P[f][8] * ray[0] + P[f][9] * ray[1] + P[f][10] * ray[2];
            alq[f] = (c0 - sm.u * c2) * sm.rz * ddepth;  // d u / d disp (the clip gate is inside du, dv)
            beq[f] = (c1 - sm.v * c2) * sm.rz * ddepth;
          }
        }
        const float wyd = (q == H - 2) ? 2.0f : 1.0f;  // row q+1 = c is the bottom border row
        float gdisp = 0.f, gu[2] = {0.f, 0.f}, gv[2] = {0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float SA = fma_(wyd, hc[f * 9 + ch * 3], hcA[f * 9 + ch * 3]);
            const float SB = fma_(wyd, hc[f * 9 + ch * 3 + 1], hcA[f * 9 + ch * 3 + 1]);
            const float SC = fma_(wyd, hc[f * 9 + ch * 3 + 2], hcA[f * 9 + ch * 3 + 2]);
            const float xq = wq.x[f][ch], yq = wq.y[ch];
            float g = SA + SB * xq + SC * yq;
            if (pi1.win == f) {
              const float df = xq - yq;
              g += pi1.w * (0.15f / 3.0f) * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
            }
            if (POSE) { gu[f] = fma_(g, dq.du[f][ch], gu[f]); gv[f] = fma_(g, dq.dv[f][ch], gv[f]); }
            else gdisp = fma_(g, dq.e[f][ch], gdisp);
          }
        if (POSE) gdisp = (gu[0] * alq[0] + gv[0] * beq[0]) + (gu[1] * alq[1] + gv[1] * beq[1]);
        const size_t gi = map_b + (size_t)q * W + gxr;
        if (out_x) p.g_reproj[gi] = gdisp;
        if (POSE && out_x) {
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const float a0 = gu[f] * dq.rz[f], a1 = gv[f] * dq.rz[f];
            const float a2 = -(gu[f] * dq.u[f] + gv[f] * dq.v[f]) * dq.rz[f];
            const float a[3] = {a0, a1, a2};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
              for (int j = 0; j < 3; ++j) gP[f * 12 + i * 4 + j] = fma_(a[i], dq.X[j], gP[f * 12 + i * 4 + j]);
              gP[f * 12 + i * 4 + 3] += a[i];
            }
          }
        }
      }
      // roll the partial-plane sums: (row c: top+mid) <- (row c: top) + hc(c) ; (row c+1: top) <- hc(c)
      {
        const float wyu = (c == 0) ? 2.0f : 1.0f;  // hc(c) as the TOP neighbour of row c+1: doubled if c is row 0
#pragma unroll
        for (int i = 0; i < 18; ++i) {
          hcA[i] = hcB[i] + hc[i];
          hcB[i] = wyu * hc[i];
        }
      }
    }

    // ================= epilogue terms of output row (EPI): row q = r-2 with GRAD, c without ===
    if (EPI) {
      const int q = GRAD ? r - 2 : r - 1;
      const PixInfo& pq = GRAD ? pi1 : pi0;
      if (q >= y_lo && q < y_hi && out_x) {
        const size_t gi = map_b + (size_t)q * W + gxr;
        const float dm = depth_of(le_disp, p.min_disp, p.range);
        const float ddepth = -(dm * dm) * p.range;
        const float dmono = p.mono_disp ? depth_of(le_mono, p.min_disp, p.range) : le_mono;
        const float m = pq.w, cm = 1.0f - m, mm = 1.0f - cm;
        const float dc = dm - dmono;
        acc_cons += fabsf(dc) * cm;
        int idx = 0;
        float best = le_mr;
        if (p.ens_reproj) {
          const float r_ens = le_er;
          if (r_ens < best) { best = r_ens; idx = 1; }
        }
        if (pq.rp < best) idx = 2;
        const float ens = (dmono + dm) / 2.0f;
        const float target = idx == 0 ? dmono : (idx == 2 ? dm : ens);
        const float dd = target - dm;
        acc_dist += fabsf(dd) * mm;
        if (p.cons_target) p.cons_target[gi] = div_(1.0f, dmono * cm + dm * (1.0f - cm));
        if (GRAD) {
          const float sc = dc > 0.f ? 1.f : (dc < 0.f ? -1.f : 0.f);
          const float sd = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
          p.g_cons[gi] = sc * cm * ddepth;
          p.g_distil[gi] = sd * (idx == 0 ? -1.0f : (idx == 2 ? 0.0f : -0.5f)) * mm * ddepth;
        }
      }
    }
    if (p.depth_out) {
      const int q = r - 1;
      if (q >= y_lo && q < y_hi && out_x)
        p.depth_out[map_b + (size_t)q * W + gxr] = depth_of(disp_b[q * W + gxr], p.min_disp, p.range);
    }

    // ================= roll the row state =====================================================
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      hsA[i] = hsB[i] + h[i];
      hsB[i] = h[i];
    }
    w1 = w0;
    if (GRAD) pi1 = pi0;
  }

  // ---- per-task partials (fixed-order second stage in pass_finalize_kernel)
  const double r0 = wave_sum_d((double)acc_rw), r1 = wave_sum_d((double)acc_w);
  const double r2 = EPI ? wave_sum_d((double)acc_cons) : 0.0, r3 = EPI ? wave_sum_d((double)acc_dist) : 0.0;
  if (lane == 0) {
    double* o = p.block_sums + (size_t)task * 8;
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
  }
  if (GRAD) {  // smoothness partials (zeros when the term is not folded in)
    const double q0 = wave_sum_d((double)acc_sx), q1 = wave_sum_d((double)acc_sy), q2 = wave_sum_d((double)acc_sd);
    if (lane == 0) { double* o = p.block_sums + (size_t)task * 8; o[4] = q0; o[5] = q1; o[6] = q2; o[7] = 0.0; }
  }
  if (POSE) {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      const float v = wave_sum(gP[i]);
      if (lane == 0) p.block_gP[(size_t)task * 24 + i] = v;
    }
  }
}

// ---- identity term: min over the two raw sources of r(src_f, target), forward only --------------
struct IdentParams { const float* target; const float* src[2]; float* ident; int B, H, W, strips, segs, rows, ntasks, per_xcd; };

__global__ __launch_bounds__(64, 4) void identity_kernel(IdentParams p) {
  constexpr int HALO = 1, CW = 62;
  const int id = blockIdx.x;
  const int task = (id & 7) * p.per_xcd + (id >> 3);
  if (task >= p.ntasks) return;
  const int per_b = p.strips * p.segs;
  const int b = task / per_b, tt = task - b * per_b;
  const int seg = tt / p.strips, strip = tt - seg * p.strips;
  const int H = p.H, W = p.W, HW = H * W, lane = threadIdx.x;
  const int y_lo = seg * p.rows, y_hi = min(y_lo + p.rows, H);
  const int gx = strip * CW - HALO + lane;
  const bool in_x = gx >= 0 && gx < W;
  const int gxr = min(max(reflect1(gx, W), 0), W - 1);
  const bool out_x = in_x && lane >= HALO && lane < 64 - HALO;
  float hsA[24], hsB[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) { hsA[i] = 0.f; hsB[i] = 0.f; }
  float x1[2][3], y1[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) { x1[0][ch] = x1[1][ch] = 0.f; y1[ch] = 0.f; }
  for (int r = max(y_lo - 1, -1); r <= y_hi; ++r) {
    const int gyr = min(max(reflect1(r, H), 0), H - 1), pix = gyr * W + gxr;
    float x0[2][3], y0[3];
    load_px3(p.target, 1, b, HW, pix, y0);
    load_px3(p.src[0], 1, b, HW, pix, x0[0]);
    load_px3(p.src[1], 1, b, HW, pix, x0[1]);
    float h[24];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float y = y0[ch];
      h[18 + ch] = hsum3(y);
      h[21 + ch] = hsum3(y * y);
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float x = x0[f][ch];
        h[f * 9 + ch * 3 + 0] = hsum3(x);
        h[f * 9 + ch * 3 + 1] = hsum3(x * x);
        h[f * 9 + ch * 3 + 2] = hsum3(x * y);
      }
    }
    const int c = r - 1;
    if (c >= y_lo && c < y_hi) {
      float ssum[2], lsum[2];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float sy = hsA[18 + ch] + h[18 + ch], syy = hsA[21 + ch] + h[21 + ch];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const float sx = hsA[f * 9 + ch * 3] + h[f * 9 + ch * 3];
          const float sxx = hsA[f * 9 + ch * 3 + 1] + h[f * 9 + ch * 3 + 1];
          const float sxy = hsA[f * 9 + ch * 3 + 2] + h[f * 9 + ch * 3 + 2];
          const float vc = clamp01(ssim_sums<false>(sx, sy, sxx, syy, sxy, nullptr, nullptr, nullptr));
          ssum[f] = ch == 0 ? vc : ssum[f] + vc;
          const float l1 = fabsf(y1[ch] - x1[f][ch]);
          lsum[f] = ch == 0 ? l1 : lsum[f] + l1;
        }
      }
      const float r0 = 0.85f * div3_(ssum[0]) + 0.15f * div3_(lsum[0]);
      const float r1 = 0.85f * div3_(ssum[1]) + 0.15f * div3_(lsum[1]);
      if (out_x) p.ident[(size_t)b * HW + (size_t)c * W + gxr] = fminf(r0, r1);
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) { hsA[i] = hsB[i] + h[i]; hsB[i] = h[i]; }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) { x1[0][ch] = x0[0][ch]; x1[1][ch] = x0[1][ch]; y1[ch] = y0[ch]; }
  }
}

int g_pass_impl = 1;   // 1 = marching (this file, fastest); 0 = LDS-tiled v1 (mal_pass.hip); 2 = LDS-tiled, 512 threads (mal_tile2.hip)
int g_march_rows = 0;  // output rows per wave task; 0 = pick so that one round of tasks fills the chip
int g_debug = 0;

MarchParams march_params(int B, int H, int W, float min_depth, float max_depth, float eps, int convention) {
  MarchParams p = {};
  p.B = B; p.H = H; p.W = W;
  p.min_disp = (float)(1.0 / (double)max_depth);
  p.range = (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth);
  p.eps = eps; p.convention = convention;
  return p;
}

int march_launch(MarchParams& p, int flags, hipStream_t st) {
  const bool grad = flags & MAL_F_GRAD, automask = flags & MAL_F_AUTOMASK, pose = flags & MAL_F_POSE_GRAD,
             epi = flags & MAL_F_EPILOGUE;
  const int cw = grad ? 60 : 62;
  p.strips = (p.W + cw - 1) / cw;
  int rows = g_march_rows;
  if (rows <= 0) {
    // Every task is one wavefront that lives for (rows + 2*halo) iterations and the kernel runs at two
    // waves per SIMD (<= 256 VGPRs): the shortest makespan is the smallest `rows` whose task count still
    // fits in ONE resident round (CUs x 4 SIMDs x 2).  Measured on MI355X at B=12 192x640: rows 13
    // (1980 tasks <= 2048) 124 us vs rows 16 135 us vs rows 12 (2112 tasks, two rounds) 164 us.
    static int slots = 0;
    if (slots == 0) {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      slots = cus * 8;
    }
    rows = 8;
    while (rows < p.H && (long long)p.B * p.strips * ((p.H + rows - 1) / rows) > slots) ++rows;
  }
  if (rows < 8) rows = 8;  // workspace is sized for 8-row segments
  p.rows = rows;
  p.segs = (p.H + rows - 1) / rows;
  p.ntasks = p.B * p.strips * p.segs;
  p.debug = g_debug;
  p.packed = ((flags & MAL_F_SRC_PACKED) ? 1 : 0) | ((flags & MAL_F_TGT_PACKED) ? 2 : 0);
  p.per_xcd = (p.ntasks + 7) / 8;
  dim3 grid(p.per_xcd * 8), block(64);
  if (!p.cam) return MAL_EINVAL;
  if (!p.cam_ready) {
    hipLaunchKernelGGL(cam_setup_kernel, dim3(p.B), dim3(64), 0, st, p.K, p.T[0], p.T[1], p.invK, p.cam);
    p.cam_ready = 1;
  }
  hipEvent_t ev0 = g_prof_start, ev1 = g_prof_stop;
  g_prof_start = g_prof_stop = nullptr;
  if (ev0) (void)hipEventRecord(ev0, st);
#define MAL_LAUNCH(G, A, P, E) hipLaunchKernelGGL((march_kernel<G, A, P, E>), grid, block, 0, st, p)
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
  return launch_status();
}

int identity_launch(const float* target_packed, const float* src0_packed, const float* src1_packed, int B, int H,
                    int W, float* ident, hipStream_t st) {
  IdentParams p;
  p.target = target_packed; p.src[0] = src0_packed; p.src[1] = src1_packed; p.ident = ident;
  p.B = B; p.H = H; p.W = W;
  p.strips = (W + 61) / 62;
  p.rows = 12;  // 4 waves per SIMD fit (launch bounds): 16 segments x 11 strips x 12 samples = 2112 <= 4096
  p.segs = (H + p.rows - 1) / p.rows;
  p.ntasks = B * p.strips * p.segs;
  p.per_xcd = (p.ntasks + 7) / 8;
  hipLaunchKernelGGL(identity_kernel, dim3(p.per_xcd * 8), dim3(64), 0, st, p);
  return launch_status();
}

}  // namespace mal

using namespace mal;

extern "C" int mal_set_option(const char* name, int value) {
  if (!name) return MAL_EINVAL;
  auto eq = [&](const char* s) { const char* a = name; while (*a && *a == *s) { ++a; ++s; } return *a == *s; };
  if (eq("pass_impl")) { if (value < 0 || value > 2) return MAL_EINVAL; g_pass_impl = value; return MAL_OK; }
  if (eq("debug")) { g_debug = value; return MAL_OK; }
  if (eq("march_rows")) { if (value < 0 || value > 4096) return MAL_EINVAL; g_march_rows = value; return MAL_OK; }
  return MAL_EINVAL;
}

extern "C" int mal_pass_fused_tiled(const float* disp, const float* disp2, const float* K, const float* inv_K,
                                    const float* const* T, const float* const* src, const float* target,
                                    const float* ident, const float* noise, const float* ext_mask,
                                    const float* sample_scale, const float* mono_depth, const float* mono_reproj,
                                    const float* ens_reproj, int B, int H, int W, int F, float min_depth,
                                    float max_depth, float eps, int convention, int flags, float* min_reproj,
                                    double* sums, float* g_reproj, float* g_cons, float* g_distil, float* const* g_T,
                                    float* consistency_target, float* depth_out, void* ws, size_t ws_bytes,
                                    void* stream);

extern "C" int mal_pass_fused_tile2(const float* disp, const float* disp2, const float* K, const float* inv_K,
                                    const float* const* T, const float* const* src, const float* target,
                                    const float* ident, const float* noise, const float* ext_mask,
                                    const float* sample_scale, const float* mono_depth, const float* mono_reproj,
                                    const float* ens_reproj, int B, int H, int W, int F, float min_depth,
                                    float max_depth, float eps, int convention, int flags, float* min_reproj,
                                    double* sums, float* g_reproj, float* g_cons, float* g_distil, float* const* g_T,
                                    float* consistency_target, float* depth_out, void* ws, size_t ws_bytes,
                                    void* stream);

extern "C" int mal_pass_fused(const float* disp, const float* disp2, const float* K, const float* inv_K,
                              const float* const* T, const float* const* src, const float* target,
                              const float* ident, const float* noise, const float* ext_mask,
                              const float* sample_scale, const float* mono_depth, const float* mono_reproj,
                              const float* ens_reproj, int B, int H, int W, int F, float min_depth, float max_depth,
                              float eps, int convention, int flags, float* min_reproj, double* sums, float* g_reproj,
                              float* g_cons, float* g_distil, float* const* g_T, float* consistency_target,
                              float* depth_out, void* ws, size_t ws_bytes, void* stream) {
  if (g_pass_impl == 2)
    return mal_pass_fused_tile2(disp, disp2, K, inv_K, T, src, target, ident, noise, ext_mask, sample_scale,
                                mono_depth, mono_reproj, ens_reproj, B, H, W, F, min_depth, max_depth, eps,
                                convention, flags, min_reproj, sums, g_reproj, g_cons, g_distil, g_T,
                                consistency_target, depth_out, ws, ws_bytes, stream);
  if (g_pass_impl == 0)
    return mal_pass_fused_tiled(disp, disp2, K, inv_K, T, src, target, ident, noise, ext_mask, sample_scale,
                                mono_depth, mono_reproj, ens_reproj, B, H, W, F, min_depth, max_depth, eps,
                                convention, flags, min_reproj, sums, g_reproj, g_cons, g_distil, g_T,
                                consistency_target, depth_out, ws, ws_bytes, stream);
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

  MarchParams p = march_params(B, H, W, min_depth, max_depth, eps, convention);
  p.disp = disp; p.disp2 = disp2; p.sample_scale = sample_scale; p.K = K; p.invK = inv_K;
  p.T[0] = T[0]; p.T[1] = T[1]; p.src[0] = src[0]; p.src[1] = src[1];
  p.target = target; p.ident = ident; p.noise = noise; p.ext_mask = ext_mask; p.mono_depth = mono_depth;
  p.mono_reproj = mono_reproj; p.ens_reproj = ens_reproj;
  p.min_reproj = min_reproj; p.g_reproj = g_reproj; p.g_cons = g_cons; p.g_distil = g_distil;
  p.cons_target = consistency_target; p.depth_out = depth_out; p.block_sums = w.block_sums; p.block_gP = w.block_gP;
  p.cam = w.cam;
  hipStream_t st = (hipStream_t)stream;
  rc = march_launch(p, flags, st);
  if (rc) return rc;
  return launch_pass_finalize(w.block_sums, w.block_gP, K, p.ntasks, p.strips * p.segs, B, sums,
                              pose ? g_T[0] : nullptr, pose ? g_T[1] : nullptr, st);
}

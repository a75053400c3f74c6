// Device-side arithmetic shared by every kernel of libmal_hip.so (gfx950 only).
//
// The translation units are compiled with -ffp-contract=off: a fused multiply-add appears
// only where it is written (fma_), so each formula below states its own rounding points.
// They are chosen to follow what the reference's PyTorch-CPU path executes (probed against
// ATen 2.10 bit for bit in the authoring container, DESIGN.md 2 "Numerics"):
//   * bmm / matmul rows: p0*x0, then fma per further k            (layers.py:164,187)
//   * grid_sample bilinear blend: nw*w, then fma for ne, sw, se    (trainer.py:1122)
//   * unnormalise: ac=True (g+1)*((S-1)/2); ac=False fma(g+1, S/2, -0.5)
//   * avg_pool 3x3: row-major sum of nine taps, then a true division by 9 (layers.py:232)
//   * channel mean: ((c0+c1)+c2)/3 with a true division              (loss_utils.py:51,53)
//   * every other elementwise op rounds once per op (separate ATen kernels upstream).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MAL_DEV __device__ __forceinline__

namespace mal {

MAL_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// a / b rounded to nearest for normal-range operands: hardware reciprocal (1 ulp), one
// Newton step on the reciprocal, then a Markstein correction of the quotient.
MAL_DEV float div_(float a, float b) {
  float y = __builtin_amdgcn_rcpf(b);
  y = fma_(fma_(-b, y, 1.0f), y, y);
  float q = a * y;
  float r = fma_(-b, q, a);
  return fma_(r, y, q);
}

// same, but falls back to the plain hardware quotient when the refinement leaves the
// finite range (b == 0, inf): used for the perspective divide where z can be anything.
MAL_DEV float div_safe_(float a, float b) {
  float y0 = __builtin_amdgcn_rcpf(b);
  float y = fma_(fma_(-b, y0, 1.0f), y0, y0);
  float q = a * y;
  float r = fma_(-b, q, a);
  float q1 = fma_(r, y, q);
  return __builtin_isfinite(q1) ? q1 : a * y0;
}

// a / c for a constant c with rc = RN(1/c): correctly rounded (Markstein) in 3 ops.
MAL_DEV float divc_(float a, float c, float rc) {
  float q = a * rc;
  float r = fma_(-c, q, a);
  return fma_(r, rc, q);
}

MAL_DEV float div9_(float a) { return divc_(a, 9.0f, 0.111111111111111111f); }
MAL_DEV float div3_(float a) { return divc_(a, 3.0f, 0.333333333333333333f); }

MAL_DEV int reflect1(int i, int n) {  // ReflectionPad2d(1): -1 -> 1, n -> n-2
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

// ------------------------------------------------------------------ geometry
// P = (K @ T)[:3, :], row-major 3x4; fma chain over k like the CPU bmm.
MAL_DEV void compose_P(const float* __restrict__ K, const float* __restrict__ T, float* P) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = K[i * 4 + 0] * T[0 * 4 + j];
      acc = fma_(K[i * 4 + 1], T[1 * 4 + j], acc);
      acc = fma_(K[i * 4 + 2], T[2 * 4 + j], acc);
      acc = fma_(K[i * 4 + 3], T[3 * 4 + j], acc);
      P[i * 4 + j] = acc;
    }
}

// disp -> depth (layers.py:14-23): scaled = min_disp + range*disp, depth = 1/scaled
MAL_DEV float depth_of(float disp, float min_disp, float range, float* scaled_out = nullptr) {
  float s = min_disp + range * disp;
  if (scaled_out) *scaled_out = s;
  return div_(1.0f, s);
}

// inv_K[:3,:3] @ [x, y, 1]  (layers.py:164)
MAL_DEV void ray_of(const float* ik /*9, row-major 3x3*/, float x, float y, float* r) {
  for (int i = 0; i < 3; ++i) {
    float acc = ik[3 * i] * x;
    acc = fma_(ik[3 * i + 1], y, acc);
    r[i] = fma_(ik[3 * i + 2], 1.0f, acc);
  }
}

struct Sample {
  float gx, gy;    // normalised grid, what Project3D returns
  float ix, iy;    // clipped sampling position in source pixels
  float mx, my;    // d ix / d u, d iy / d v : 1 inside, 0 where the border clip is active
  float u, v, rz;  // projected pixel, 1 / (z + eps)
};

// Project3D (layers.py:184-199 / dualrefine/layers.py:216-226) followed by grid_sample's
// unnormalise + border clip.  X = depth * ray (the homogeneous 1 is implicit).
MAL_DEV Sample project_pixel(const float* P, const float* X, float eps, int W, int H, int convention) {
  Sample s;
  float c[3];
  for (int i = 0; i < 3; ++i) {
    float acc = P[4 * i] * X[0];
    acc = fma_(P[4 * i + 1], X[1], acc);
    acc = fma_(P[4 * i + 2], X[2], acc);
    c[i] = fma_(P[4 * i + 3], 1.0f, acc);
  }
  float zp = c[2] + eps;
  s.u = div_safe_(c[0], zp);
  s.v = div_safe_(c[1], zp);
  s.rz = __builtin_amdgcn_rcpf(zp);
  float ix, iy;
  if (convention == 0) {
    float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    s.gx = (div_(s.u, wm1) - 0.5f) * 2.0f;
    s.gy = (div_(s.v, hm1) - 0.5f) * 2.0f;
    ix = (s.gx + 1.0f) * (wm1 * 0.5f);
    iy = (s.gy + 1.0f) * (hm1 * 0.5f);
  } else {
    float wf = (float)W, hf = (float)H;
    s.gx = div_(2.0f * (s.u + 0.5f), wf) - 1.0f;
    s.gy = div_(2.0f * (s.v + 0.5f), hf) - 1.0f;
    ix = fma_(s.gx + 1.0f, wf * 0.5f, -0.5f);
    iy = fma_(s.gy + 1.0f, hf * 0.5f, -0.5f);
  }
  float xmax = (float)(W - 1), ymax = (float)(H - 1);
  // fmaxf/fminf drop NaNs, so a degenerate projection still yields an in-range index
  float cx = fminf(fmaxf(ix, 0.0f), xmax);
  float cy = fminf(fmaxf(iy, 0.0f), ymax);
  s.mx = (cx != 0.0f && cx != xmax) ? 1.0f : 0.0f;  // border itself counts as clipped (ATen)
  s.my = (cy != 0.0f && cy != ymax) ? 1.0f : 0.0f;
  s.ix = cx;
  s.iy = cy;
  return s;
}

// unnormalise + clip of a caller-supplied grid value (granular grid_sample entry point)
MAL_DEV float unnormalize_clip(float g, int size, int align_corners, float* mult) {
  float sf, i;
  if (align_corners) {
    sf = (float)(size - 1) * 0.5f;
    i = (g + 1.0f) * sf;
  } else {
    sf = (float)size * 0.5f;
    i = fma_(g + 1.0f, sf, -0.5f);
  }
  float mx = (float)(size - 1);
  float c = fminf(fmaxf(i, 0.0f), mx);
  *mult = (c != 0.0f && c != mx) ? sf : 0.0f;
  return c;
}

struct Taps {
  int o00, o01, o10, o11;    // offsets into one channel plane
  float nw, ne, sw, se;      // weights
  float tx, ty, ex, ey;
};

MAL_DEV Taps make_taps(float ix, float iy, int W, int H) {
  Taps t;
  float x0f = floorf(ix), y0f = floorf(iy);
  t.tx = ix - x0f;
  t.ex = 1.0f - t.tx;
  t.ty = iy - y0f;
  t.ey = 1.0f - t.ty;
  t.nw = t.ey * t.ex;
  t.ne = t.ey * t.tx;
  t.sw = t.ty * t.ex;
  t.se = t.ty * t.tx;
  int x0 = (int)x0f, y0 = (int)y0f;
  int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);  // x0+1 == W only with weight 0
  t.o00 = y0 * W + x0;
  t.o01 = y0 * W + x1;
  t.o10 = y1 * W + x0;
  t.o11 = y1 * W + x1;
  return t;
}

MAL_DEV float blend(const Taps& t, float a, float b, float c, float d) {
  float o = a * t.nw;
  o = fma_(b, t.ne, o);
  o = fma_(c, t.sw, o);
  return fma_(d, t.se, o);
}

// Packed colour images: one texel (r, g, b[, pad]) per pixel, kTexel floats apart.  A tap is ONE per-lane gather.
#ifndef MAL_TEXEL_FLOATS
#define MAL_TEXEL_FLOATS 3  // 12-byte texels: -25 % texel bytes, measured 3 % faster per step than padded 16-byte ones
#endif
constexpr int kTexel = MAL_TEXEL_FLOATS;
#if MAL_TEXEL_FLOATS == 4
typedef float texel_t __attribute__((ext_vector_type(4)));
#else
typedef float texel_t __attribute__((ext_vector_type(3), aligned(4)));  // 12 bytes apart: dwordx3 at 4-byte alignment
#endif
MAL_DEV texel_t ld_texel(const float* base, size_t pix) {
  return *reinterpret_cast<const texel_t*>(reinterpret_cast<const char*>(base) + pix * (kTexel * 4));
}
MAL_DEV texel_t make_texel(float r, float g, float b) {
#if MAL_TEXEL_FLOATS == 4
  return (texel_t){r, g, b, 0.f};
#else
  return (texel_t){r, g, b};
#endif
}
MAL_DEV void st_texel(float* base, size_t pix, texel_t v) {
  *reinterpret_cast<texel_t*>(reinterpret_cast<char*>(base) + pix * (kTexel * 4)) = v;
}

// the four bilinear taps of the three colour channels.  planar: (B,3,H,W), 12 dword gathers;
// packed: (B,H,W,kTexel) made by mal_pack_texels, 4 texel gathers (the per-lane gather instruction
// count, not the bytes, is what the texture-address path charges for)
MAL_DEV void load_taps(const float* src, int packed, int b, int HW, const Taps& t, float* a, float* bb, float* c,
                       float* d) {
  if (packed) {
    const float* sp = src + (size_t)b * HW * kTexel;
    const texel_t A = ld_texel(sp, t.o00), Bv = ld_texel(sp, t.o01), C = ld_texel(sp, t.o10), D = ld_texel(sp, t.o11);
    a[0] = A.x; a[1] = A.y; a[2] = A.z;
    bb[0] = Bv.x; bb[1] = Bv.y; bb[2] = Bv.z;
    c[0] = C.x; c[1] = C.y; c[2] = C.z;
    d[0] = D.x; d[1] = D.y; d[2] = D.z;
  } else {
    const float* sb = src + (size_t)b * 3 * HW;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float* pl = sb + ch * HW;
      a[ch] = pl[t.o00]; bb[ch] = pl[t.o01]; c[ch] = pl[t.o10]; d[ch] = pl[t.o11];
    }
  }
}

// the three colour channels of one pixel of a planar (B,3,H,W) or packed (B,H,W,kTexel) image
MAL_DEV void load_px3(const float* img, int packed, int b, int HW, int pix, float* out) {
  if (packed) {
    const texel_t v = ld_texel(img + (size_t)b * HW * kTexel, pix);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
  } else {
    const float* pl = img + (size_t)b * 3 * HW + pix;
    out[0] = pl[0]; out[1] = pl[HW]; out[2] = pl[2 * (size_t)HW];
  }
}

// d out / d ix and d out / d iy for one channel
MAL_DEV void blend_grad(const Taps& t, float a, float b, float c, float d, float* dx, float* dy) {
  *dx = (b - a) * t.ey + (d - c) * t.ty;
  *dy = (c - a) * t.ex + (d - b) * t.tx;
}

// ------------------------------------------------------------------ SSIM
constexpr float kC1 = 0.0001f;  // 0.01 ** 2
constexpr float kC2 = 0.0009f;  // 0.03 ** 2

struct SsimStats {
  float mu_x, mu_y, sig_x, sig_y, sig_xy;  // as layers.py:247-252
  float n1, n2, d1, d2;
  float gate;  // 1 where the un-clamped value lies in [0,1] (torch.clamp passes gradient there)
};

// sums of the nine taps: sx, sy, sxx, syy, sxy (each a row-major sequential sum)
MAL_DEV float ssim_from_sums(float sx, float sy, float sxx, float syy, float sxy, SsimStats* st) {
  float mu_x = div9_(sx), mu_y = div9_(sy);
  float sig_x = div9_(sxx) - mu_x * mu_x;
  float sig_y = div9_(syy) - mu_y * mu_y;
  float sig_xy = div9_(sxy) - mu_x * mu_y;
  float n1 = 2.0f * mu_x * mu_y + kC1;
  float n2 = 2.0f * sig_xy + kC2;
  float d1 = mu_x * mu_x + mu_y * mu_y + kC1;
  float d2 = sig_x + sig_y + kC2;
  float n = n1 * n2, d = d1 * d2;
  float v = (1.0f - div_(n, d)) / 2.0f;
  if (st) {
    st->mu_x = mu_x; st->mu_y = mu_y; st->sig_x = sig_x; st->sig_y = sig_y; st->sig_xy = sig_xy;
    st->n1 = n1; st->n2 = n2; st->d1 = d1; st->d2 = d2;
    st->gate = (v >= 0.0f && v <= 1.0f) ? 1.0f : 0.0f;
  }
  return v;  // un-clamped; caller clamps to [0,1] (and needs v for the clamp's gradient gate)
}

// Partials of S = n/d wrt (mu_x, E[x^2], E[xy]) and, for the y side, (mu_y, E[y^2]).
// d out / d x(p) for a tap p of the window = k/9 * (dmx + 2*x(p)*dsxx + y(p)*dsxy), with
// k = -1/2 * gate (the clamp passes gradient on [0,1] inclusive, like torch.clamp).
struct SsimGrad { float dmx, dsxx, dsxy, dmy, dsyy; };

MAL_DEV SsimGrad ssim_partials(const SsimStats& s) {
  SsimGrad g;
  float n = s.n1 * s.n2, d = s.d1 * s.d2;
  float rd = div_(1.0f, d);
  float S = n * rd;
  g.dmx = (2.0f * s.mu_y * (s.n2 - s.n1) - S * (2.0f * s.mu_x * (s.d2 - s.d1))) * rd;
  g.dmy = (2.0f * s.mu_x * (s.n2 - s.n1) - S * (2.0f * s.mu_y * (s.d2 - s.d1))) * rd;
  g.dsxx = -S * s.d1 * rd;
  g.dsyy = g.dsxx;
  g.dsxy = 2.0f * s.n1 * rd;
  return g;
}

// one v_med3_f32 instead of canonicalise + max + min; a NaN comes out as 0 either way (fmaxf drops it; med3 of a NaN is the
// minimum of the other two)
#ifdef MAL_CLAMP_MINMAX  // A/B
MAL_DEV float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
#else
MAL_DEV float clamp01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }
#endif

// ------------------------------------------------------------------ reductions
MAL_DEV float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
MAL_DEV double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

}  // namespace mal

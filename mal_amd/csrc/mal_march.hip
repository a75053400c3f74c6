// The fused pass, second formulation: register-resident marching (no barriers).
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
//   * the kernel is bound by VALU issue (a wave64 op takes 4 cycles on the 16-lane SIMD), so the
//     six colour values of a pixel (2 candidates x 3 channels) live as three register PAIRS and the
//     arithmetic on them is packed fp32 (v_pk_fma/mul/add_f32: two values per lane per issue):
//         pair 0 = (r, g) of candidate 0,  pair 1 = (r, g) of candidate 1,  pair 2 = (b0, b1)
//     (r, g) is how a texel arrives from the gather, so no shuffles are needed; everything
//     per candidate (projection, tap weights, pose terms) is a pair over the two candidates.
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
#include <atomic>
#include "mal_device.h"
#include "mal_pairs.h"
#include <mutex>

namespace mal {

extern std::atomic<hipEvent_t> g_prof_start, g_prof_stop;

// -DMAL_STAGE_MARKS (scripts/valu_budget.py): stage boundaries of the row loop as labelled comments in the compiler's
// listing, each behind a scheduling barrier so that no instruction crosses one -- the per-stage instruction budget of
// profiles/r04_valu_budget.json.  The shipped build has neither the comments nor the barriers.
#ifdef MAL_STAGE_MARKS
#define MAL_MARK(n) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MAL_STAGE " #n); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MAL_MARK(n) do { } while (0)
#endif

typedef __attribute__((address_space(4))) const float cfloat;  // constant address space: uniform loads are s_load

// the pointer is made opaque per call so the loads stay where they are written (not hoisted out of the
// row loop and spilled); P[e] = (P_frame0[e], P_frame1[e]) lands in an aligned scalar register pair
MAL_DEV void load_cam(const cfloat* cam, f2 (&P)[12], float (&ik)[9]) {
  asm volatile("" : "+s"(cam));
#pragma unroll
  for (int e = 0; e < 12; ++e) P[e] = (f2){cam[2 * e], cam[2 * e + 1]};
#pragma unroll
  for (int e = 0; e < 9; ++e) ik[e] = cam[24 + e];
}

// one block per sample
__global__ void cam_setup_kernel(const float* K, const float* T0, const float* T1, const float* invK, float* cam) {
  cam_fill(K, T0, T1, invK, cam, blockIdx.x, threadIdx.x);
}

// Colour pairs of one pixel: x[0] = (r,g) of candidate 0, x[1] = (r,g) of candidate 1, x[2] = (b0, b1)
struct WarpRow {
  f2 x[3];
  f2 yrg; float yb;     // target
};
struct DerivRow {       // chain-rule data of that pixel, same pairing
  f2 e[3];              // d x / d disp  (through u, v, depth; border clip folded in)
  f2 du[3], dv[3];      // POSE only: d x / d u, d x / d v
};
struct PixInfo { float rp, w; int win; };

// Project3D + unnormalise + border clip of mal_device.h::project_pixel for both frames at once
struct Sample2 { f2 ix, iy, mx, my, u, v, rz; };

// a / c with y = the refined reciprocal of c that div_ would form (wave-uniform, made once per wave)
MAL_DEV f2 divc2_(f2 a, float c, float y) {
  const f2 q = a * bc(y);
  const f2 r = fma2(bc(-c), q, a);
  return fma2(r, bc(y), q);
}
MAL_DEV float refined_rcp(float b) {
  const float y = __builtin_amdgcn_rcpf(b);
  return fma_(fma_(-b, y, 1.0f), y, y);
}

// rw, rh: refined reciprocals of the grid normalisation's divisors (W-1, H-1 for convention 0; W, H for 1)
MAL_DEV Sample2 project2_tail(const f2 (&c)[3], float eps, int W, int H, int convention, float rw, float rh);
MAL_DEV Sample2 project2(const f2 (&P)[12], const float (&X)[3], float eps, int W, int H, int convention, float rw,
                         float rh) {
  f2 c[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f2 acc = P[4 * i] * bc(X[0]);
    acc = fma2(P[4 * i + 1], bc(X[1]), acc);
    acc = fma2(P[4 * i + 2], bc(X[2]), acc);
    c[i] = acc + P[4 * i + 3];  // == fma(P, 1, acc)
  }
  return project2_tail(c, eps, W, H, convention, rw, rh);
}
// ... with a point per frame (FRAMED passes: the two frames' candidates are warped with two different disparities)
MAL_DEV Sample2 project2(const f2 (&P)[12], const f2 (&X)[3], float eps, int W, int H, int convention, float rw, float rh) {
  f2 c[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f2 acc = P[4 * i] * X[0];
    acc = fma2(P[4 * i + 1], X[1], acc);
    acc = fma2(P[4 * i + 2], X[2], acc);
    c[i] = acc + P[4 * i + 3];
  }
  return project2_tail(c, eps, W, H, convention, rw, rh);
}
MAL_DEV Sample2 project2_tail(const f2 (&c)[3], float eps, int W, int H, int convention, float rw, float rh) {
  Sample2 s;
  const f2 zp = c[2] + bc(eps);
  s.u = div_safe2_(c[0], zp);
  s.v = div_safe2_(c[1], zp);
  s.rz = rcp2(zp);
  f2 ix, iy;
  if (convention == 0) {
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    const f2 gx = (divc2_(s.u, wm1, rw) - bc(0.5f)) * bc(2.0f);
    const f2 gy = (divc2_(s.v, hm1, rh) - bc(0.5f)) * bc(2.0f);
    ix = (gx + bc(1.0f)) * bc(wm1 * 0.5f);
    iy = (gy + bc(1.0f)) * bc(hm1 * 0.5f);
  } else {
    const float wf = (float)W, hf = (float)H;
    const f2 gx = divc2_(bc(2.0f) * (s.u + bc(0.5f)), wf, rw) - bc(1.0f);
    const f2 gy = divc2_(bc(2.0f) * (s.v + bc(0.5f)), hf, rh) - bc(1.0f);
    ix = fma2(gx + bc(1.0f), bc(wf * 0.5f), bc(-0.5f));
    iy = fma2(gy + bc(1.0f), bc(hf * 0.5f), bc(-0.5f));
  }
  const float xmax = (float)(W - 1), ymax = (float)(H - 1);
  // border clamp as ONE v_med3_f32 per value (fminf(fmaxf()) is canonicalise + max + min: 12 instructions for the four);
  // a NaN -- a degenerate projection -- still yields an in-range index: med3 of a NaN is the minimum of the other two, 0,
  // which is what fmaxf made of it
  s.ix = (f2){__builtin_amdgcn_fmed3f(ix.x, 0.0f, xmax), __builtin_amdgcn_fmed3f(ix.y, 0.0f, xmax)};
  s.iy = (f2){__builtin_amdgcn_fmed3f(iy.x, 0.0f, ymax), __builtin_amdgcn_fmed3f(iy.y, 0.0f, ymax)};
  // the border itself counts as clipped (ATen)
  s.mx = (f2){(s.ix.x != 0.0f && s.ix.x != xmax) ? 1.0f : 0.0f, (s.ix.y != 0.0f && s.ix.y != xmax) ? 1.0f : 0.0f};
  s.my = (f2){(s.iy.x != 0.0f && s.iy.x != ymax) ? 1.0f : 0.0f, (s.iy.y != 0.0f && s.iy.y != ymax) ? 1.0f : 0.0f};
  return s;
}

// The warp of one pixel (both frames) in two halves, so that work that does not need the gathered
// texels can be placed between the gather instructions and their first use.
struct PendingWarp {
  texel_t t[2][4];             // the four taps of both frames: (r, g, b[, -])
  f2 nw, ne, sw, se;           // tap weights, pairs over the two frames
  f2 ex, ey, tx, ty;           // DERIV
  f2 mx, my, du_ddisp, dv_ddisp;
  f2 u, v, rz;                 // POSE: the projection, kept for the pose terms of the gradient row
};

// the parameter-block fields the warp needs, read together at the top of an iteration (one scalar-load wait)
struct WarpConsts {
  const float* src[2]; int packed, debug, W, H, convention; float min_disp, range, eps, rw, rh;
  unsigned* dbg; unsigned dbg_n, dbg_off; bool dbg_on;  // DBG: decision planes, plane stride, this pixel's byte offset, lane writes
};
// decision plane `plane` (mal_hip.h MAL_DEC_*), byte offset `boff` of the pixel inside a (B,1,H,W) map
MAL_DEV void dec_store(unsigned* dbg, unsigned n, int plane, unsigned boff, unsigned v) {
  *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(dbg + (size_t)plane * n) + boff) = v;
}

// projection, tap weights and the eight gathers
// (Round 5 tried a next-row PREFETCH here for the cold regime -- one dword per frame of the source row the next iteration's taps
// will newly touch, issued behind the gathers, dropped at the top of the next iteration: +2 vector-memory instructions per row
// made the teacher's pass 9 % slower warm and 12 % slower cold, profiles/r05_gather_prefetch_ab.txt; removed.)
// LEAN (the specialised passes of the whole-step lists: packed texels, H*W*12 < 2^24 checked by march_launch): the byte
// offsets of the four taps are formed in fp32 -- every product and sum below 2^24 is exact -- instead of with 32-bit integer
// multiplies (v_mul_lo_u32 is quarter rate: the compiler does not keep __umul24 for operands it cannot bound)
// FRAMED: frame 1's candidate is warped with its own disparity, dispv1 (frame 0's with dispv)
template <bool DERIV, bool POSE, bool DBG, bool LEAN, bool FRAMED = false, class BeforeGathers>
MAL_DEV void warp_issue(const WarpConsts& p, const f2 (&P)[12], const float (&ik)[9], int b, int gyr, int gxr,
                        float dispv, PendingWarp& w, BeforeGathers before_gathers, float dispv1 = 0.f) {
  static_assert(!FRAMED || POSE, "FRAMED: pose variants only");
  const int W = p.W, H = p.H, HW = H * W, pix = gyr * W + gxr;
  const float depth = depth_of(dispv, p.min_disp, p.range);
  float ray[3], X[3];
  ray_of(ik, (float)gxr, (float)gyr, ray);
  X[0] = depth * ray[0]; X[1] = depth * ray[1]; X[2] = depth * ray[2];
  Sample2 s;
  if (FRAMED) {
    const float depth1 = depth_of(dispv1, p.min_disp, p.range);
    const f2 X2[3] = {(f2){X[0], depth1 * ray[0]}, (f2){X[1], depth1 * ray[1]}, (f2){X[2], depth1 * ray[2]}};
    s = project2(P, X2, p.eps, W, H, p.convention, p.rw, p.rh);
  } else s = project2(P, X, p.eps, W, H, p.convention, p.rw, p.rh);
  MAL_MARK(10);  // projection done; taps, operand requests, gathers follow
  // make_taps of mal_device.h for both frames
  const f2 x0f = (f2){floorf(s.ix.x), floorf(s.ix.y)}, y0f = (f2){floorf(s.iy.x), floorf(s.iy.y)};
  int oo[2][4];
  unsigned bo12[2][4];  // LEAN: byte offsets of the taps inside the sample's texel image
  if (LEAN) {
    const float xm = (float)(W - 1), ym = (float)(H - 1), w12 = (float)(W * kTexel * 4);
    const f2 x1f = (f2){fminf(x0f.x + 1.0f, xm), fminf(x0f.y + 1.0f, xm)};  // x0+1 == W only with weight 0
    const f2 y1f = (f2){fminf(y0f.x + 1.0f, ym), fminf(y0f.y + 1.0f, ym)};
    const f2 r0 = y0f * bc(w12), r1 = y1f * bc(w12), c0 = x0f * bc((float)(kTexel * 4)), c1 = x1f * bc((float)(kTexel * 4));
    const f2 o00 = r0 + c0, o01 = r0 + c1, o10 = r1 + c0, o11 = r1 + c1;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      bo12[f][0] = (unsigned)o00[f]; bo12[f][1] = (unsigned)o01[f]; bo12[f][2] = (unsigned)o10[f]; bo12[f][3] = (unsigned)o11[f];
    }
  }
#pragma unroll
  for (int f = 0; f < 2 && !LEAN; ++f) {
    const int x0 = (int)x0f[f], y0 = (int)y0f[f];
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);  // x0+1 == W only with weight 0
    // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): rows and H*W are below 2^24 (march_launch checks)
    const int r0 = (int)__umul24((unsigned)y0, (unsigned)W), r1 = (int)__umul24((unsigned)y1, (unsigned)W);
    oo[f][0] = r0 + x0; oo[f][1] = r0 + x1; oo[f][2] = r1 + x0; oo[f][3] = r1 + x1;
    if (p.debug & 1) { oo[f][0] = oo[f][1] = oo[f][2] = oo[f][3] = pix; }
    if (DBG && p.dbg_on)
      dec_store(p.dbg, p.dbg_n, MAL_DEC_TAP0 + f, p.dbg_off,
                (unsigned)x0 | ((unsigned)y0 << 12) | (s.mx[f] == 0.f ? 1u << 24 : 0u) | (s.my[f] == 0.f ? 1u << 25 : 0u));
  }
  before_gathers();
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int (&o)[4] = oo[f];
    if (LEAN) {
      const float* sp = p.src[f] + (size_t)b * HW * kTexel;
#pragma unroll
      for (int k = 0; k < 4; ++k) w.t[f][k] = ldt(sp, bo12[f][k]);
    } else if (p.packed & 1) {
      const float* sp = p.src[f] + (size_t)b * HW * kTexel;
#pragma unroll
      for (int k = 0; k < 4; ++k) w.t[f][k] = ldt(sp, __umul24((unsigned)o[k], (unsigned)(kTexel * 4)));
    } else {
      const float* p0 = p.src[f] + (size_t)b * 3 * HW;
      const float* p1 = p0 + HW;
      const float* p2 = p1 + HW;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned bo = (unsigned)o[k] * 4u;
        w.t[f][k] = make_texel(ldf(p0, bo), ldf(p1, bo), ldf(p2, bo));
      }
    }
  }
  w.tx = s.ix - x0f; w.ex = bc(1.0f) - w.tx; w.ty = s.iy - y0f; w.ey = bc(1.0f) - w.ty;
  w.nw = w.ey * w.ex; w.ne = w.ey * w.tx; w.sw = w.ty * w.ex; w.se = w.ty * w.tx;
  w.mx = s.mx; w.my = s.my;
  if (POSE) { w.u = s.u; w.v = s.v; w.rz = s.rz; }
  if (DERIV && !POSE) {
    // d c_i / d depth = P_i[:3] . ray ;  u = c0/z', v = c1/z'
    const float ddepth = -(depth * depth) * p.range;
    const f2 c0 = P[0] * bc(ray[0]) + P[1] * bc(ray[1]) + P[2] * bc(ray[2]);
    const f2 c1 = P[4] * bc(ray[0]) + P[5] * bc(ray[1]) + P[6] * bc(ray[2]);
    const f2 c2 = P[8] * bc(ray[0]) + P[9] * bc(ray[1]) + P[10] * bc(ray[2]);
    w.du_ddisp = (c0 - s.u * c2) * s.rz * bc(ddepth) * s.mx;
    w.dv_ddisp = (c1 - s.v * c2) * s.rz * bc(ddepth) * s.my;
  }
}

// blend (and its derivatives) once the texels are there
template <bool DERIV, bool POSE>
MAL_DEV void warp_finish(PendingWarp& pw, f2 (&x)[3], DerivRow& d) {
  f2 dxb, dyb;  // b channel of both frames
#pragma unroll
  for (int f = 0; f < 2; ++f) {
#pragma unroll
#if defined(MAL_SHADOW) && MAL_SHADOW
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(pw.t[f][k]));  // whole texels: one gather each, not 8+4 bytes (volatile: see SHADOW)
#else
    for (int k = 0; k < 4; ++k) asm("" : "+v"(pw.t[f][k]));  // whole texels: one gather each, not 8+4 bytes
#endif
    const texel_t A = pw.t[f][0], Bv = pw.t[f][1], C = pw.t[f][2], D = pw.t[f][3];
    const f2 a = (f2){A.x, A.y}, bb = (f2){Bv.x, Bv.y}, c = (f2){C.x, C.y}, dd = (f2){D.x, D.y};
    const float nw_ = pw.nw[f], ne_ = pw.ne[f], sw_ = pw.sw[f], se_ = pw.se[f];
    // blend (mal_device.h): nw*a, then fma for ne, sw, se
    f2 o = a * bc(nw_);
    o = fma2(bb, bc(ne_), o);
    o = fma2(c, bc(sw_), o);
    x[f] = fma2(dd, bc(se_), o);
    float oz = A.z * nw_;
    oz = fma_(Bv.z, ne_, oz);
    oz = fma_(C.z, sw_, oz);
    x[2][f] = fma_(D.z, se_, oz);
    if (DERIV) {
      const float ex_ = pw.ex[f], ey_ = pw.ey[f], tx_ = pw.tx[f], ty_ = pw.ty[f];
      const f2 dx = (bb - a) * bc(ey_) + (dd - c) * bc(ty_);
      const f2 dy = (c - a) * bc(ex_) + (dd - bb) * bc(tx_);
      dxb[f] = (Bv.z - A.z) * ey_ + (D.z - C.z) * ty_;
      dyb[f] = (C.z - A.z) * ex_ + (D.z - Bv.z) * tx_;
      if (POSE) { d.du[f] = dx * bc(pw.mx[f]); d.dv[f] = dy * bc(pw.my[f]); }
      else d.e[f] = dx * bc(pw.du_ddisp[f]) + dy * bc(pw.dv_ddisp[f]);
    }
  }
  if (DERIV) {
    if (POSE) { d.du[2] = dxb * pw.mx; d.dv[2] = dyb * pw.my; }
    else d.e[2] = dxb * pw.du_ddisp + dyb * pw.dv_ddisp;
  }
}

// EXPORT (the teacher's pass of the --temporal step, in front of the producer): a gradient pass that also leaves what the
// forward-only pass in front of the producer leaves -- the warped images (twice), min / winner (twice), automask weight.
// The body is shared; the kernels below instantiate it (EXPORT has its own entry point so that the names and the code of
// the other instantiations stay what they were).
// SPEC != 0: specialisations for the passes of the whole-step lists -- packed texels on both sides, no depth map out, no
// experiment switches (kSpecLean), and optional operands fixed at compile time, so that the operand requests, pointer
// selects and branches those options cost in every row are compiled out (march_launch checks that a launch qualifies).
enum : int {
  kSpecLean = 1, kSpecNoDisp2 = 2,
  kSpecExtNo = 4, kSpecExtYes = 8,      // external mask: absent / present (neither: decided per launch)
  kSpecCostNo = 16, kSpecCostYes = 32,  // matching mask (lowest cost + the teacher's disparity): absent / present
  kSpecMonoYes = 64,                    // the teacher's disparity is given as disparity (epilogue)
  kSpecNoScale = 128,                   // no per-sample scale
  kSpecNoNoise = 256,                   // no tie-break noise map (the whole-step lists fold it into the identity map: Philox)
  kSpecFramed = 2048,                   // a disparity per frame (MarchParams::framed; not a LEAN specialisation)
  kSpecConvA = 512, kSpecConvB = 1024,  // Project3D convention 0 (ManyDepth, align_corners=True) / 1 (DualRefine) known at compile time
  kSpecTeacher = kSpecLean | kSpecNoDisp2 | kSpecExtNo | kSpecCostNo | kSpecNoScale | kSpecConvA,
  kSpecStudent = kSpecLean | kSpecNoDisp2 | kSpecExtYes | kSpecCostYes | kSpecMonoYes | kSpecConvA,      // whole-step list, scale 0
  kSpecStudentNoCost = kSpecLean | kSpecNoDisp2 | kSpecExtYes | kSpecCostNo | kSpecMonoYes | kSpecConvA, // four-scale list, scales > 0
  kSpecStudentNoEpi = kSpecLean | kSpecNoDisp2 | kSpecExtYes | kSpecCostYes | kSpecConvA,  // --temporal step: the epilogue is its own launch
  kSpecRefine = kSpecLean | kSpecNoDisp2 | kSpecCostNo | kSpecMonoYes | kSpecNoScale | kSpecConvB        // DualRefine, deq iterations > 0
};
// -DMAL_PROBE_W3 (scripts/w3_probe.sh; NOT a product build -- its results are wrong on purpose): what would a third wave per
// SIMD buy the teacher's gradient pass?  The pass holds 246 VGPRs and 19.4 KB of LDS per wave; three waves need <= 168 and
// <= 13.6 KB.  The probe compiles the SAME row loop under the 168-register cap (the compiler spills the rest to scratch) and
// folds the ring's 24 floats per slot onto 16 (slots 16..23 alias 8..15: same LDS instructions, colliding values), and is
// timed against the shipped kernel: if three waves WITH spill traffic are not faster than two without, no restructuring of
// the row state that reaches 168 registers without spills can be expected to pay either.
#ifdef MAL_PROBE_W3
#define MAL_RI(j) ((j) < 16 ? (j) : (j) - 8)
#define MAL_RING_POSE 16
#define MAL_TEACHER_WAVES 3
#else
#define MAL_RI(j) (j)
#define MAL_RING_POSE 24
#define MAL_TEACHER_WAVES 2
#endif
#ifndef MAL_FIN_UNIFORM
#define MAL_FIN_UNIFORM 1
#endif
template <bool GRAD, bool AUTOMASK, bool POSE, bool EPI, bool DBG, bool TEMPORAL, bool EXPORT, int SPEC = 0>
MAL_DEV void march_body() {
  constexpr bool LEAN = (SPEC & kSpecLean) != 0, NO_DISP2 = (SPEC & kSpecNoDisp2) != 0;
  // --avg_reprojection / --no_ssim (MarchParams::avg, ::no_ssim, wave-uniform): in the generic gradient passes only, so that
  // no instantiation a whole-step list of the default options launches carries the tests
  constexpr bool VARIANTS = GRAD && !LEAN && !TEMPORAL && !EXPORT;
  constexpr bool EXT_NO = (SPEC & kSpecExtNo) != 0, EXT_YES = (SPEC & kSpecExtYes) != 0;
  constexpr bool COST_NO = (SPEC & kSpecCostNo) != 0, COST_YES = (SPEC & kSpecCostYes) != 0;
  constexpr bool MONO_YES = (SPEC & kSpecMonoYes) != 0, NO_SCALE = (SPEC & kSpecNoScale) != 0;
  constexpr bool NO_NOISE = (SPEC & kSpecNoNoise) != 0;
  constexpr bool FRAMED = (SPEC & kSpecFramed) != 0;  // frame -1 warped with disp, frame +1 with disp2; two gradient maps
  static_assert(!FRAMED || (GRAD && POSE && !NO_DISP2 && !EPI && !TEMPORAL && !EXPORT), "FRAMED: pose-gradient pass without epilogue");
#ifdef MAL_CONV_FIXED  // A/B: Project3D's convention fixed at compile time in the specialised passes (measured 1 % SLOWER than the
  constexpr int CONV = (SPEC & kSpecConvA) ? 0 : ((SPEC & kSpecConvB) ? 1 : -1);  // scalar branch: profiles/r04_hsum_variants_ab.txt)
#else
  constexpr int CONV = -1;
#endif
  constexpr bool OUTS = EXPORT || (!GRAD && !EPI);  // the outputs the producer / the fused sweep read
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
  // Rows are LOGICAL rows of this task: odd segments see the sample upside down (physical row = H-1 - logical), so
  // they walk bottom-up and a segment boundary is reached by both of its tasks at the same time -- the halo rows one
  // of them re-reads are then still in the XCD's L2.  Box filter and reflection padding are symmetric under the flip;
  // only addresses and the ray's row use the physical row (prow).
  const int ph_lo = seg * p.rows, ph_hi = min(ph_lo + p.rows, H);
  const bool flip = (seg & 1) != 0 && p.flip_odd != 0;
  const int y_lo = flip ? H - ph_hi : ph_lo, y_hi = flip ? H - ph_lo : ph_hi;
  const int yf = flip ? H - 1 : 0;
  auto prow = [&](int y) { return abs(y - yf); };  // y in [0, H-1]

  if (TEMPORAL && p.region) {
    // Correction sweep of the --temporal step: the pass in front of the producer (EXPORT) left this task's gradient rows,
    // boundary rows and pose partials as they are when no synthesised candidate takes part.  A synthesised image differs
    // from the warped one only inside the region map, so decisions can change within 1 pixel of it and the gradient (3x3
    // windows around changed decisions, cotangents arriving through syn) within 2: a task whose gradient pixels -- rows
    // [ph_lo-1, ph_hi], its own 60 columns -- keep 3 rows / 4 columns away from every marked pixel has nothing to redo.
    const unsigned char* reg = p.region + (size_t)b * HW;
    const int c_lo = strip * CW - HALO - 4;
    const int ca = min(max(c_lo + lane, 0), W - 1), cb = min(max(c_lo + 64 + (lane & 7), 0), W - 1);
    unsigned any = 0u;
    for (int y = max(ph_lo - 3, 0); y < min(ph_hi + 3, H); ++y) any |= (unsigned)reg[y * W + ca] | (unsigned)reg[y * W + cb];
    if (!__any((int)(any & 1u))) return;
  }

  // P = (K T)[:3,:] of both frames and inv_K[:3,:3] (cam_setup_kernel wrote them, 40 floats per sample).
  // They are fetched with scalar loads where they are used, every iteration, rather than held across
  // the loop: 33 resident scalars overflowed the SGPR file and the spill traffic (v_readlane + hazard
  // nops, ~230 instructions per row) cost more than a few s_load per row through the scalar cache.
  const cfloat* cam_b = (const cfloat*)(p.cam + (size_t)b * kCamFloats);

  // ---- lane geometry
  const int gx = strip * CW - HALO + lane;
  const bool in_x = gx >= 0 && gx < W;
  const int gxr = min(max(reflect1(gx, W), 0), W - 1);
  const bool out_x = in_x && lane >= HALO && lane < 64 - HALO;
  // adjoint of the horizontal reflection: what this lane's partials count for when they are
  // shifted to the right neighbour (it is column 0 feeding column 1) / to the left neighbour
#ifdef MAL_EXP_NOREFL  // experiment: what the adjoint's border multipliers cost
  constexpr float sL = 1.0f, sR = 1.0f;
#else
  const float sL = gx == 0 ? 2.0f : 1.0f, sR = gx == W - 1 ? 2.0f : 1.0f;
#endif
  // ... the same weights as the RECEIVING lane sees them (horizontal sums through LDS, below): what arrives from the left
  // neighbour counts twice in column 1, what arrives from the right one in column W-2
  const float wL = gx == 1 ? 2.0f : 1.0f, wR = gx == W - 2 ? 2.0f : 1.0f;
  (void)sL; (void)sR; (void)wL; (void)wR;

  // (B,1,H,W) maps are addressed as (kernel-argument pointer) + (32-bit per-lane byte offset that already holds the
  // sample's base): no 64-bit pointer arithmetic per map and row (check_shape bounds a map below 2^31 bytes)
  const float* disp_b = p.disp;
  const float* disp2_b = NO_DISP2 ? nullptr : p.disp2;
  const unsigned lane_off = (unsigned)gxr * 4u + (unsigned)b * (unsigned)HW * 4u;
  auto moff = [&](int row) { return (unsigned)(prow(row) * W) * 4u + lane_off; };  // row in [0, H-1] (logical)
  const float sscale = (!NO_SCALE && p.sample_scale) ? (p.sample_scale_is_mask ? 1.0f - p.sample_scale[b] : p.sample_scale[b]) : 1.0f;

  // ---- running state (pairs as in WarpRow: index k*3 + {x, x^2, xy} for the colour pairs k)
  // The horizontal sums of the two previous rows: A = row r-2, B = row r-1.  The window sum of centre row r-1 is
  // formed as (A + h(r)) + B -- outer rows first -- so that it is the SAME fp32 number whether the task walks down or
  // up: a task recomputes its neighbour's boundary rows as halo, and a near-tie decision (argmin, automask) must not
  // come out differently in the two tasks.
  f2 hsA[9], hsB[9];
  f2 hyA[2], hyB[2];             // same for the target's (r,g): sum y, sum y^2
  float hzA[2], hzB[2];          // and its b channel
  f2 hcA[GRAD ? 9 : 1], hcB[GRAD ? 9 : 1];  // same for the 18 partial planes: output row c-1, c
  WarpRow w1;                    // row r-1 (raw values, for the L1 term of the centre row)
  PixInfo pi1;                   // row c-1 = r-2 (decided one iteration ago)
  f2 gP[POSE ? 12 : 1];          // d loss / d P of both frames
  float acc_rw = 0.f, acc_w = 0.f, acc_cons = 0.f, acc_dist = 0.f;  // per-lane partials (<= rows terms each)
#pragma unroll
  for (int i = 0; i < 9; ++i) { hsA[i] = bc(0.f); hsB[i] = bc(0.f); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { hyA[i] = bc(0.f); hyB[i] = bc(0.f); hzA[i] = 0.f; hzB[i] = 0.f; }
  if (GRAD)
#pragma unroll
    for (int i = 0; i < 9; ++i) { hcA[i] = bc(0.f); hcB[i] = bc(0.f); }
  if (POSE)
#pragma unroll
    for (int i = 0; i < 12; ++i) gP[i] = bc(0.f);
  pi1.rp = 0.f; pi1.w = 0.f; pi1.win = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) w1.x[k] = bc(0.f);
  w1.yrg = bc(0.f); w1.yb = 0.f;

  // rows r = y_lo-HALO .. y_hi-1+HALO are warped (reflected when outside the image; row -2 is never
  // needed); the statistics row is c = r-1, the gradient row q = r-2
  // what the gradient row (two iterations behind the warped row) needs of its pixel -- the warped
  // values, the target and the chain-rule numbers -- waits in a 3-slot per-lane LDS ring (no bank
  // conflicts: lane-private dwords; no barrier: one wavefront) instead of ~50 registers or a re-warp
  constexpr int RING = GRAD ? (POSE ? MAL_RING_POSE : 12) : 1;  // x, then (du, dv, u/v/rz) or e; the target row is re-read from memory
  __shared__ float s_ring[GRAD ? 3 : 1][RING][64];
  // Horizontal 3-sums of the gradient passes go through LDS instead of DPP (round 4; -DMAL_HSUM_DPP keeps the written-out
  // v_add_f32_dpp blocks of round 3 for A/B).  scripts/dpp_probe.hip: at two waves per SIMD EVERY cross-lane VALU form
  // (wave_shr, row_shr, quad_perm, v_mov_dpp, v_fmac_dpp) costs a wave 9.2 cycles of its issue timeline against 5.0 for a
  // plain instruction, so a 3-sum by two DPP adds costs 18.4 -- and the same sum as one LDS write, two LDS reads and two
  // plain adds 9.1 (profiles/r04_dpp_probe.txt).  A lane stores four values (16 bytes, lane stride 16: conflict-free
  // ds_write_b128), reads its two neighbours' four (two ds_read_b128) and adds with packed instructions; 42 sums per row are
  // 11 writes + 22 reads + 42 packed adds instead of 84 DPP adds + 18 packed multiplies (the adjoint's border weights ride
  // on the receiving side as fused multiply-adds: x 1 or x 2 is exact, so the rounding is that of the DPP form, bit for
  // bit).  LDS operations of a wave execute in order, so ONE 1 KB buffer serves all groups without waits in between.
  // Lanes 0 and 63 take their own value for the missing neighbour (DPP's bound_ctrl gave 0): those are halo lanes whose sums
  // feed only halo lanes, never an output pixel.
  // Measured (same box, teacher pass replayed alone, profiles/r04_hsum_variants_ab.txt): LDS for the 18 partial planes -1 %,
  // LDS for the 24 statistic planes +2 % (their results are needed at once: the LDS round trip is exposed where the DPP adds
  // pipeline), both +2.5 %, and the gathers'-shadow ordering +7 % -- although the row loop's vector-ALU pipe cycles fall by
  // 13 %: the pass is not bound by the vector ALU's throughput (LABBOOK.md 6).  Shipped: partial planes through LDS, the rest DPP.
  // Build-time switches (same-box A/B with scripts/build_variant.py):
  //   MAL_HSUM_H   the 24 statistic planes:  0 = DPP blocks, 1 = LDS, three groups in flight, 2 = LDS, all groups in flight
  //   MAL_HSUM_HC  the 18 partial planes:    0 = DPP blocks, 1 = LDS, three groups in flight, 2 = LDS, all groups in flight
  //   MAL_HSUM_FWD the forward-only passes:  0 = DPP (the compiler's peephole), 1 = LDS
  //   MAL_SHADOW   1 = target sums / L1 term / pose re-derivation pinned between the gathers and the blend
#ifndef MAL_HSUM_H
#define MAL_HSUM_H 0
#endif
#ifndef MAL_HSUM_HC
#define MAL_HSUM_HC 2
#endif
#ifndef MAL_HSUM_FWD
#define MAL_HSUM_FWD 0
#endif
#ifndef MAL_SHADOW
#define MAL_SHADOW 0
#endif
  constexpr int H_MODE = GRAD ? MAL_HSUM_H : (MAL_HSUM_FWD ? 2 : 0), HC_MODE = MAL_HSUM_HC;
  constexpr bool HSUM_LDS = H_MODE != 0 || (GRAD && HC_MODE != 0);
  __shared__ f4 s_stage[HSUM_LDS ? 64 : 1];
  const int laneL = max(lane - 1, 0), laneR = min(lane + 1, 63);
  // The exchange is a cross-lane communication through memory: lane i's store must be visible to lanes i-1 / i+1 before their
  // loads, and the loads must be done before the next group's stores reuse the cell.  The hardware gives that for free (a
  // wave's LDS operations execute in order); __builtin_amdgcn_wave_barrier() -- no instruction, a convergent scheduling
  // barrier -- makes the compiler keep it too, whatever a later version can prove about the addresses (advisor, round 4).
  auto nb4 = [&](f4 v, f4& L, f4& R) __attribute__((always_inline)) {
    s_stage[lane] = v;
    __builtin_amdgcn_wave_barrier();
    L = s_stage[laneL];
    R = s_stage[laneR];
    __builtin_amdgcn_wave_barrier();
  };
  auto nb2 = [&](f2 v, f2& L, f2& R) __attribute__((always_inline)) {  // a half group: the first 8 bytes of the lane's cell
    *reinterpret_cast<f2*>(&s_stage[lane]) = v;
    __builtin_amdgcn_wave_barrier();
    L = *reinterpret_cast<const f2*>(&s_stage[laneL]);
    R = *reinterpret_cast<const f2*>(&s_stage[laneR]);
    __builtin_amdgcn_wave_barrier();
  };
  // SHADOW (round 4): what an iteration can do WITHOUT the texels of the row being warped is placed between the issue of the
  // eight gathers and the blend that waits for them -- the one long memory latency of the iteration, of which round 3 covered
  // ~25 instructions: the horizontal sums of the target row (they need only the target texel, requested one row ahead), the
  // L1 term of the statistics row (raw values of the previous row) and, in the pose variants, the re-derivation of the point
  // and of d u / d disp, d v / d disp of the gradient row (u, v, 1/z from the ring, the disparity of two rows ago).
  constexpr bool SHADOW = MAL_SHADOW && H_MODE != 0 && GRAD;
  // ... the pose re-derivation only where its seven results fit next to the rest (the temporal and epilogue variants spill)
  constexpr bool SHADOW_POSE = SHADOW && POSE && !TEMPORAL && !EPI && !DBG;
  struct PosePrep { f2 alq, beq; float X[3]; f2 Xf0, Xf1, Xf2; };  // Xf*: the point per frame (FRAMED; scalars: an array member went to scratch)
  int it = 0;
  // One-row halo (p.bnd != nullptr, gradient passes of the whole-step list): a task warps ONE row beyond each end of its
  // segment and evaluates the statistics / decisions of its OWN rows only.  The gradient of a boundary row then lacks
  // the window row that belongs to the neighbouring task -- but that neighbour holds everything this row's pixel
  // contributes to the missing term (its warped values and chain-rule numbers: the neighbour's halo row, in its ring),
  // so it evaluates the term itself, one gradient row beyond each end of its segment, and leaves it in the boundary
  // scratch rows p.bnd ([b][segment][first/last row][W]); pose terms go into its own per-task partials.  The assembly
  // kernel adds the scratch rows to the map in fixed order.  rows+2 warped rows instead of rows+4, no statistics row is
  // evaluated twice (so no decision can come out differently in two tasks), rows+2 gradient rows instead of rows.
  // h1e = 1 with the one-row halo (else 0): warped rows [y_lo-HALO+h1e, y_hi-1+HALO-h1e], statistics rows
  // [y_lo-(HALO-1)+h1e, y_hi-1+(HALO-1)-h1e], gradient rows [y_lo-h1e, y_hi-1+h1e] inside the image
  const int h1e = (GRAD && p.bnd != nullptr) ? 1 : 0;
  const int r_first = max(y_lo - HALO + h1e, -1), r_warp_last = y_hi - 1 + HALO - h1e;
  // A wave runs its stages in order and the vector-memory counter retires in order, so a load that is
  // waited for soon after it was issued exposes a full memory round trip (two waves per SIMD hide little),
  // and so does any wait that follows younger loads.  Hence EVERY per-pixel operand except the gathers is
  // requested one iteration ahead (struct Ahead): the only wait for them sits at the top of the next
  // iteration, when they have had a whole iteration to arrive.  The gathers are issued as soon as the
  // projection allows; the work that does not need them sits between their issue and the blend.
  struct Ahead {
    float disp, disp2, y[3];                      // the row to warp
    float ident, noise, ext, mono, cost;          // statistics row c (TEMPORAL: ident = forced weight, noise = forced winner)
    float e_mono, e_mr, e_er, e_ensd;             // epilogue row (e_ensd: the learnt ensemble's disparity, --learn_ens)
    float gc[TEMPORAL ? 6 : 1];                   // TEMPORAL: d loss / d warped colour that arrives through syn, gradient row
  };
  // the map pointers of the parameter block, read together (one scalar-load wait per iteration)
  struct Maps {
    const float *ident, *noise, *ext_mask, *lowest_cost, *mono_disp, *mono_depth, *mono_reproj, *ens_reproj, *target; int packed;
    const float* ens_disp;                                                           // EPI, --learn_ens
    const float* forced_w; const unsigned char* forced_arg; const float* gcol[2];  // TEMPORAL
    const float* fin_gn;                                                             // TEMPORAL
  };
  auto maps_of = [&](CParams& pp) {
    Maps m;
    m.ident = AUTOMASK ? pp.ident : nullptr; m.noise = (AUTOMASK && !NO_NOISE) ? pp.noise : nullptr;
    m.ext_mask = EXT_NO ? nullptr : pp.ext_mask; m.lowest_cost = COST_NO ? nullptr : pp.lowest_cost;
    m.mono_disp = (COST_NO && !EPI) ? nullptr : pp.mono_disp;
    m.mono_depth = EPI ? pp.mono_depth : nullptr; m.mono_reproj = EPI ? pp.mono_reproj : nullptr;
    m.ens_reproj = EPI ? pp.ens_reproj : nullptr; m.target = pp.target; m.packed = LEAN ? 3 : pp.packed;
    m.forced_w = TEMPORAL ? pp.forced_w : nullptr; m.forced_arg = TEMPORAL ? pp.forced_arg : nullptr;
    m.gcol[0] = TEMPORAL ? pp.g_color[0] : nullptr; m.gcol[1] = TEMPORAL ? pp.g_color[1] : nullptr;
    m.fin_gn = TEMPORAL ? pp.fin_gn : nullptr;
    m.ens_disp = EPI ? pp.ens_disp : nullptr;
    return m;
  };
  auto row_of = [&](int rr) { return min(max(reflect1(rr, H), 0), H - 1); };
  // everything iteration `rr` will need
  auto request = [&](const Maps& pp, int rr, Ahead& a) {
    // No branches here: a fixed number of loads lets the compiler count exactly how many are younger than the
    // gathers (the blend then waits for the gathers only).  A map that is absent is read from the disparity map
    // of the sample instead (a valid address, L2-resident) and the value is dropped.
    const bool packed_t = (pp.packed & 2) != 0;
    const unsigned pix = (unsigned)(prow(row_of(rr)) * W + gxr);
    const unsigned od = moff(row_of(rr));
    a.disp = ldf(disp_b, od);
    a.disp2 = 0.f;
    if (!NO_DISP2) {
      const float v = ldf(disp2_b ? disp2_b : disp_b, od);
      a.disp2 = disp2_b ? v : 0.f;
    }
    load_rgb(pp.target, packed_t, b, HW, pix, a.y);  // one texel (packed) or three planes
    const unsigned oc = moff(min(max(rr - 1, 0), H - 1));  // statistics row c = rr-1
    auto opt = [&](const float* m, unsigned off, float absent) {
      const float v = ldf(m ? m : disp_b, off);
      return m ? v : absent;
    };
    a.ident = 0.f; a.noise = 0.f;
    if (AUTOMASK && !TEMPORAL) { a.ident = ldf(pp.ident, oc); if (!NO_NOISE) a.noise = opt(pp.noise, oc, 0.f); }
    if (TEMPORAL) {
      a.ident = ldf(pp.forced_w, oc);
      a.noise = (float)*(pp.forced_arg + (oc >> 2));
      // (kernel-argument pointer) + (32-bit offset holding sample and channel: a (B,3,H,W) image is below 2^32 bytes) -- with
      // a pointer per plane the compiler, out of scalar registers, formed six 64-bit vector addresses per row and spilled
      const unsigned og = moff(min(max(rr - 2, 0), H - 1)) + 2u * (unsigned)b * (unsigned)HW * 4u;  // pixel (b, 0, y, x) of a (B,3,H,W) image
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const unsigned o = og + (unsigned)ch * (unsigned)HW * 4u;
#pragma unroll
        for (int f = 0; f < 2; ++f) a.gc[f * 3 + ch] = ldf(pp.gcol[f], o);
      }
    }
    a.ext = 1.f; a.mono = 0.f; a.cost = 1.f;
    if (EXT_YES) a.ext = ldf(pp.ext_mask, oc);
    else if (!EXT_NO) a.ext = opt(pp.ext_mask, oc, 1.f);
    if (COST_YES) {
      a.mono = ldf(pp.mono_disp, oc); a.cost = ldf(pp.lowest_cost, oc);
    } else if (!COST_NO) {
      a.mono = opt(pp.lowest_cost ? pp.mono_disp : nullptr, oc, 0.f);
      a.cost = opt(pp.lowest_cost, oc, 1.f);
    }
    const unsigned oq = GRAD ? moff(min(max(rr - 2, 0), H - 1)) : oc;  // epilogue row
    a.e_mono = 0.f; a.e_mr = 0.f; a.e_er = 0.f; a.e_ensd = 0.f;
    if (EPI) {
      a.e_mono = ldf(MONO_YES ? pp.mono_disp : (pp.mono_disp ? pp.mono_disp : pp.mono_depth), oq);
      a.e_mr = ldf(pp.mono_reproj, oq);
      a.e_er = opt(pp.ens_reproj, oq, 0.f);
      a.e_ensd = opt(pp.ens_disp, oq, 0.f);
    }
    if (TEMPORAL) a.e_mono = opt(pp.fin_gn, oq, 0.f);  // the smoothness gradient of the gradient row (fin_out)
  };
  Ahead nxt;
  request(maps_of(p), r_first, nxt);
  float fin_cR = 0.f, fin_cS = 0.f, fin_inv = 0.f, fin_corr = 0.f;
  if (TEMPORAL && p.fin_out) {
    const float gg = p.fin_g_total ? *p.fin_g_total : 1.0f;
    fin_cR = p.fin_coefs[0] * gg; fin_cS = p.fin_coefs[4] * gg;
    fin_inv = div_(1.0f, (float)p.fin_stats[b] + 1e-7f);
    fin_corr = (float)p.fin_stats[2 * p.B + b];
#if MAL_FIN_UNIFORM
    // wave-uniform (the sample's scalars): keep them in scalar registers -- this instantiation sits at the 256-VGPR limit
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    fin_cR = uni(fin_cR); fin_cS = uni(fin_cS); fin_inv = uni(fin_inv); fin_corr = uni(fin_corr);
#endif
  }
  // reciprocals of the grid normalisation's divisors, once per wave (scalar registers)
  const float norm_rw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(
      int, refined_rcp(p.convention == 0 ? (float)(W - 1) : (float)W))));
  const float norm_rh = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(
      int, refined_rcp(p.convention == 0 ? (float)(H - 1) : (float)H))));
  // experiments (mal_set_option("debug", 64)): cycles per stage of this wave, written over min_reproj[task*8..]
#ifdef MAL_STAGE_TIMERS  // build with -DMAL_STAGE_TIMERS for scripts/gpu_stage_times.py
  unsigned tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = clock64();
#endif
  f2 y2rg = bc(0.f);
  float y2b = 0.f;  // target pixel of row r-2
  float dv_1 = 0.f, dv_2 = 0.f;  // disparity of rows r-1, r-2 (the pose terms of the gradient row re-project it)
  float dw_1 = 0.f, dw_2 = 0.f;  // FRAMED: frame 1's
  // ================= epilogue terms of output row (EPI): row q = r-2 with GRAD, c without ===
  auto epilogue = [&](CParams& p, int r, const PixInfo& pq, float le_disp, float le_mono, float le_mr, float le_er,
                      unsigned so_c, unsigned so_q, bool has_mdisp, bool has_er, float le_ensd) __attribute__((always_inline)) {
    const int q = GRAD ? r - 2 : r - 1;
    if (q >= y_lo && q < y_hi && out_x) {
      const unsigned go = GRAD ? so_q : so_c;
      const float dm = depth_of(le_disp, p.min_disp, p.range);
      const float ddepth = -(dm * dm) * p.range;
      const float dmono = has_mdisp ? depth_of(le_mono, p.min_disp, p.range) : le_mono;
      const float m = pq.w, cm = 1.0f - m, mm = 1.0f - cm;
      const float dc = dm - dmono;
      acc_cons += fabsf(dc) * cm;
      int idx = 0;
      float best = le_mr;
      if (has_er) {
        const float r_ens = le_er;
        if (r_ens < best) { best = r_ens; idx = 1; }
      }
      if (pq.rp < best) idx = 2;
      const bool learnt = p.ens_disp != nullptr;  // --learn_ens: the ensemble's depth is a map of its own
      float dens = 0.f, ens = (dmono + dm) / 2.0f;
      if (learnt) { dens = depth_of(le_ensd, p.min_disp, p.range); ens = dens; }  // wave-uniform
      const float target = idx == 0 ? dmono : (idx == 2 ? dm : ens);
      const float dd = target - dm;
      acc_dist += fabsf(dd) * mm;
      if (DBG) dec_store(p.dbg, (unsigned)(p.B * HW), MAL_DEC_DISTIL, go, (unsigned)idx);
      if (p.cons_target) stf(p.cons_target, go, div_(1.0f, dmono * cm + dm * (1.0f - cm)));
      if (GRAD) {
        const float gc = sgnf(dc) * cm * ddepth;
        const float gd = sgnf(dd) * (idx == 0 ? -1.0f : (idx == 2 ? 0.0f : (learnt ? -1.0f : -0.5f))) * mm * ddepth;
        if (learnt && p.g_ens)  // d |ens - dm| / d ens_disp where the ensemble won, with the distillation term's loss weight
          stf(p.g_ens, go, idx == 1 ? p.merge_distil * (sgnf(dd) * mm * (-(dens * dens) * p.range)) : 0.0f);
        if (p.dual_distil && p.g_ens)  // --dual_distil: d |mono - dm| / d mono_disp where the teacher won (wave-uniform flag)
          stf(p.g_ens, go, (idx == 0 && has_mdisp) ? p.merge_distil * (sgnf(dd) * mm * (-(dmono * dmono) * p.range)) : 0.0f);
        if (p.g_distil) {
          stf(p.g_cons, go, gc);
          stf(p.g_distil, go, gd);
        } else {
          stf(p.g_cons, go, fma_(p.merge_cons, gc, p.merge_distil * gd));
        }
      }
    }
  };
  // ================= the gradient row q = r-2 of iteration r (hc: the horizontal sums of row c = r-1's partial planes)
  // d u / d disp, d v / d disp and the point of row q (pose variants): u, v, 1/z come back from the ring; the point and the
  // depth derivatives are re-derived
  auto pose_prep = [&](int q, int it, PosePrep& o) __attribute__((always_inline)) {
    float (*slot)[64] = s_ring[(it + 1) % 3];
    const f2 pq_u = (f2){slot[MAL_RI(18)][lane], slot[MAL_RI(19)][lane]}, pq_v = (f2){slot[MAL_RI(20)][lane], slot[MAL_RI(21)][lane]};
    const f2 pq_rz = (f2){slot[MAL_RI(22)][lane], slot[MAL_RI(23)][lane]};
    const float depth = depth_of(dv_2, p.min_disp, p.range);
    const float ddepth = -(depth * depth) * p.range;
    float ray[3], ik[9];
    f2 P[12];
    load_cam(cam_b, P, ik);
    ray_of(ik, (float)gxr, (float)prow(q), ray);
    o.X[0] = depth * ray[0]; o.X[1] = depth * ray[1]; o.X[2] = depth * ray[2];
    const f2 c0 = P[0] * bc(ray[0]) + P[1] * bc(ray[1]) + P[2] * bc(ray[2]);
    const f2 c1 = P[4] * bc(ray[0]) + P[5] * bc(ray[1]) + P[6] * bc(ray[2]);
    const f2 c2 = P[8] * bc(ray[0]) + P[9] * bc(ray[1]) + P[10] * bc(ray[2]);
    f2 dd = bc(ddepth);
    if (FRAMED) {  // frame 1's point and depth derivative come from its own disparity
      const float depth1 = depth_of(dw_2, p.min_disp, p.range);
      dd = (f2){ddepth, -(depth1 * depth1) * p.range};
      o.Xf0 = (f2){o.X[0], depth1 * ray[0]}; o.Xf1 = (f2){o.X[1], depth1 * ray[1]}; o.Xf2 = (f2){o.X[2], depth1 * ray[2]};
    }
    o.alq = (c0 - pq_u * c2) * pq_rz * dd;  // d u / d disp (the clip gate is inside du, dv)
    o.beq = (c1 - pq_v * c2) * pq_rz * dd;
  };
  auto gradient_row = [&](CParams& p, int r, int it, const f2* hc, const Ahead& cur, unsigned so_q,
                          float le_mono, const PosePrep* prep) __attribute__((always_inline)) {
  // ================= stage G: output row q = c-1 = r-2 ======================================
  const int q = r - 2, c = r - 1;
  const bool own_q = q >= y_lo && q < y_hi;  // false: a boundary row of the neighbouring task (one-row halo)
  if (q >= y_lo - h1e && q < y_hi + h1e && (unsigned)q < (unsigned)H && (LEAN || !(p.debug & 4))) {  // wave-uniform
    WarpRow wq;
    DerivRow dq;
    f2 pq_u = bc(0.f), pq_v = bc(0.f), pq_rz = bc(0.f);
    {
      float (*slot)[64] = s_ring[(it + 1) % 3];  // written two iterations ago
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        wq.x[k] = (f2){slot[MAL_RI(2 * k)][lane], slot[MAL_RI(2 * k + 1)][lane]};
        if (POSE) {
          dq.du[k] = (f2){slot[MAL_RI(6 + 2 * k)][lane], slot[MAL_RI(6 + 2 * k + 1)][lane]};
          dq.dv[k] = (f2){slot[MAL_RI(12 + 2 * k)][lane], slot[MAL_RI(12 + 2 * k + 1)][lane]};
        } else {
          dq.e[k] = (f2){slot[MAL_RI(6 + 2 * k)][lane], slot[MAL_RI(6 + 2 * k + 1)][lane]};
        }
      }
      if (POSE) {
        pq_u = (f2){slot[MAL_RI(18)][lane], slot[MAL_RI(19)][lane]}; pq_v = (f2){slot[MAL_RI(20)][lane], slot[MAL_RI(21)][lane]};
        pq_rz = (f2){slot[MAL_RI(22)][lane], slot[MAL_RI(23)][lane]};
      }
      wq.yrg = y2rg; wq.yb = y2b;  // target of row q = r-2, kept from its own iteration
    }
    const float wyd = (q == H - 2) ? 2.0f : 1.0f;  // row q+1 = c is the bottom border row
    // L1 term of the winner: w * 0.15/3 * sign(x - y)
    float lw = pi1.w * (0.15f / 3.0f);
    if (VARIANTS && p.no_ssim) lw = pi1.w * (1.0f / 3.0f);
    float lw0 = pi1.win == 0 ? lw : 0.f, lw1 = pi1.win != 0 ? lw : 0.f;
    if (VARIANTS && pi1.win == 2) { lw0 = 0.5f * lw; lw1 = lw0; }
    const f2 lwk[3] = {bc(lw0), bc(lw1), (f2){lw0, lw1}};
    f2 g[3];
    // only the winner's L1 term is non-zero: take the sign of the winner's differences and let the zero
    // weight of the other candidate discard it
    const bool w1_ = pi1.win != 0;
    const f2 dwin = (w1_ ? wq.x[1] : wq.x[0]) - wq.yrg;
    const float dwb = (w1_ ? wq.x[2].y : wq.x[2].x) - wq.yb;
    const f2 sgrg = (f2){sgnf(dwin.x), sgnf(dwin.y)};
    const float sgb = sgnf(dwb);
    if (DBG && out_x && own_q)
      dec_store(p.dbg, (unsigned)(p.B * HW), MAL_DEC_L1, so_q,
                (unsigned)(int)(sgrg.x + 1.0f) | ((unsigned)(int)(sgrg.y + 1.0f) << 2) | ((unsigned)(int)(sgb + 1.0f) << 4));
    f2 sgk[3] = {sgrg, sgrg, bc(sgb)};
    if (VARIANTS && pi1.win == 2) {  // both candidates carry an L1 term: each channel pair its own signs
      const f2 d0 = wq.x[0] - wq.yrg, d1 = wq.x[1] - wq.yrg, d2 = wq.x[2] - bc(wq.yb);
      sgk[0] = (f2){sgnf(d0.x), sgnf(d0.y)}; sgk[1] = (f2){sgnf(d1.x), sgnf(d1.y)}; sgk[2] = (f2){sgnf(d2.x), sgnf(d2.y)};
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f2 SA = fma2(bc(wyd), hc[k * 3], hcA[k * 3]);
      const f2 SB = fma2(bc(wyd), hc[k * 3 + 1], hcA[k * 3 + 1]);
      const f2 SC = fma2(bc(wyd), hc[k * 3 + 2], hcA[k * 3 + 2]);
      const f2 xq = wq.x[k], yq = k < 2 ? wq.yrg : bc(wq.yb);
      const f2 sg = sgk[k];
      g[k] = fma2(lwk[k], sg, fma2(SC, yq, fma2(SB, xq, SA)));
    }
    if (TEMPORAL) {  // what reaches the warped colours through the synthesised images (own rows only: x 0 on a boundary row)
      const f2 ownf = bc(own_q ? 1.0f : 0.0f);
      g[0] = fma2((f2){cur.gc[0], cur.gc[1]}, ownf, g[0]);
      g[1] = fma2((f2){cur.gc[3], cur.gc[4]}, ownf, g[1]);
      g[2] = fma2((f2){cur.gc[2], cur.gc[5]}, ownf, g[2]);
    }
    MAL_MARK(61);  // ring read, L1 term, vertical adjoint sums, d loss / d warped colour done
    float gdisp, gdisp1 = 0.f;
    if (POSE) {
      PosePrep here;
      if (!prep) pose_prep(q, it, here);  // (the drain iterations; the row loop hands in what it formed in the gathers' shadow)
      const PosePrep& pp = prep ? *prep : here;
      const f2 alq = pp.alq, beq = pp.beq;
      const float (&X)[3] = pp.X;
      const f2 tu0 = g[0] * dq.du[0], tu1 = g[1] * dq.du[1], tu2 = g[2] * dq.du[2];
      const f2 tv0 = g[0] * dq.dv[0], tv1 = g[1] * dq.dv[1], tv2 = g[2] * dq.dv[2];
      const f2 gu = (f2){(tu0.x + tu0.y) + tu2.x, (tu1.x + tu1.y) + tu2.y};
      const f2 gv = (f2){(tv0.x + tv0.y) + tv2.x, (tv1.x + tv1.y) + tv2.y};
      const f2 gd = gu * alq + gv * beq;
      gdisp = gd.x + gd.y;
      if (FRAMED) { gdisp = gd.x; gdisp1 = gd.y; }  // each frame's candidate has its own disparity map
      MAL_MARK(62);  // chain rule to the disparity done; pose partials follow
      if (out_x) {
        const f2 a0 = gu * pq_rz, a1 = gv * pq_rz, a2 = -(gu * pq_u + gv * pq_v) * pq_rz;
        const f2 a[3] = {a0, a1, a2};
        const f2 Xj[3] = {FRAMED ? pp.Xf0 : bc(X[0]), FRAMED ? pp.Xf1 : bc(X[1]), FRAMED ? pp.Xf2 : bc(X[2])};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) gP[i * 4 + j] = fma2(a[i], Xj[j], gP[i * 4 + j]);
          gP[i * 4 + 3] += a[i];
        }
      }
    } else {
      const f2 t0 = g[0] * dq.e[0], t1 = g[1] * dq.e[1], t2 = g[2] * dq.e[2];
      const f2 ts = (t0 + t1) + t2;
      gdisp = ts.x + ts.y;
    }
    if (out_x) {
      if (!own_q) {  // the neighbour's boundary row: this task's window row of its gradient, into the scratch rows
        const int py = prow(q), sq = py / p.rows;
        const unsigned which = (py == sq * p.rows) ? 0u : 1u;
        stf(p.bnd, ((unsigned)((b * p.segs + sq) * 2) + which) * (unsigned)W * 4u + (unsigned)gxr * 4u, gdisp);
        if (FRAMED) stf(p.bnd2, ((unsigned)((b * p.segs + sq) * 2) + which) * (unsigned)W * 4u + (unsigned)gxr * 4u, gdisp1);
      } else if (TEMPORAL && p.fin_out) stf(p.fin_out, so_q, fma_(fin_cR, gdisp, fin_cS * (le_mono * fin_inv - fin_corr)));
      else {
        stf(p.g_reproj, so_q, gdisp);
        if (FRAMED) stf(p.g_reproj2, so_q, gdisp1);
      }
    }
  }
  MAL_MARK(63);  // gradient row stored
  // roll the partial-plane sums: (row c: top+mid) <- (row c: top) + hc(c) ; (row c+1: top) <- hc(c)
  {
    const float wyu = (c == 0) ? 2.0f : 1.0f;  // hc(c) as the TOP neighbour of row c+1: doubled if c is row 0
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      hcA[i] = hcB[i] + hc[i];
      hcB[i] = bc(wyu) * hc[i];
    }
  }
  };
  for (int r = r_first; r <= r_warp_last; ++r, ++it) {
    CParams* kp = kp0;
    asm volatile("" : "+s"(kp));
    CParams& p = *kp;
    PixInfo pi0;  // what stage S decides for the centre row c = r-1
    pi0.rp = 0.f; pi0.w = 0.f; pi0.win = 0;
    const Maps mp = maps_of(p);
    const bool has_noise = mp.noise != nullptr, has_ext = EXT_YES || mp.ext_mask != nullptr,
               has_cost = COST_YES || mp.lowest_cost != nullptr, has_mdisp = MONO_YES || mp.mono_disp != nullptr,
               has_er = mp.ens_reproj != nullptr;
    WarpConsts wc;
    wc.src[0] = p.src[0]; wc.src[1] = p.src[1]; wc.packed = mp.packed; wc.debug = LEAN ? 0 : p.debug; wc.W = W; wc.H = H;
    wc.convention = CONV >= 0 ? CONV : p.convention; wc.min_disp = p.min_disp; wc.range = p.range; wc.eps = p.eps;
    wc.rw = norm_rw; wc.rh = norm_rh;
    wc.dbg = DBG ? p.dbg : nullptr; wc.dbg_n = (unsigned)(p.B * HW);
    wc.dbg_off = 0; wc.dbg_on = false;
    if (DBG) { wc.dbg_on = r >= y_lo && r < y_hi && out_x; wc.dbg_off = moff(row_of(r)); }
#ifdef MAL_STAGE_TIMERS
    auto tick = [&](int i) {
      if (wc.debug & 64) { const unsigned long long t = clock64(); tacc[i] += (unsigned)(t - tprev); tprev = t; }
    };
#else
    auto tick = [](int) {};
#endif
    tick(0); MAL_MARK(0);  // loop overhead + parameter loads
    // byte offsets (sample base included) of this lane's pixel in the rows this iteration writes: c = r-1 and q = r-2
    const unsigned so_c = moff(min(max(r - 1, 0), H - 1)), so_q = GRAD ? moff(min(max(r - 2, 0), H - 1)) : so_c;
    // ---- what the previous iteration requested for this one
    const Ahead cur = nxt;
    const float ld_ident = cur.ident, ld_noise = cur.noise, ld_ext = cur.ext, ld_mono = cur.mono, ld_cost = cur.cost;
    const float le_disp = GRAD ? dv_2 : dv_1;  // the epilogue row's own disparity (rows r-2 / r-1 of this sweep)
    const float le_mono = cur.e_mono, le_mr = cur.e_mr, le_er = cur.e_er;
    // ================= stage W: warp row r (reflected if outside the image) ==================
    const int gyr = prow(row_of(r));  // physical row: addresses and the ray
    WarpRow w0;
    const float dv_ = (!FRAMED && disp2_b) ? (cur.disp + cur.disp2) / 2.0f : cur.disp;
    const float dw_ = FRAMED ? cur.disp2 : 0.f;
    w0.yrg = (f2){cur.y[0], cur.y[1]}; w0.yb = cur.y[2];
    // the issue phase (projection, operand requests, gathers) is what the rest of the iteration waits for:
    // run it at raised wave priority so the sibling wave's arithmetic does not delay it (measured -1.7 %)
    __builtin_amdgcn_s_setprio(3);
    PendingWarp pw;
    {
      f2 P[12];
      float ik[9];
      load_cam(cam_b, P, ik);
      // the next iteration's operands go out between the projection and the gathers (measured: behind the
      // gathers is 5 % slower even when the blend then waits for the gathers only)
#ifdef MAL_TAPS_INT  // A/B: integer tap offsets (round 3) in the specialised passes too
      warp_issue<GRAD, POSE, DBG, false>(wc, P, ik, b, gyr, gxr, dv_, pw, [&]() { request(mp, r + 1, nxt); });
#else
      warp_issue<GRAD, POSE, DBG, LEAN && !DBG, FRAMED>(wc, P, ik, b, gyr, gxr, dv_, pw, [&]() { request(mp, r + 1, nxt); }, dw_);
#endif
    }
    __builtin_amdgcn_s_setprio(0);
    tick(1); MAL_MARK(1);  // small loads, prefetch, projection, gathers issued
    // ---- in the shadow of the gathers (see SHADOW above)
    f2 sh_hy[2] = {bc(0.f), bc(0.f)}, sh_l15 = bc(0.f);
    float sh_hz[2] = {0.f, 0.f};
    PosePrep prep;
    if (SHADOW) {
      f4 L4, R4;
      f2 L2, R2;
      const f2 yy = w0.yrg * w0.yrg;
      const float yb2 = w0.yb * w0.yb;
      nb4((f4){w0.yrg.x, w0.yrg.y, yy.x, yy.y}, L4, R4);
      nb2((f2){w0.yb, yb2}, L2, R2);
      if (SHADOW_POSE) pose_prep(r - 2, it, prep);
      // L1 term of the statistics row c = r-1 (its raw values are the previous iteration's)
      const f2 l0 = w1.yrg - w1.x[0], l1 = w1.yrg - w1.x[1], l2 = bc(w1.yb) - w1.x[2];
      const f2 lsum = (f2){(fabsf(l0.x) + fabsf(l0.y)) + fabsf(l2.x), (fabsf(l1.x) + fabsf(l1.y)) + fabsf(l2.y)};
      sh_l15 = bc(0.15f) * div3_2(lsum);
      sh_hy[0] = ((f2){L4.x, L4.y} + w0.yrg) + (f2){R4.x, R4.y};
      sh_hy[1] = ((f2){L4.z, L4.w} + yy) + (f2){R4.z, R4.w};
      sh_hz[0] = (L2.x + w0.yb) + R2.x;
      sh_hz[1] = (L2.y + yb2) + R2.y;
      // keep it there: the scheduler prices a gather like any load and pulls the blend (and its wait) up in front of this
      // block.  Two ordered (volatile) empty asm statements pin the order by data flow: this one consumes what the block
      // computed, the ones that open the blend (warp_finish) produce the texels it reads.
      asm volatile("" : "+v"(sh_hy[0]), "+v"(sh_hy[1]), "+v"(sh_hz[0]), "+v"(sh_hz[1]), "+v"(sh_l15));
      if (SHADOW_POSE)
        asm volatile("" : "+v"(prep.alq), "+v"(prep.beq), "+v"(prep.X[0]), "+v"(prep.X[1]), "+v"(prep.X[2]));
    }
    auto finish_warp = [&]() {
      DerivRow d0;
      warp_finish<GRAD, POSE>(pw, w0.x, d0);
      if (GRAD && !(wc.debug & 2)) {
        float (*slot)[64] = s_ring[it % 3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          slot[MAL_RI(2 * k)][lane] = w0.x[k].x; slot[MAL_RI(2 * k + 1)][lane] = w0.x[k].y;
          if (POSE) {
            slot[MAL_RI(6 + 2 * k)][lane] = d0.du[k].x; slot[MAL_RI(6 + 2 * k + 1)][lane] = d0.du[k].y;
            slot[MAL_RI(12 + 2 * k)][lane] = d0.dv[k].x; slot[MAL_RI(12 + 2 * k + 1)][lane] = d0.dv[k].y;
          } else {
            slot[MAL_RI(6 + 2 * k)][lane] = d0.e[k].x; slot[MAL_RI(6 + 2 * k + 1)][lane] = d0.e[k].y;
          }
        }
        if (POSE) {
          slot[MAL_RI(18)][lane] = pw.u.x; slot[MAL_RI(19)][lane] = pw.u.y; slot[MAL_RI(20)][lane] = pw.v.x; slot[MAL_RI(21)][lane] = pw.v.y;
          slot[MAL_RI(22)][lane] = pw.rz.x; slot[MAL_RI(23)][lane] = pw.rz.y;
        }
      }
    };

    if (EPI && GRAD) epilogue(p, r, pi1, le_disp, le_mono, le_mr, le_er, so_c, so_q, has_mdisp, has_er, cur.e_ensd);
    tick(2); MAL_MARK(2);  // epilogue terms
    finish_warp();
    tick(3); MAL_MARK(3);  // gather wait, blend, ring write
    if (OUTS) {  // the pass in front of the temporal-hint producer: the warped images, planar
      float* const c0 = p.color_out[0];
      float* const c1 = p.color_out[1];
      if (c0 && r >= y_lo && r < y_hi && out_x) {
        // (kernel-argument pointer) + (32-bit offset holding sample and channel plane): a pointer per plane costs the loop six
        // 64-bit vector address computations (the scalar registers are exhausted); march_launch bounds the buffers below 2^32 bytes
        const unsigned og = moff(r) - (unsigned)b * (unsigned)HW * 4u;  // pixel offset inside a plane
        const unsigned hw4 = (unsigned)HW * 4u;
        const unsigned o0 = og + (unsigned)b * (p.color_out_stride ? (unsigned)p.color_out_stride : 3u * (unsigned)HW) * 4u;
        const unsigned o1 = o0 + hw4, o2 = o1 + hw4;
        stf(c0, o0, w0.x[0].x); stf(c0, o1, w0.x[0].y); stf(c0, o2, w0.x[2].x);
        stf(c1, o0, w0.x[1].x); stf(c1, o1, w0.x[1].y); stf(c1, o2, w0.x[2].y);
        if (p.color_out2[0]) {
          float* const d0 = p.color_out2[0];
          float* const d1 = p.color_out2[1];
          const unsigned q0 = og + 3u * (unsigned)b * hw4, q1 = q0 + hw4, q2 = q1 + hw4;
          stf(d0, q0, w0.x[0].x); stf(d0, q1, w0.x[0].y); stf(d0, q2, w0.x[2].x);
          stf(d1, q0, w0.x[1].x); stf(d1, q1, w0.x[1].y); stf(d1, q2, w0.x[2].y);
        }
      }
    }

    // ================= stage H: horizontal 3-sums of row r ====================================
    f2 h[9], hy[2];
    float hz[2];
    if (GRAD || H_MODE != 0) {  // 24 values through LDS or three written-out blocks of DPP adds (mal_pairs.h)
      f2 in[12];
      in[0] = w0.yrg; in[1] = w0.yrg * w0.yrg; in[2] = (f2){w0.yb, w0.yb * w0.yb};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f2 x = w0.x[k], y = k < 2 ? w0.yrg : bc(w0.yb);
        in[3 + k * 3 + 0] = x; in[3 + k * 3 + 1] = x * x; in[3 + k * 3 + 2] = x * y;
      }
      f2 out[12];
      if (SHADOW) {
        out[0] = sh_hy[0]; out[1] = sh_hy[1]; out[2] = (f2){sh_hz[0], sh_hz[1]};
        f4 L[4], R[4];
        f2 L2, R2;
        // the candidates' planes: x, x^2, x*y of the three colour pairs (18 values: four groups and a half)
#pragma unroll
        for (int g = 0; g < 4; ++g) nb4((f4){in[3 + 2 * g].x, in[3 + 2 * g].y, in[4 + 2 * g].x, in[4 + 2 * g].y}, L[g], R[g]);
        nb2(in[11], L2, R2);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          out[3 + 2 * g] = ((f2){L[g].x, L[g].y} + in[3 + 2 * g]) + (f2){R[g].x, R[g].y};
          out[4 + 2 * g] = ((f2){L[g].z, L[g].w} + in[4 + 2 * g]) + (f2){R[g].z, R[g].w};
        }
        out[11] = (L2 + in[11]) + R2;
      } else if (H_MODE != 0) {
        // H_MODE 1: three groups in flight at a time (24 registers of neighbours; all six at once spill the pose variants)
        constexpr int NG = H_MODE == 1 ? 3 : 6;
#pragma unroll
        for (int h0 = 0; h0 < 6; h0 += NG) {
          f4 L[NG], R[NG];
#pragma unroll
          for (int g = 0; g < NG; ++g)
            nb4((f4){in[2 * (h0 + g)].x, in[2 * (h0 + g)].y, in[2 * (h0 + g) + 1].x, in[2 * (h0 + g) + 1].y}, L[g], R[g]);
#pragma unroll
          for (int g = 0; g < NG; ++g) {  // (left + own) + right, as the DPP form
            out[2 * (h0 + g)] = ((f2){L[g].x, L[g].y} + in[2 * (h0 + g)]) + (f2){R[g].x, R[g].y};
            out[2 * (h0 + g) + 1] = ((f2){L[g].z, L[g].w} + in[2 * (h0 + g) + 1]) + (f2){R[g].z, R[g].w};
          }
          if (H_MODE == 1) __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        float v8[8], r8[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v8[2 * i] = in[g * 4 + i].x; v8[2 * i + 1] = in[g * 4 + i].y; }
        hsum3_block8(v8, r8);
#pragma unroll
        for (int i = 0; i < 4; ++i) out[g * 4 + i] = (f2){r8[2 * i], r8[2 * i + 1]};
      }
      }
      hy[0] = out[0]; hy[1] = out[1]; hz[0] = out[2].x; hz[1] = out[2].y;
#pragma unroll
      for (int i = 0; i < 9; ++i) h[i] = out[3 + i];
    } else {
      hy[0] = hsum3(w0.yrg); hy[1] = hsum3(w0.yrg * w0.yrg);
      hz[0] = hsum3(w0.yb);  hz[1] = hsum3(w0.yb * w0.yb);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f2 x = w0.x[k], y = k < 2 ? w0.yrg : bc(w0.yb);
        h[k * 3 + 0] = hsum3(x);
        h[k * 3 + 1] = hsum3(x * x);
        h[k * 3 + 2] = hsum3(x * y);
      }
    }

    tick(4); MAL_MARK(4);  // horizontal sums
    // ================= stage S: statistics of centre row c = r-1 ==============================
    const int c = r - 1;
    const bool c_valid = c >= 0 && c < H && c >= y_lo - (HALO - 1) + h1e && c <= y_hi - 1 + (HALO - 1) - h1e;
    f2 coef[GRAD ? 9 : 1];
    if (GRAD)
#pragma unroll
      for (int i = 0; i < 9; ++i) coef[i] = bc(0.f);
    if (c_valid && !(wc.debug & 8)) {  // wave-uniform
      // the target's side of the statistics, shared by the two candidates
      const f2 syq = (hyA[0] + hy[0]) + hyB[0], syyq = (hyA[1] + hy[1]) + hyB[1];
      const float syz = (hzA[0] + hz[0]) + hzB[0], syyz = (hzA[1] + hz[1]) + hzB[1];
      const f2 vyq = fma2(-syq, syq, bc(9.0f) * syyq), d1yq = fma2(syq, syq, bc(kC1s));
      const float vyz = fma_(-syz, syz, 9.0f * syyz), d1yz = fma_(syz, syz, kC1s);
      f2 v[3], vc[3], pa[GRAD ? 3 : 1], pb[GRAD ? 3 : 1], pcq[GRAD ? 3 : 1];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f2 sx = (hsA[k * 3] + h[k * 3]) + hsB[k * 3], sxx = (hsA[k * 3 + 1] + h[k * 3 + 1]) + hsB[k * 3 + 1],
                 sxy = (hsA[k * 3 + 2] + h[k * 3 + 2]) + hsB[k * 3 + 2];
        v[k] = ssim_sums2<GRAD>(sx, k < 2 ? syq : bc(syz), sxx, k < 2 ? vyq : bc(vyz), k < 2 ? d1yq : bc(d1yz), sxy,
                                &pa[GRAD ? k : 0], &pb[GRAD ? k : 0], &pcq[GRAD ? k : 0]);
        vc[k] = (f2){clamp01(v[k].x), clamp01(v[k].y)};
      }
      MAL_MARK(50);  // window sums + SSIM of six values done; L1, min, masks, coefficients follow
      const f2 ssum = (f2){(vc[0].x + vc[0].y) + vc[2].x, (vc[1].x + vc[1].y) + vc[2].y};
      f2 l15 = sh_l15, lmean = bc(0.f);
      if (!SHADOW) {
        const f2 l0 = w1.yrg - w1.x[0], l1 = w1.yrg - w1.x[1], l2 = bc(w1.yb) - w1.x[2];
        const f2 lsum = (f2){(fabsf(l0.x) + fabsf(l0.y)) + fabsf(l2.x), (fabsf(l1.x) + fabsf(l1.y)) + fabsf(l2.y)};
        lmean = div3_2(lsum);
        l15 = bc(0.15f) * lmean;
      }
      f2 rr = bc(0.85f) * div3_2(ssum) + l15;
      if (VARIANTS && p.no_ssim) rr = lmean;  // --no_ssim: r = mean_c |t - p| (not in MAL_SHADOW builds, whose shadow code forms l15 only)
      pi0.win = (rr.y < rr.x) ? 1 : 0;
      pi0.rp = pi0.win ? rr.y : rr.x;
      if (VARIANTS && p.avg) { pi0.win = 2; pi0.rp = (rr.x + rr.y) * 0.5f; }  // winner "2": both candidates, half each
      if (OUTS) {  // ... the winner among the two warped candidates, twice: the copy the fused sweep updates
        unsigned char* const am = p.argmin_out;
        if (am && out_x && c >= y_lo && c < y_hi) {
          stb(am, so_c >> 2, (unsigned char)pi0.win);
          unsigned char* const am2 = p.argmin_out2;
          if (am2) stb(am2, so_c >> 2, (unsigned char)pi0.win);
        }
      }
      float w = 1.0f;
      const unsigned go = so_c;
      if (AUTOMASK && !TEMPORAL) {
        float idn = ld_ident;
        if (has_noise) idn += ld_noise * 0.00001f;
        w = (pi0.rp <= idn) ? 1.0f : 0.0f;
      }
      if (TEMPORAL) {  // decided over all four candidates by the materialised-candidate kernels: a warped candidate
        const int arg4 = (int)ld_noise;  // carries weight only where one of them won (then it is this pass's own winner)
        if (DBG && out_x && c >= y_lo && c < y_hi)
          dec_store(wc.dbg, wc.dbg_n, MAL_DEC_WIN, go, (unsigned)(arg4 < 2 ? pi0.win : arg4) | (ld_ident != 0.f ? 4u : 0u));
        w = arg4 < 2 ? ld_ident : 0.0f;
      } else if (DBG && out_x && c >= y_lo && c < y_hi)
        dec_store(wc.dbg, wc.dbg_n, MAL_DEC_WIN, go, (unsigned)pi0.win | (w != 0.f ? 4u : 0u));
      if (has_ext) {
        float em = ld_ext;
        if (has_cost) {  // consistency_mask *= compute_matching_mask (trainer.py:592-593,1066-1076)
          const float mono = depth_of(ld_mono, wc.min_disp, wc.range);
          const float matching = div_safe_(1.0f, ld_cost);
          const bool ok = (div_safe_(matching - mono, mono) < 1.0f) && (div_safe_(mono - matching, matching) < 1.0f);
          em = ok ? em : em * 0.0f;
          if (p.cmask_out && out_x && c >= y_lo && c < y_hi) stf(p.cmask_out, go, em);
        }
        w *= em;
      }
      w *= sscale;
      if (!in_x) w = 0.f;  // not a pixel: contributes nothing (its statistics only served as halo)
      pi0.w = w;
      if (out_x && c >= y_lo && c < y_hi) {
        if (p.min_reproj) stf(p.min_reproj, go, pi0.rp);
        if (OUTS) {  // the pass in front of the producer: second copy of the min, the automask weight
          if (p.min_reproj2) stf(p.min_reproj2, go, pi0.rp);
          if (p.weight_out) stf(p.weight_out, go, w);
        }
        acc_rw += pi0.rp * w;
        acc_w += w;
      }
      if (GRAD) {
        // partials of the WINNING candidate only: w * 0.85/3 (channel mean) * -1/2 (v = (1-S)/2); the box
        // mean's 1/9 lives in the sums; torch.clamp passes gradient on [0,1] inclusive (clamped == raw)
        const float kk = -w * (0.85f / 3.0f) * 0.5f;
        float kk0 = pi0.win == 0 ? kk : 0.f, kk1 = pi0.win != 0 ? kk : 0.f;
        if (VARIANTS && pi0.win == 2) { kk0 = 0.5f * kk; kk1 = kk0; }  // --avg_reprojection: d mean / d r_f = 1/2
        if (VARIANTS && p.no_ssim) { kk0 = 0.f; kk1 = 0.f; }
        const f2 g0 = (f2){vc[0].x == v[0].x ? kk0 : 0.f, vc[0].y == v[0].y ? kk0 : 0.f};
        const f2 g1 = (f2){vc[1].x == v[1].x ? kk1 : 0.f, vc[1].y == v[1].y ? kk1 : 0.f};
        const f2 g2 = (f2){vc[2].x == v[2].x ? kk0 : 0.f, vc[2].y == v[2].y ? kk1 : 0.f};
        const f2 gk[3] = {g0, g1, g2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          coef[k * 3 + 0] = gk[k] * pa[k];
          coef[k * 3 + 1] = gk[k] * pb[k];
          coef[k * 3 + 2] = gk[k] * pcq[k];
        }
      }
    }

    tick(5); MAL_MARK(5);  // statistics, SSIM, masks
    if (GRAD) {
      // ================= stage HC: horizontal sums of the partial planes of row c ==============
      f2 hc[9];
      if (HC_MODE != 0) {
        // (w_left * left + own) + w_right * right: the border weights of the adjoint on the receiving side, fused (exact products)
        constexpr int NG = HC_MODE == 1 ? 3 : 5;
#pragma unroll
        for (int h0 = 0; h0 < 5; h0 += NG) {  // HC_MODE 1: groups 0-2, then 3-4 (the last one holds a single pair)
          f4 L[NG], R[NG];
#pragma unroll
          for (int g = 0; g < NG && h0 + g < 5; ++g) {
            const int a = 2 * (h0 + g);
            nb4(a + 1 < 9 ? (f4){coef[a].x, coef[a].y, coef[a + 1].x, coef[a + 1].y} : (f4){coef[a].x, coef[a].y, 0.f, 0.f}, L[g], R[g]);
          }
#pragma unroll
          for (int g = 0; g < NG && h0 + g < 5; ++g) {
            const int a = 2 * (h0 + g);
            hc[a] = fma2((f2){R[g].x, R[g].y}, bc(wR), fma2((f2){L[g].x, L[g].y}, bc(wL), coef[a]));
            if (a + 1 < 9) hc[a + 1] = fma2((f2){R[g].z, R[g].w}, bc(wR), fma2((f2){L[g].z, L[g].w}, bc(wL), coef[a + 1]));
          }
          if (HC_MODE == 1) __builtin_amdgcn_sched_barrier(0);
        }
      } else
#pragma unroll
      for (int g = 0; g < 3; ++g) {  // three planes (six values) per written-out block of DPP adds
        float l6[6], c6[6], t6[6], r6[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const f2 vv = coef[g * 3 + i], l = vv * bc(sL), rr = vv * bc(sR);
          l6[2 * i] = l.x; l6[2 * i + 1] = l.y; c6[2 * i] = vv.x; c6[2 * i + 1] = vv.y; t6[2 * i] = rr.x; t6[2 * i + 1] = rr.y;
        }
        hsum3_block6(l6, c6, t6, r6);
#pragma unroll
        for (int i = 0; i < 3; ++i) hc[g * 3 + i] = (f2){r6[2 * i], r6[2 * i + 1]};
      }
      MAL_MARK(60);  // horizontal sums of the 18 partial planes done
      gradient_row(p, r, it, hc, cur, so_q, le_mono, SHADOW_POSE ? &prep : nullptr);
    }

    tick(6); MAL_MARK(6);  // partial-plane sums, gradient row
    if (EPI && !GRAD) epilogue(p, r, pi0, le_disp, le_mono, le_mr, le_er, so_c, so_q, has_mdisp, has_er, cur.e_ensd);
    if (!LEAN && p.depth_out) {
      const int q = r - 1;
      if (q >= y_lo && q < y_hi && out_x)
        stf(p.depth_out, so_c, depth_of(ldf(disp_b, so_c), wc.min_disp, wc.range));
    }

    // ================= roll the row state =====================================================
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      hsA[i] = hsB[i];
      hsB[i] = h[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      hyA[i] = hyB[i]; hyB[i] = hy[i];
      hzA[i] = hzB[i]; hzB[i] = hz[i];
    }
    if (GRAD) { y2rg = w1.yrg; y2b = w1.yb; }
    w1 = w0;
    if (GRAD) pi1 = pi0;
    dv_2 = dv_1; dv_1 = dv_;
    if (FRAMED) { dw_2 = dw_1; dw_1 = dw_; }
    tick(7); MAL_MARK(7);  // rolls
  }
  // ---- one-row halo: nothing is left to warp, two gradient rows are -- the segment's last own row (its window row below
  // belongs to the neighbour: zero here) and the neighbour's first row (this task's last statistics row is its window row
  // above).  Their operands were requested by the last warp iteration (own row) or are not needed (boundary row).
  if (GRAD) {
    const int r_last = min(y_hi - 1 + h1e, H - 1) + 2;  // the last gradient row is two iterations behind
    for (int r = r_warp_last + 1; r <= r_last; ++r, ++it) {
      CParams* kp = kp0;
      asm volatile("" : "+s"(kp));
      CParams& p = *kp;
      const Ahead cur = nxt;
      const unsigned so_c = moff(min(max(r - 1, 0), H - 1)), so_q = moff(min(max(r - 2, 0), H - 1));
      if (EPI) epilogue(p, r, pi1, dv_2, cur.e_mono, cur.e_mr, cur.e_er, so_c, so_q, p.mono_disp != nullptr, p.ens_reproj != nullptr, cur.e_ensd);
      f2 hc0[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) hc0[i] = bc(0.f);
      gradient_row(p, r, it, hc0, cur, so_q, cur.e_mono, nullptr);
      y2rg = w1.yrg; y2b = w1.yb;
      pi1.rp = 0.f; pi1.w = 0.f; pi1.win = 0;
      dv_2 = dv_1;
      if (FRAMED) dw_2 = dw_1;
    }
  }
#ifdef MAL_STAGE_TIMERS
  if ((p.debug & 64) && lane < 8 && p.min_reproj) p.min_reproj[(size_t)task * 8 + lane] = (float)tacc[lane];
#endif

  // ---- per-task partials (fixed-order second stage in pass_finalize_kernel)
  const double r0 = wave_sum_d((double)acc_rw), r1 = wave_sum_d((double)acc_w);
  const double r2 = EPI ? wave_sum_d((double)acc_cons) : 0.0, r3 = EPI ? wave_sum_d((double)acc_dist) : 0.0;
  if (lane == 0) {
    double* o = p.block_sums + (size_t)task * 8;
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
  }
  if (POSE) {  // block_gP[task][f][12]
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const float v0 = wave_sum(gP[i].x), v1 = wave_sum(gP[i].y);
      if (lane == 0) { p.block_gP[(size_t)task * 24 + i] = v0; p.block_gP[(size_t)task * 24 + 12 + i] = v1; }
    }
  }
}

// the parameter block is read from the kernarg segment inside the body (see there)
template <bool GRAD, bool AUTOMASK, bool POSE, bool EPI, bool DBG = false, bool TEMPORAL = false>
__global__ __launch_bounds__(64, 2) void march_kernel(MarchParams p_kernarg) {
  march_body<GRAD, AUTOMASK, POSE, EPI, DBG, TEMPORAL, false>();
}
#ifdef MAL_EXPERIMENTS  // option "temporal_spec" (measured slower, LABBOOK.md 6): not in the default build
template <bool DBG>
__global__ __launch_bounds__(64, 2) void march_export_kernel(MarchParams p_kernarg) {
  march_body<true, true, true, false, DBG, false, true>();
}
#endif
// the teacher's gradient pass of the whole-step lists (the north-star kernel), with / without the temporal hint
#ifndef MAL_EXP_KEEP_NOISE
#define MAL_EXP_KEEP_NOISE 0
#endif
template <bool TEMPORAL, bool NOISE = false>  // NOISE: a tie-break noise map is given (else it is part of the identity map)
__global__ __launch_bounds__(64, MAL_TEACHER_WAVES) void march_teacher_kernel(MarchParams p_kernarg) {
  march_body<true, true, true, false, false, TEMPORAL, false, kSpecTeacher | (NOISE || TEMPORAL || MAL_EXP_KEEP_NOISE ? 0 : kSpecNoNoise)>();
}
// ... and the student's gradient pass with the consistency / distillation epilogue.  (The forward-only passes gain nothing
// from a specialisation -- same-box: ensemble 35.17 -> 35.12 us, the pass in front of the producer 56.8 -> 58.0 us -- and
// keep the generic instantiations.)
template <int SPEC>
__global__ __launch_bounds__(64, 2) void march_student_kernel(MarchParams p_kernarg) {
  march_body<true, false, false, true, false, false, false, SPEC>();
}
// ... and without the epilogue (the --temporal step forms the consistency / distillation terms in step_epilogue_kernel, so
// that this pass depends on nothing the teacher's side produces and runs beside the temporal hint's producer)
__global__ __launch_bounds__(64, 2) void march_student_noepi_kernel(MarchParams p_kernarg) {
  march_body<true, false, false, false, false, false, false, kSpecStudentNoEpi>();
}
// ... and the student's TEMPORAL gradient sweep of the --main_temporal step: no mask operand at all (forced_w is the whole weight)
__global__ __launch_bounds__(64, 2) void march_student_temporal_kernel(MarchParams p_kernarg) {
  march_body<true, false, false, false, false, true, false, kSpecTeacher>();
}
// DualRefine's passes of the deq iterations > 0: teacher-style pass (automask, pose gradients) with the consistency epilogue
__global__ __launch_bounds__(64, 2) void march_refine_kernel(MarchParams p_kernarg) {
  march_body<true, true, true, true, false, false, false, kSpecRefine>();
}
// DualRefine's pose-update losses (dualrefine/trainer.py:457-480,699-767): teacher-style pass whose two candidates are warped
// with two different disparities (MarchParams::framed)
template <bool AUTOMASK, bool DBG>
__global__ __launch_bounds__(64, 2) void march_framed_kernel(MarchParams p_kernarg) {
  march_body<true, AUTOMASK, true, false, DBG, false, false, kSpecFramed>();
}
// ... specialised like the other passes of the one-call lists (packed texels, no mask operands): what mal_dr_loss_fwd launches
__global__ __launch_bounds__(64, 2) void march_framed_lean_kernel(MarchParams p_kernarg) {
  march_body<true, true, true, false, false, false, false, kSpecFramed | kSpecLean | kSpecExtNo | kSpecCostNo | kSpecNoScale>();
}

#ifdef MAL_EXPERIMENTS  // option "march3" (measured slower, LABBOOK.md 6): not in the default build
#include "experiments/march3.inc"
#endif  // MAL_EXPERIMENTS

// ---- identity term + texel packing: min over the two raw sources of r(src_f, target)
// (manydepth/loss_utils.py:92-101, forward only) in the same sweep that turns the two planar (B,3,H,W)
// sources into the texels (B,H,W,kTexel) the warp gathers from.  Same marching structure and the same
// colour pairs as march_kernel: x[0] = (r,g) of source 0, x[1] = (r,g) of source 1, x[2] = (b0, b1).
struct IdentParams {
  const float* target; const float* src[2];   // planar (B,3,H,W)
  float* packed[2];                           // the sources as (B,H,W,kTexel) out, nullable
  float* packed_target;                       // the target as (B,H,W,kTexel) out, nullable
  float* ident;                               // (B,1,H,W) out
  int B, H, W, strips, segs, rows, ntasks, per_xcd;
  int pose_blocks; StepPoses sp;              // the last pose_blocks workgroups: poses + camera block of sample b
  TieNoise tn;                                // ident += 1e-5 * N(0,1) (Philox), see mal_march.h
  SmoothParams sm;                            // edge-aware smoothness of up to two disparity maps in the same sweep
  int variant;                                // VARIANTS instantiation only: bit 0 --no_ssim, bit 1 --avg_reprojection
  int dr_blocks; DrExtra dr;                  // DualRefine's one-call step: its prologue as the launch's last dr_blocks workgroups
  int ms_blocks; MsExtra ms;                  // the four-scale step: noise maps + upsampled disparities, behind those
};

// TEXIN: the three images arrive as (B,H,W,3) texels already -- a (B,3,H,W) tensor in torch.channels_last IS that layout --:
// one 12-byte load per image and pixel instead of three 4-byte ones, and no texel copy is written (p.packed* are NULL: the
// passes gather from the caller's tensors).
template <bool TEXIN, bool VARIANTS = false>
__global__ __launch_bounds__(64, 3) void pack_identity_kernel(IdentParams p) {
  constexpr int HALO = 1, CW = 62;
  const int id = blockIdx.x;
  if (id >= p.per_xcd * 8 + p.pose_blocks + p.dr_blocks) {  // the four-scale step: noise maps, upsampled disparities
    ms_extra_block(p.ms, id - p.per_xcd * 8 - p.pose_blocks - p.dr_blocks, threadIdx.x);
    return;
  }
  if (id >= p.per_xcd * 8 + p.pose_blocks) {  // DualRefine's one-call step: camera blocks and noise maps of every slot
    dr_extra_block(p.dr, id - p.per_xcd * 8 - p.pose_blocks, threadIdx.x);
    return;
  }
  if (id >= p.per_xcd * 8) {  // whole-step list: poses of both frames and the camera block of sample b
    const int b = id - p.per_xcd * 8, tid = threadIdx.x;
    if (b == 0 && tid < 3) p.sp.ticket[tid] = 0u;  // completion counter of step_final_kernel, [1..2]: counts of the fused sweep's task order
    if (tid < 2) pose_fwd_one(p.sp.pose, tid, b);
    __syncthreads();  // T of this sample, written by threads 0/1 to global memory, is visible to the block
    cam_fill(p.sp.K, p.sp.pose.T[0], p.sp.pose.T[1], p.sp.invK, p.sp.cam, b, tid);
    return;
  }
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
  const float* tb = p.target + (size_t)b * 3 * HW;
  const float* s0 = p.src[0] + (size_t)b * 3 * HW;
  const float* s1 = p.src[1] + (size_t)b * 3 * HW;
  f2 hsA[9], hsB[9], hyA[2], hyB[2];
  float hzA[2], hzB[2];
#pragma unroll
  for (int i = 0; i < 9; ++i) { hsA[i] = bc(0.f); hsB[i] = bc(0.f); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { hyA[i] = bc(0.f); hyB[i] = bc(0.f); hzA[i] = 0.f; hzB[i] = 0.f; }
  f2 x1[3] = {bc(0.f), bc(0.f), bc(0.f)}, y1rg = bc(0.f);
  float y1b = 0.f;
  auto row_of = [&](int rr) { return min(max(reflect1(rr, H), 0), H - 1); };
  float tn4[4] = {0.f, 0.f, 0.f, 0.f};  // tie-break noise of the current group of four rows of this column
  // ---- edge-aware smoothness of the step's two disparity maps (layers.py:210-223 on disp / (mean + 1e-7),
  // loss_utils.py:119-121), riding on the target rows this sweep holds anyway (this kernel is bandwidth-bound with
  // arithmetic to spare; the marching passes are not).  The per-sample 1/(mean+1e-7) is a positive constant of the
  // sample: it is factored out of |d n| and applied to the per-sample sums afterwards (step_final_kernel); per task:
  // sum |dx d| w, sum |dy d| w, sum gn*d, sum d.  Every edge is summed once: by the task that owns its upper / left pixel.
  const int nsm = p.sm.n;
  const float sm_nx = 1.0f / ((float)p.B * (float)H * (float)(W - 1)), sm_ny = 1.0f / ((float)p.B * (float)(H - 1) * (float)W);
  float sm_prev[2] = {0.f, 0.f}, sm_g1[2] = {0.f, 0.f};
  float acc_sx[2] = {0.f, 0.f}, acc_sy[2] = {0.f, 0.f}, acc_sd[2] = {0.f, 0.f}, acc_d[2] = {0.f, 0.f};
  const unsigned sm_lane = (unsigned)(b * HW + gxr) * 4u;
  // no branches around loads (see march_kernel's request()): an absent map is read from the target image instead (a
  // valid address) and the value dropped, so the number of loads per iteration is fixed and the waits can be counted
  auto sm_request = [&](int rr, float (&dst)[2]) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      // (the value of an absent map is never used: no select here, which would wait for the load at once)
      dst[m] = ldf(m < nsm ? p.sm.disp[m] : p.target, (unsigned)(row_of(rr) * W) * 4u + sm_lane);
    }
  };
  const unsigned long long tn_step = p.tn.on ? (p.tn.counter ? *p.tn.counter : p.tn.step) : 0ull;
  // the nine planes of a pixel, requested one iteration ahead (see march_kernel)
  struct Px { float t[3], a[3], c[3]; };
  auto request = [&](int rr, Px& q) {
    const unsigned bo = (unsigned)(row_of(rr) * W + gxr) * 4u;
    if (TEXIN) {
      const texel_t t = ldt(tb, bo * (unsigned)kTexel), a = ldt(s0, bo * (unsigned)kTexel), c = ldt(s1, bo * (unsigned)kTexel);
      q.t[0] = t.x; q.t[1] = t.y; q.t[2] = t.z; q.a[0] = a.x; q.a[1] = a.y; q.a[2] = a.z; q.c[0] = c.x; q.c[1] = c.y; q.c[2] = c.z;
      return;
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      q.t[ch] = ldf(tb + (size_t)ch * HW, bo); q.a[ch] = ldf(s0 + (size_t)ch * HW, bo); q.c[ch] = ldf(s1 + (size_t)ch * HW, bo);
    }
  };
  // The operands of a row are requested TWO iterations ahead into two alternating buffers (the loop body is unrolled
  // by two so that no register copy ties a row to the loads issued one iteration ago): with one row ahead every
  // iteration waited for its predecessor's loads AND stores -- the stores sit in divergent blocks, so the compiler
  // cannot count them and falls back to vmcnt(0) -- and the sweep ran latency-bound.
  const int r_first = max(y_lo - 1, -1);
  Px bufA, bufB;
  float smA[2], smB[2];
  request(r_first, bufA);
  sm_request(r_first, smA);
  request(r_first + 1, bufB);
  sm_request(r_first + 1, smB);
  auto body = [&](const int r, Px& buf, float (&smb)[2]) {
    const Px cur = buf;
    const float sm_cur[2] = {smb[0], smb[1]};
    if (r >= y_lo && r < y_hi && out_x && p.packed[0]) {  // an image row owned by this task: emit its texels
      const size_t bo = (size_t)(r * W + gxr);
      st_texel(p.packed[0] + (size_t)b * HW * kTexel, bo, make_texel(cur.a[0], cur.a[1], cur.a[2]));
      st_texel(p.packed[1] + (size_t)b * HW * kTexel, bo, make_texel(cur.c[0], cur.c[1], cur.c[2]));
      if (p.packed_target)
        st_texel(p.packed_target + (size_t)b * HW * kTexel, bo, make_texel(cur.t[0], cur.t[1], cur.t[2]));
    }
    const f2 x0[3] = {(f2){cur.a[0], cur.a[1]}, (f2){cur.c[0], cur.c[1]}, (f2){cur.a[2], cur.c[2]}};
    const f2 y0rg = (f2){cur.t[0], cur.t[1]};
    const float y0b = cur.t[2];
    if (nsm > 0) {  // wave-uniform
      const bool row_ok = r >= 0 && r < H;
      // edge weights from the target: right edge (r,x)-(r,x+1) and up edge (r-1,x)-(r,x)
      const float eR = (fabsf(y0rg.x - dpp_shl1(y0rg.x)) + fabsf(y0rg.y - dpp_shl1(y0rg.y))) + fabsf(y0b - dpp_shl1(y0b));
      const bool hx = row_ok && in_x && gx + 1 < W;
      const float wxr = hx ? __expf(-(eR * (1.0f / 3.0f))) : 0.f;
      const bool vy = row_ok && r >= 1 && in_x;
      const float eU = (fabsf(y1rg.x - y0rg.x) + fabsf(y1rg.y - y0rg.y)) + fabsf(y1b - y0b);
      const float wyu = vy ? __expf(-(eU * (1.0f / 3.0f))) : 0.f;
      const int qs = r - 1;  // row finished now: its down edge is this up edge
      const bool own_q = qs >= y_lo && qs < y_hi && out_x, own_r = r >= y_lo && r < y_hi && out_x;
      const unsigned oq = (unsigned)(max(qs, 0) * W) * 4u + sm_lane, orow = (unsigned)(row_of(r) * W) * 4u + sm_lane;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (m >= nsm) break;
        const float n0 = sm_cur[m];
        const float dfx = n0 - dpp_shl1(n0), dfy = sm_prev[m] - n0;
        const float sx = sgnf(dfx) * wxr * sm_nx, sy = sgnf(dfy) * wyu * sm_ny;
        if (own_q) {
          const float g = sm_g1[m] + sy;
          stf(p.sm.gn[m], oq, g);
          acc_sd[m] += g * sm_prev[m];
          acc_sy[m] += fabsf(dfy) * wyu;
        }
        if (own_r) { acc_sx[m] += fabsf(dfx) * wxr; acc_d[m] += n0; }
        if (p.sm.dec[m]) {  // parity instrumentation: the signs taken (MAL_DEC_SMOOTH_X / _Y planes)
          if (own_r) dec_store(p.sm.dec[m], (unsigned)(p.B * HW), MAL_DEC_SMOOTH_X, orow, (unsigned)(int)(sgnf(dfx) + 1.0f));
          if (own_q && vy) dec_store(p.sm.dec[m], (unsigned)(p.B * HW), MAL_DEC_SMOOTH_Y, oq, (unsigned)(int)(sgnf(dfy) + 1.0f));
        }
        sm_g1[m] = (sx - dpp_shr1(sx)) - sy;
        sm_prev[m] = n0;
      }
    }
    f2 h[9], hy[2];
    float hz[2];
    hy[0] = hsum3(y0rg); hy[1] = hsum3(y0rg * y0rg);
    hz[0] = hsum3(y0b);  hz[1] = hsum3(y0b * y0b);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f2 x = x0[k], y = k < 2 ? y0rg : bc(y0b);
      h[k * 3 + 0] = hsum3(x);
      h[k * 3 + 1] = hsum3(x * x);
      h[k * 3 + 2] = hsum3(x * y);
    }
    const int c = r - 1;
    if (c >= y_lo && c < y_hi) {
      const f2 syq = (hyA[0] + hy[0]) + hyB[0], syyq = (hyA[1] + hy[1]) + hyB[1];  // outer rows first, as in march_kernel
      const float syz = (hzA[0] + hz[0]) + hzB[0], syyz = (hzA[1] + hz[1]) + hzB[1];
      const f2 vyq = fma2(-syq, syq, bc(9.0f) * syyq), d1yq = fma2(syq, syq, bc(kC1s));
      const float vyz = fma_(-syz, syz, 9.0f * syyz), d1yz = fma_(syz, syz, kC1s);
      f2 vc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f2 sx = (hsA[k * 3] + h[k * 3]) + hsB[k * 3], sxx = (hsA[k * 3 + 1] + h[k * 3 + 1]) + hsB[k * 3 + 1],
                 sxy = (hsA[k * 3 + 2] + h[k * 3 + 2]) + hsB[k * 3 + 2];
        const f2 v = ssim_sums2<false>(sx, k < 2 ? syq : bc(syz), sxx, k < 2 ? vyq : bc(vyz), k < 2 ? d1yq : bc(d1yz), sxy,
                                       nullptr, nullptr, nullptr);
        vc[k] = (f2){clamp01(v.x), clamp01(v.y)};
      }
      const f2 ssum = (f2){(vc[0].x + vc[0].y) + vc[2].x, (vc[1].x + vc[1].y) + vc[2].y};
      const f2 l0 = y1rg - x1[0], l1 = y1rg - x1[1], l2 = bc(y1b) - x1[2];
      const f2 lsum = (f2){(fabsf(l0.x) + fabsf(l0.y)) + fabsf(l2.x), (fabsf(l1.x) + fabsf(l1.y)) + fabsf(l2.y)};
      f2 rr = bc(0.85f) * div3_2(ssum) + bc(0.15f) * div3_2(lsum);
      if (VARIANTS && (p.variant & 1)) rr = div3_2(lsum);  // --no_ssim
      float idn = fminf(rr.x, rr.y);
      if (VARIANTS && (p.variant & 2)) idn = (rr.x + rr.y) * 0.5f;  // --avg_reprojection (dualrefine/trainer.py:569-570)
      if (p.tn.on) {  // wave-uniform
        if ((c & 3) == 0 || c == y_lo) tie_noise4(p.tn.seed, tn_step, (unsigned)(b * HW + (c & ~3) * W + gxr), tn4);
        const int j = c & 3;
        const float nz = j == 0 ? tn4[0] : (j == 1 ? tn4[1] : (j == 2 ? tn4[2] : tn4[3]));
        idn += nz * 0.00001f;
        if (p.tn.noise_out && out_x) stf(p.tn.noise_out + (size_t)b * HW, (unsigned)(c * W + gxr) * 4u, nz);
      }
      if (out_x) stf(p.ident + (size_t)b * HW, (unsigned)(c * W + gxr) * 4u, idn);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) { hsA[i] = hsB[i]; hsB[i] = h[i]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { hyA[i] = hyB[i]; hyB[i] = hy[i]; hzA[i] = hzB[i]; hzB[i] = hz[i]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) x1[k] = x0[k];
    y1rg = y0rg; y1b = y0b;
    // refill this buffer for row r+2 LAST, behind a scheduling barrier: its registers are free by now, so the loads
    // land in the registers the loop carries (requested earlier, they would be copied at the back edge -- and a copy of a
    // register that is being loaded waits for the load)
    __builtin_amdgcn_sched_barrier(0);
    request(r + 2, buf);
    sm_request(r + 2, smb);
  };
  // rows r_first .. y_hi in pairs; an odd count runs one row past y_hi (every store is guarded by the row range, loads
  // are clamped to the image)
  for (int r = r_first; r <= y_hi; r += 2) {
    body(r, bufA, smA);
    body(r + 1, bufB, smB);
  }
  if (nsm > 0) {  // per-task partials [task][map][4], fixed-order second stage in step_final_kernel
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const double q0 = wave_sum_d((double)acc_sx[m]), q1 = wave_sum_d((double)acc_sy[m]);
      const double q2 = wave_sum_d((double)acc_sd[m]), q3 = wave_sum_d((double)acc_d[m]);
      if (lane == 0) { double* o = p.sm.partials + (size_t)task * 8 + m * 4; o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3; }
    }
  }
}

opt_t g_pass_impl{1};   // 1 = marching (this file, fastest); 0 = LDS-tiled v1 (mal_pass.hip); 2 = LDS-tiled, 512 threads (mal_tile2.hip)
extern opt_t g_costvol_impl;  // mal_costvol.hip
opt_t g_march_flip{1};  // odd segments bottom-up (mal_set_option("march_flip", 0|1))
opt_t g_march_rows{0};  // output rows per wave task; 0 = pick so that one round of tasks fills the chip
opt_t g_pack_rows{10};  // rows per task of the identity / packing sweep: 20 segments x 11 strips x 12 samples = 2640 tasks <= 3072 (three waves per SIMD); same-box steps, round 3: 10: 0.2010 / 0.3270 ms (--distil / headline), 9: 0.2019 / 0.3290, 11: 0.2043 / 0.3291, 12: 0.329 (headline)
opt_t g_march_rows_fwd{0};  // the same for the forward-only passes (<= 168 VGPRs: three waves per SIMD); 0 = automatic
opt_t g_debug{0};
std::atomic<unsigned*> g_dec_next{nullptr};  // mal_decisions_next_pass (one-shot, taken with an atomic exchange): decision planes for the next instrumentable gradient pass
extern opt_t g_photo_impl;  // mal_photo_march.hip
extern opt_t g_epi_bwd_planes;  // mal_epipolar.hip
extern opt_t g_epi_probe;       // mal_epipolar.hip
extern opt_t g_syn_rows;        // mal_photo_march.hip
extern opt_t g_syn_queue;       // mal_photo_march.hip
extern opt_t g_step_overlap;    // mal_step.hip
extern opt_t g_student_overlap; // mal_step.hip
extern opt_t g_tail_overlap;    // mal_step.hip
extern opt_t g_sweeps_batched;  // mal_step.hip
extern opt_t g_ms_fold;         // mal_step_ms.hip
extern opt_t g_dyn_small_blocks; // mal_dyn.hip
extern opt_t g_side_priority;   // mal_step.hip
extern opt_t g_side_order;      // mal_step.hip
extern opt_t g_march_halo1;     // mal_step.hip
extern opt_t g_temporal_spec;   // mal_step.hip
opt_t g_march_lean{1};         // option "march_lean": the teacher's passes without the optional operands' code (0: generic, A/B)
opt_t g_march3{0};             // option "march3": the teacher's gradient pass as a three-wave pipeline (0: one wave per strip)

MarchParams march_params(int B, int H, int W, float min_depth, float max_depth, float eps, int convention) {
  MarchParams p = {};
  p.B = B; p.H = H; p.W = W;
  p.min_disp = (float)(1.0 / (double)max_depth);
  p.range = (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth);
  p.eps = eps; p.convention = convention;
  return p;
}

// the task decomposition of a pass over (B,H,W) with `flags`: strips x segs tasks per sample, `rows` output rows each
static void march_decompose(MarchParams& p, int flags) {
  const bool grad = flags & MAL_F_GRAD;
  const int cw = grad ? 60 : 62;
  p.strips = (p.W + cw - 1) / cw;
  int rows = grad ? g_march_rows : (g_march_rows_fwd > 0 ? g_march_rows_fwd : g_march_rows);
  if (rows <= 0) {
    // Every task is one wavefront that lives for (rows + 2*halo) iterations and the kernel runs at two
    // waves per SIMD (<= 256 VGPRs): the shortest makespan is the smallest `rows` whose task count still
    // fits in ONE resident round (CUs x 4 SIMDs x 2).  Measured on MI355X at B=12 192x640: rows 13
    // (1980 tasks <= 2048) 124 us vs rows 16 135 us vs rows 12 (2112 tasks, two rounds) 164 us.
    static int slots = 0;
    if (slots == 0) {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      (void)hipGetLastError();
      if (cus <= 0) cus = 256;
      slots = cus * 8;
    }
    rows = 8;
    while (rows < p.H && (long long)p.B * p.strips * ((p.H + rows - 1) / rows) > slots) ++rows;
  }
  if (rows < 8) rows = 8;  // workspace is sized for 8-row segments
  p.rows = rows;
  p.segs = (p.H + rows - 1) / rows;
  p.ntasks = p.B * p.strips * p.segs;
  p.per_xcd = (p.ntasks + 7) / 8;
}

void march_geometry(int B, int H, int W, int flags, int* strips, int* segs, int* rows) {
  MarchParams p = {};
  p.B = B; p.H = H; p.W = W;
  march_decompose(p, flags);
  if (strips) *strips = p.strips;
  if (segs) *segs = p.segs;
  if (rows) *rows = p.rows;
}

int march_launch(MarchParams& p, int flags, hipStream_t st) {
  const bool grad = flags & MAL_F_GRAD, automask = flags & MAL_F_AUTOMASK, pose = flags & MAL_F_POSE_GRAD,
             epi = flags & MAL_F_EPILOGUE;
  march_decompose(p, flags);
  p.debug = g_debug;
  p.flip_odd = g_march_flip;
  p.packed = ((flags & MAL_F_SRC_PACKED) ? 1 : 0) | ((flags & MAL_F_TGT_PACKED) ? 2 : 0);
  dim3 grid(p.per_xcd * 8), block(64);
  if (!p.cam) return MAL_EINVAL;
  if ((long long)p.H * p.W >= (1ll << 24)) return MAL_ESHAPE;  // tap offsets use 24-bit multiplies
  if (p.color_out[0] && (long long)p.B * (p.color_out_stride ? (long long)p.color_out_stride : 3ll * p.H * p.W) * 4 >= (1ll << 32))
    return MAL_ESHAPE;  // the exported planes are addressed with 32-bit byte offsets
  if (p.forced_w && (long long)p.B * 3 * p.H * p.W * 4 >= (1ll << 32)) return MAL_ESHAPE;  // ... and so are the (B,3,H,W) colour cotangents
  if (!p.cam_ready) {
    hipLaunchKernelGGL(cam_setup_kernel, dim3(p.B), dim3(64), 0, st, p.K, p.T[0], p.T[1], p.invK, p.cam);
    p.cam_ready = 1;
  }
  // mal_profile_next_pass times the next TEACHER gradient sweep (GRAD + AUTOMASK + POSE, no epilogue: the north-star
  // kernel, with or without the temporal hint), not whichever marching launch happens to come first
  const bool named = grad && pose && automask && !epi;
  hipEvent_t ev0 = named ? g_prof_start.exchange(nullptr) : nullptr, ev1 = named ? g_prof_stop.exchange(nullptr) : nullptr;
  if (ev0) (void)hipEventRecord(ev0, st);
  // the specialisation without the optional operands (march_body LEAN)
  if ((p.avg || p.no_ssim) && (!grad || p.forced_w || p.color_out[0])) return MAL_EINVAL;  // generic gradient passes only
  const bool lean0 = g_march_lean && p.packed == 3 && !p.depth_out && p.debug == 0 && !p.dbg && !p.avg && !p.no_ssim &&
                     (long long)p.H * p.W * (kTexel * 4) < (1ll << 24);  // tap byte offsets are formed in fp32 there
  const bool no_maps = !p.ext_mask && !p.lowest_cost && !p.sample_scale;
  const bool conv_a = p.convention == 0, conv_b = p.convention == 1;
#ifdef MAL_CONV_FIXED
  const bool conv_teacher = conv_a;  // (A/B build: the convention is compiled into the specialisations)
#else
  const bool conv_teacher = conv_a || conv_b;  // the specialised teacher pass reads the convention per launch: DualRefine's iteration 0 too
#endif
  const bool lean = lean0 && conv_teacher && no_maps && !p.disp2;                                                // kSpecTeacher
  const bool lean_student = lean0 && conv_a && !p.disp2 && p.ext_mask && p.lowest_cost && p.mono_disp;          // kSpecStudent
  const bool lean_student_nc = lean0 && conv_a && !p.disp2 && p.ext_mask && !p.lowest_cost && p.mono_disp;      // kSpecStudentNoCost
  const bool lean_refine = lean0 && conv_b && !p.disp2 && !p.lowest_cost && p.mono_disp && !p.sample_scale;     // kSpecRefine
#define MAL_LAUNCH(G, A, P, E) hipLaunchKernelGGL((march_kernel<G, A, P, E>), grid, block, 0, st, p)
#ifdef MAL_EXPERIMENTS
  if (grad && p.color_out[0]) {  // the teacher's pass of the --temporal step in front of the producer
    if (!(pose && automask && !epi) || p.forced_w || !p.color_out[1]) return MAL_EINVAL;
    if (p.dbg) {
      if (p.H >= 4096 || p.W >= 4096) return MAL_EINVAL;
      hipLaunchKernelGGL(march_export_kernel<true>, grid, block, 0, st, p);
    } else hipLaunchKernelGGL(march_export_kernel<false>, grid, block, 0, st, p);
  } else if (g_march3 && grad && pose && automask && !epi && p.bnd && p.packed == 3 && !p.forced_w && !p.disp2 && !p.ext_mask &&
             !p.sample_scale && !p.lowest_cost && !p.depth_out && p.ident && p.g_reproj && p.block_gP) {
    // the teacher's gradient pass of the whole-step lists: three-wave pipeline per strip
    if (p.dbg) {
      if (p.H >= 4096 || p.W >= 4096) return MAL_EINVAL;
      hipLaunchKernelGGL(march3_kernel<true>, grid, dim3(192), 0, st, p);
    } else hipLaunchKernelGGL(march3_kernel<false>, grid, dim3(192), 0, st, p);
  } else
#else
  if (grad && p.color_out[0]) return MAL_EINVAL;  // an exporting gradient pass exists in -DMAL_EXPERIMENTS builds only
#endif
  if (p.framed) {  // a disparity per frame: gradient + pose-gradient pass without epilogue, two gradient maps
    if (!(grad && pose && !epi) || !p.disp2 || !p.g_reproj || !p.g_reproj2 || (p.bnd != nullptr) != (p.bnd2 != nullptr) ||
        p.forced_w || p.color_out[0] || p.lowest_cost || p.depth_out)
      return MAL_EINVAL;
    if (p.dbg && (p.H >= 4096 || p.W >= 4096)) return MAL_EINVAL;
    const bool framed_lean = g_march_lean && p.packed == 3 && p.debug == 0 && !p.avg && !p.no_ssim && !p.ext_mask && !p.sample_scale &&
                             (long long)p.H * p.W * (kTexel * 4) < (1ll << 24);
    if (automask) {
      if (p.dbg) hipLaunchKernelGGL((march_framed_kernel<true, true>), grid, block, 0, st, p);
      else if (framed_lean) hipLaunchKernelGGL(march_framed_lean_kernel, grid, block, 0, st, p);
      else hipLaunchKernelGGL((march_framed_kernel<true, false>), grid, block, 0, st, p);
    } else {
      if (p.dbg) hipLaunchKernelGGL((march_framed_kernel<false, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((march_framed_kernel<false, false>), grid, block, 0, st, p);
    }
  } else if (p.forced_w && grad && !pose && !automask && !epi) {
    // TEMPORAL student pass (--main_temporal): forced_w is the pass's whole weight (consistency x matching x (1 - augmentation),
    // as the forward pass in front of the producer formed it), so no mask operand is read again
    if (!p.forced_arg || !p.g_color[0] || !p.g_color[1] || p.ext_mask || p.lowest_cost || p.sample_scale) return MAL_EINVAL;
    if (p.dbg) {
      if (p.H >= 4096 || p.W >= 4096) return MAL_EINVAL;
      hipLaunchKernelGGL((march_kernel<true, false, false, false, true, true>), grid, block, 0, st, p);
    } else if (lean0 && conv_a && !p.disp2) hipLaunchKernelGGL(march_student_temporal_kernel, grid, block, 0, st, p);
    else hipLaunchKernelGGL((march_kernel<true, false, false, false, false, true>), grid, block, 0, st, p);
  } else if (p.forced_w) {  // TEMPORAL teacher pass
    if (!(grad && pose && automask && !epi) || !p.forced_arg || !p.g_color[0] || !p.g_color[1]) return MAL_EINVAL;
    if (lean && !p.dbg) hipLaunchKernelGGL(march_teacher_kernel<true>, grid, block, 0, st, p);
    else if (p.dbg) {
      if (p.H >= 4096 || p.W >= 4096) return MAL_EINVAL;
      hipLaunchKernelGGL((march_kernel<true, true, true, false, true, true>), grid, block, 0, st, p);
    } else hipLaunchKernelGGL((march_kernel<true, true, true, false, false, true>), grid, block, 0, st, p);
  } else if (lean && !p.dbg && grad && pose && automask && !epi) {  // the teacher's pass of the whole-step lists
    if (p.noise) hipLaunchKernelGGL((march_teacher_kernel<false, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((march_teacher_kernel<false, false>), grid, block, 0, st, p);
  } else if (p.dbg) {  // instrumented instantiations exist for the two gradient passes of the whole-step list
    if (p.H >= 4096 || p.W >= 4096) return MAL_EINVAL;
    if (grad && pose && automask && !epi) hipLaunchKernelGGL((march_kernel<true, true, true, false, true>), grid, block, 0, st, p);
    else if (grad && pose && automask && epi) hipLaunchKernelGGL((march_kernel<true, true, true, true, true>), grid, block, 0, st, p);  // DualRefine, deq iterations > 0
    else if (grad && pose && !automask && !epi) hipLaunchKernelGGL((march_kernel<true, false, true, false, true>), grid, block, 0, st, p);  // DualRefine, --disable_automasking
    else if (grad && pose && !automask && epi) hipLaunchKernelGGL((march_kernel<true, false, true, true, true>), grid, block, 0, st, p);
    else if (grad && !pose && !automask && epi) hipLaunchKernelGGL((march_kernel<true, false, false, true, true>), grid, block, 0, st, p);
    else if (grad && !pose && !automask && !epi) hipLaunchKernelGGL((march_kernel<true, false, false, false, true>), grid, block, 0, st, p);
    else return MAL_EINVAL;
  } else if (lean_student && grad && !pose && !automask && epi) {
    hipLaunchKernelGGL(march_student_kernel<kSpecStudent>, grid, block, 0, st, p);
  } else if (lean0 && conv_a && !p.disp2 && p.ext_mask && p.lowest_cost && p.mono_disp && grad && !pose && !automask && !epi) {
    hipLaunchKernelGGL(march_student_noepi_kernel, grid, block, 0, st, p);
  } else if (lean_student_nc && grad && !pose && !automask && epi) {
    hipLaunchKernelGGL(march_student_kernel<kSpecStudentNoCost>, grid, block, 0, st, p);
  } else if (lean_refine && grad && pose && automask && epi && !p.forced_w) {
    hipLaunchKernelGGL(march_refine_kernel, grid, block, 0, st, p);
  } else if (!grad) {
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

int pack_identity_tasks_per_sample(int H, int W) { return ((W + 61) / 62) * ((H + g_pack_rows - 1) / g_pack_rows); }

int pack_identity_launch(const float* target, const float* src0, const float* src1, int B, int H, int W,
                         float* packed0, float* packed1, float* packed_target, float* ident, hipStream_t st,
                         const StepPoses* poses, const TieNoise* noise, const SmoothParams* smooth, int* tasks_per_sample,
                         bool texel_in, int variant, const DrExtra* dr, const MsExtra* ms) {
  IdentParams p = {};
  p.variant = variant;
  if (dr) { p.dr = *dr; p.dr_blocks = dr->slots * (dr->noise_blocks + dr->B); }
  if (ms) { p.ms = *ms; p.ms_blocks = ms->n_noise * ms->noise_blocks + ms->n_up * ms->up_blocks; }
  if (noise) p.tn = *noise;
  if (smooth) p.sm = *smooth;
  p.pose_blocks = poses ? B : 0;
  if (poses) p.sp = *poses;
  p.target = target; p.src[0] = src0; p.src[1] = src1; p.packed[0] = packed0; p.packed[1] = packed1; p.packed_target = packed_target; p.ident = ident;
  p.B = B; p.H = H; p.W = W;
  p.strips = (W + 61) / 62;
  p.rows = g_pack_rows;  // three waves per SIMD fit (launch bounds, <= 168 VGPRs with the smoothness terms)
  p.segs = (H + p.rows - 1) / p.rows;
  p.ntasks = B * p.strips * p.segs;
  p.per_xcd = (p.ntasks + 7) / 8;
  if (tasks_per_sample) *tasks_per_sample = p.strips * p.segs;
  if (variant) {  // --no_ssim / --avg_reprojection identity term: its own instantiation (the default one stays as it is)
    if (texel_in) return MAL_EINVAL;
    hipLaunchKernelGGL((pack_identity_kernel<false, true>), dim3(p.per_xcd * 8 + p.pose_blocks + p.dr_blocks + p.ms_blocks), dim3(64), 0, st, p);
  } else if (texel_in) {
    if (packed0 || packed1 || packed_target) return MAL_EINVAL;  // nothing to repack: the inputs are the texels
    hipLaunchKernelGGL(pack_identity_kernel<true>, dim3(p.per_xcd * 8 + p.pose_blocks + p.dr_blocks + p.ms_blocks), dim3(64), 0, st, p);
  } else
  hipLaunchKernelGGL(pack_identity_kernel<false>, dim3(p.per_xcd * 8 + p.pose_blocks + p.dr_blocks + p.ms_blocks), dim3(64), 0, st, p);
  return launch_status();
}

}  // namespace mal

using namespace mal;

extern "C" int mal_decisions_next_pass(uint32_t* planes) {
  g_dec_next = planes;
  return MAL_OK;
}

extern "C" int mal_march_geometry(int B, int H, int W, int flags, int* strips, int* segs, int* rows, int* iterations) {
  if (B <= 0 || H < 2 || W < 2) return MAL_ESHAPE;
  MarchParams p = {};
  p.B = B; p.H = H; p.W = W;
  march_decompose(p, flags);
  if (strips) *strips = p.strips;
  if (segs) *segs = p.segs;
  if (rows) *rows = p.rows;
  if (iterations) *iterations = p.rows + 2 * ((flags & MAL_F_GRAD) ? 2 : 1);  // row-loop iterations of a full task
  // (with the one-row halo of the whole-step list a gradient task still runs rows + 4 iterations, the last two without a warp)
  return MAL_OK;
}

static std::mutex g_option_mutex;
extern "C" int mal_build_has_experiments(void) {
#ifdef MAL_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

extern "C" int mal_set_option(const char* name, int value) {
  if (!name) return MAL_EINVAL;
  auto eq = [&](const char* s) { const char* a = name; while (*a && *a == *s) { ++a; ++s; } return *a == *s; };
  std::lock_guard<std::mutex> lock(g_option_mutex);
#ifdef MAL_EXPERIMENTS
  constexpr bool kExp = true;
#else
  constexpr bool kExp = false;  // the losing formulations are not compiled in: their switches accept the default only
#endif
  if (eq("pass_impl")) { if (value < 0 || value > 2 || (!kExp && value != 1)) return MAL_EINVAL; g_pass_impl = value; return MAL_OK; }
  if (eq("debug")) { g_debug = value; return MAL_OK; }
  if (eq("photo_impl")) { if (value < 0 || value > 1) return MAL_EINVAL; g_photo_impl = value; return MAL_OK; }
  if (eq("costvol_impl")) { if (value < 0 || value > 1) return MAL_EINVAL; g_costvol_impl = value; return MAL_OK; }
  if (eq("step_overlap")) { if (value < 0 || value > 2) return MAL_EINVAL; g_step_overlap = value; return MAL_OK; }
  if (eq("march_halo1")) { g_march_halo1 = value != 0; return MAL_OK; }
  if (eq("student_overlap")) { g_student_overlap = value != 0; return MAL_OK; }
  if (eq("ms_fold")) { g_ms_fold = value != 0; return MAL_OK; }
  if (eq("sweeps_batched")) { g_sweeps_batched = value != 0; return MAL_OK; }
  if (eq("tail_overlap")) { g_tail_overlap = value != 0; return MAL_OK; }
  if (eq("dyn_small_blocks")) { g_dyn_small_blocks = value != 0; return MAL_OK; }
  if (eq("side_order")) { g_side_order = value != 0; return MAL_OK; }
  if (eq("side_priority")) { g_side_priority = value != 0; return MAL_OK; }  // read when a side stream is first created
  if (eq("temporal_spec")) { if (!kExp && value) return MAL_EINVAL; g_temporal_spec = value != 0; return MAL_OK; }
  if (eq("march3")) { if (!kExp && value) return MAL_EINVAL; g_march3 = value != 0; return MAL_OK; }
  if (eq("march_lean")) { g_march_lean = value != 0; return MAL_OK; }
  if (eq("syn_queue")) { if (!kExp && value) return MAL_EINVAL; g_syn_queue = value != 0; return MAL_OK; }
  if (eq("syn_rows")) { if (value < 2 || value > 64) return MAL_EINVAL; g_syn_rows = value; return MAL_OK; }
  if (eq("epi_bwd_planes")) { if (value < 0 || value > 2) return MAL_EINVAL; g_epi_bwd_planes = value; return MAL_OK; }
  if (eq("epi_probe")) { g_epi_probe = value; return MAL_OK; }
  if (eq("march_flip")) { g_march_flip = value != 0; return MAL_OK; }
  if (eq("pack_rows")) { if (value < 4 || value > 4096) return MAL_EINVAL; g_pack_rows = value; return MAL_OK; }
  if (eq("march_rows_fwd")) { if (value < 0 || value > 4096) return MAL_EINVAL; g_march_rows_fwd = value; return MAL_OK; }
  if (eq("march_rows")) { if (value < 0 || value > 4096) return MAL_EINVAL; g_march_rows = value; return MAL_OK; }
  return MAL_EINVAL;
}

#ifdef MAL_EXPERIMENTS  // the LDS-tiled first formulations (mal_pass.hip, mal_tile2.hip; option "pass_impl")
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
#endif

extern "C" int mal_pass_fused(const float* disp, const float* disp2, const float* K, const float* inv_K,
                              const float* const* T, const float* const* src, const float* target,
                              const float* ident, const float* noise, const float* ext_mask,
                              const float* sample_scale, const float* mono_depth, const float* mono_reproj,
                              const float* ens_reproj, int B, int H, int W, int F, float min_depth, float max_depth,
                              float eps, int convention, int flags, float* min_reproj, double* sums, float* g_reproj,
                              float* g_cons, float* g_distil, float* const* g_T, float* consistency_target,
                              float* depth_out, void* ws, size_t ws_bytes, void* stream) {
#ifdef MAL_EXPERIMENTS
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
#endif
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
  if (grad && ((pose && automask && !epi) || (!pose && !automask && epi)))
    p.dbg = g_dec_next.exchange(nullptr);  // one-shot (tests): the instrumented instantiation of the same kernel
  // the gradient passes take the one-row halo here too: the scratch rows are folded into g_reproj by the finalize launch
  if (grad && g_march_halo1) p.bnd = w.bnd;
  rc = march_launch(p, flags, st);
  if (rc) return rc;
  BoundaryFold fold = {p.bnd, g_reproj, B, H, W, p.rows, p.segs};
  return launch_pass_finalize(w.block_sums, w.block_gP, K, p.ntasks, p.strips * p.segs, B, sums,
                              pose ? g_T[0] : nullptr, pose ? g_T[1] : nullptr, st, 4, p.bnd ? &fold : nullptr);
}

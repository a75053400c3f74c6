// Device-side helpers of the marching kernels (mal_march.hip, mal_photo_march.hip): DPP horizontal sums,
// packed fp32 pairs (two values per lane per issue), uniform-base global accesses, SSIM on window sums.
#pragma once
#include "mal_device.h"

namespace mal {

MAL_DEV float dpp_shr1(float v) {  // lane i <- lane i-1 (lane 0 <- 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
MAL_DEV float dpp_shl1(float v) {  // lane i <- lane i+1 (lane 63 <- 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
MAL_DEV float hsum3(float v) { return (dpp_shr1(v) + v) + dpp_shl1(v); }

constexpr float kC1s = 81.0f * 0.0001f;  // 81 * C1
constexpr float kC2s = 81.0f * 0.0009f;  // 81 * C2

// ---------------------------------------------------------------- packed fp32 (two values per lane)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
MAL_DEV f2 bc(float v) { return (f2){v, v}; }
MAL_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
MAL_DEV f2 rcp2(f2 a) { return (f2){__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
MAL_DEV f2 hsum3(f2 v) { return (f2){hsum3(v.x), hsum3(v.y)}; }

// ---- the same horizontal sums as ONE block of v_add_f32_dpp per group of values.  The compiler forms v_add_f32_dpp from
// update_dpp + fadd only when its DPP-combine pass finds the pair within a short scan window of the pre-RA instruction
// order; in the gradient instantiations of the marching kernel that depends on unrelated code (measured: 0 of 84 combined
// after a change elsewhere in the loop = +84 VALU instructions per row, +7 %).  Written out, the instruction count is fixed.
// Hazards (gfx9 family; the compiler's hazard pass does not look inside inline assembly): a VGPR written by a VALU
// instruction must not be read through DPP in the next two wait states, and a VALU write of EXEC (v_cmpx) needs five before
// a DPP instruction.  The block opens with s_nop 4 (five wait states: covers both whatever the compiler placed in front of
// it; +3 idle cycles per block against s_nop 1, nine blocks per row), and inside it no DPP operand is a register the block
// wrote.
#define MAL_DPP_SHR1 " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define MAL_DPP_SHL1 " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// r[i] = (shr1(v[i]) + v[i]) + shl1(v[i]), eight values
MAL_DEV void hsum3_block8(const float (&v)[8], float (&r)[8]) {
  asm("s_nop 4\n"
      "v_add_f32_dpp %0, %8, %8" MAL_DPP_SHR1 "v_add_f32_dpp %1, %9, %9" MAL_DPP_SHR1
      "v_add_f32_dpp %2, %10, %10" MAL_DPP_SHR1 "v_add_f32_dpp %3, %11, %11" MAL_DPP_SHR1
      "v_add_f32_dpp %4, %12, %12" MAL_DPP_SHR1 "v_add_f32_dpp %5, %13, %13" MAL_DPP_SHR1
      "v_add_f32_dpp %6, %14, %14" MAL_DPP_SHR1 "v_add_f32_dpp %7, %15, %15" MAL_DPP_SHR1
      "v_add_f32_dpp %0, %8, %0" MAL_DPP_SHL1 "v_add_f32_dpp %1, %9, %1" MAL_DPP_SHL1
      "v_add_f32_dpp %2, %10, %2" MAL_DPP_SHL1 "v_add_f32_dpp %3, %11, %3" MAL_DPP_SHL1
      "v_add_f32_dpp %4, %12, %4" MAL_DPP_SHL1 "v_add_f32_dpp %5, %13, %5" MAL_DPP_SHL1
      "v_add_f32_dpp %6, %14, %6" MAL_DPP_SHL1 "v_add_f32_dpp %7, %15, %7" MAL_DPP_SHL1
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
}
// r[i] = (shr1(l[i]) + c[i]) + shl1(t[i]), six values (the adjoint's border weights differ per direction)
MAL_DEV void hsum3_block6(const float (&l)[6], const float (&c)[6], const float (&t)[6], float (&r)[6]) {
  asm("s_nop 4\n"
      "v_add_f32_dpp %0, %6, %12" MAL_DPP_SHR1 "v_add_f32_dpp %1, %7, %13" MAL_DPP_SHR1
      "v_add_f32_dpp %2, %8, %14" MAL_DPP_SHR1 "v_add_f32_dpp %3, %9, %15" MAL_DPP_SHR1
      "v_add_f32_dpp %4, %10, %16" MAL_DPP_SHR1 "v_add_f32_dpp %5, %11, %17" MAL_DPP_SHR1
      "v_add_f32_dpp %0, %18, %0" MAL_DPP_SHL1 "v_add_f32_dpp %1, %19, %1" MAL_DPP_SHL1
      "v_add_f32_dpp %2, %20, %2" MAL_DPP_SHL1 "v_add_f32_dpp %3, %21, %3" MAL_DPP_SHL1
      "v_add_f32_dpp %4, %22, %4" MAL_DPP_SHL1 "v_add_f32_dpp %5, %23, %5" MAL_DPP_SHL1
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5])
      : "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3]), "v"(l[4]), "v"(l[5]),
        "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]),
        "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]));
}
// div_ / div_safe_ / div3_ of mal_device.h, element for element
MAL_DEV f2 div2_(f2 a, f2 b) {
  f2 y = rcp2(b);
  y = fma2(fma2(-b, y, bc(1.0f)), y, y);
  const f2 q = a * y;
  const f2 r = fma2(-b, q, a);
  return fma2(r, y, q);
}
MAL_DEV f2 div_safe2_(f2 a, f2 b) {
  const f2 y0 = rcp2(b);
  const f2 y = fma2(fma2(-b, y0, bc(1.0f)), y0, y0);
  const f2 q = a * y;
  const f2 r = fma2(-b, q, a);
  const f2 q1 = fma2(r, y, q);
  const f2 alt = a * y0;
  return (f2){__builtin_isfinite(q1.x) ? q1.x : alt.x, __builtin_isfinite(q1.y) ? q1.y : alt.y};
}
MAL_DEV f2 div3_2(f2 a) {
  const f2 q = a * bc(0.333333333333333333f);
  const f2 r = fma2(bc(-3.0f), q, a);
  return fma2(r, bc(0.333333333333333333f), q);
}
// Global accesses as (wave-uniform base) + (32-bit per-lane BYTE offset): the saddr form of global_load/
// global_store.  One offset register serves every map that is read at the same pixel and no 64-bit vector
// address arithmetic is needed (check_shape bounds every tensor below 2^31 bytes).
MAL_DEV float ldf(const float* base, unsigned boff) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff);
}
MAL_DEV f4 ldf4(const float* base, unsigned boff) {
  return *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(base) + boff);
}
MAL_DEV texel_t ldt(const float* base, unsigned boff) {
  return *reinterpret_cast<const texel_t*>(reinterpret_cast<const char*>(base) + boff);
}
MAL_DEV void stf(float* base, unsigned boff, float v) { *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff) = v; }
MAL_DEV void stb(unsigned char* base, unsigned boff, unsigned char v) { *(base + boff) = v; }  // scalar base + 32-bit offset
// A 32-bit byte offset as a value the compiler cannot trace back: used INSIDE a conditional block in front of ldf / stf there.
// global_load/store take (scalar base) + (32-bit vector offset) only when instruction selection sees the offset's zero-extension
// next to the access; an offset computed before a branch has its zero-extension hoisted in front of it, the block behind the
// branch then sees a 64-bit value and the access is built from a 64-bit vector add (v_lshl_add_u64 + two VGPRs) instead.
MAL_DEV unsigned fresh(unsigned v) { asm("" : "+v"(v)); return v; }
// the three colour channels of pixel `pix` of sample b: planar (B,3,H,W) or packed (B,H,W,kTexel)
MAL_DEV void load_rgb(const float* img, int packed, int b, int HW, unsigned pix, float* out) {
  if (packed) {
    const texel_t v = ldt(img + (size_t)b * HW * kTexel, pix * (unsigned)(kTexel * 4));
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
  } else {
    const float* pl = img + (size_t)b * 3 * HW;
    out[0] = ldf(pl, pix * 4u); out[1] = ldf(pl + HW, pix * 4u); out[2] = ldf(pl + 2 * (size_t)HW, pix * 4u);
  }
}
MAL_DEV float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// SSIM of two windows at once from their sums; un-clamped (1 - S)/2 and, with GRAD, the partials of S
// wrt the window sums of x: dS/d(sum x), 2 dS/d(sum x^2) (the 2 of d x^2/dx folded in), dS/d(sum xy)
template <bool GRAD>
MAL_DEV f2 ssim_sums2(f2 sx, f2 sy, f2 sxx, f2 vy /* 9 syy - sy^2 */, f2 d1y /* sy^2 + 81 C1 */, f2 sxy, f2* dsx,
                      f2* dsxx2, f2* dsxy) {
  const f2 pxy = sx * sy;
  const f2 n1 = fma2(bc(2.0f), pxy, bc(kC1s));
  const f2 n2 = fma2(bc(2.0f), fma2(bc(9.0f), sxy, -pxy), bc(kC2s));
  const f2 d1 = fma2(sx, sx, d1y);
  const f2 d2 = (fma2(-sx, sx, bc(9.0f) * sxx) + vy) + bc(kC2s);
  const f2 n = n1 * n2, d = d1 * d2;
  f2 rd = rcp2(d);
  rd = fma2(fma2(-d, rd, bc(1.0f)), rd, rd);
  const f2 S = n * rd;
  if (GRAD) {
    const f2 t1 = sy * (n2 - n1);
    const f2 t2 = (S * sx) * (d2 - d1);
    *dsx = (bc(2.0f) * (t1 - t2)) * rd;
    *dsxx2 = (bc(-18.0f) * S) * (d1 * rd);
    *dsxy = (bc(18.0f) * n1) * rd;
  }
  return fma2(S, bc(-0.5f), bc(0.5f));
}

}  // namespace mal

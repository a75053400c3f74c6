// N2: the temporal-hint producer's per-sample arithmetic -- manydepth/dyn_utils.py:6-119,
// `fill_dynamic_obj` + `generate_dynamic_instance` -- given the matched instance masks of the two warped
// frames.  (Mask2Former and the Hungarian matcher that produce the masks stay outside this library.)
//
// Upstream this is a TorchScript loop over instances with (num, 3, H, W) temporaries per call and per
// sample; here it is two launches for up to 16 samples and no temporaries (the scalar kernels first; the 16-byte /
// four-pixel forms further down are what runs for W % 16 == 0 / W % 4 == 0):
//   dyn_extents_kernel   per (instance, frame): which rows / columns >= 1 hold mask pixels (the reference
//                        weighs the mask by the row / column index, so index 0 is invisible) -> low, top,
//                        right, left per band of rows; the synthesis kernel's workgroups merge the bands and derive,
//                        per instance, the displacement: of (low_next-low_last,
//                        top_next-top_last) the one of larger magnitude (the first on a tie), halved and rounded
//                        half-to-even; columns alike; replace=1 zeroes magnitudes < 3   (dyn_utils.py:53-103)
//   dyn_synth_fwd_kernel per pixel p:
//        A(p)   = sum_i [p - d_i inside and mask_last_i(p - d_i)] img_last(p - d_i),  any(p) = or_i [...]
//        bg(p)  = or_i (mask_last_i & ~mask_next_i)(p) ? img_next(p) : img_last(p)
//        ori_last(p) = or_i (mask_last_i | mask_next_i)(p) ? (any(p) ? A(p) : bg(p)) : img_last(p)
//      and symmetrically ori_next with -d_i; the five predicates are kept as a flag byte per pixel (:6-36,106-119)
//   dyn_synth_bwd_kernel the adjoint in gather form: what p receives directly, plus, for every instance whose
//                        mask holds p, the cotangent at the pixel p was copied to.
#include "mal_common.h"
#include "mal_device.h"

namespace mal {

// row of the mask tensor that holds instance i (the torch indexing this replaces raised IndexError on a bad selection;
// a kernel cannot: the index is clamped into the tensor, so nothing outside it is ever addressed)
MAL_DEV size_t dyn_row(const long long* sel, int i, int rows) {
  long long r = sel ? sel[i] : (long long)i;
  if (rows > 0) r = r < 0 ? 0 : (r >= rows ? (long long)rows - 1 : r);
  return (size_t)r;
}

struct DynParams {
  const uint8_t* mask_last; const uint8_t* mask_next;  // (num,H,W) bytes, non-zero = set
  // instance i is row idx_*[i] of the mask tensor (the matcher's selection, dyn_utils.py:147-150, applied here instead of
  // by one gather launch per sample and frame); nullptr = row i
  const long long* idx_last; const long long* idx_next;
  int rows_last, rows_next;  // rows of the mask tensors (0 = not given): a selection outside [0, rows) is clamped, never read
  int num, C, H, W, replace;
  const float* img_last; const float* img_next;         // (C,H,W)
  float* ori_last; float* ori_next;
  int* ext;       // [num][2][kExtChunks][4]: low, top, right, left of (last, next), per band of rows
  int* delta;     // [num][2]: row, column displacement of the "last" copy
  uint8_t* flags; // [H*W]: bit0 region, bit1 any_last, bit2 any_next, bit3 last's background is img_next, bit4 next's is img_last
  const float* g_ori_last; const float* g_ori_next; float* g_img_last; float* g_img_next;
  int prefilled;                       // forward: ori_* already hold img_*: only region pixels are written
  float* g_tmp_last; float* g_tmp_next;  // backward in place (g_img_* == g_ori_*): scratch for the region pixels
};

// Up to kDynBatch samples per launch (blockIdx.z = sample): a step's temporal hint is 12 samples x 3 kernels of a few
// microseconds each when launched one by one
constexpr int kDynBatch = 16;
struct DynBatch { DynParams s[kDynBatch]; };
// the extents of one (instance, frame) mask are found by kExtChunks workgroups, one band of rows each (72 workgroups
// for 12 samples x 3 instances x 2 frames left most of the chip idle); the displacement kernel merges the bands
constexpr int kExtChunks = 8;

__global__ __launch_bounds__(1024) void dyn_extents_kernel(DynBatch bt) {
  extern __shared__ int sh[];  // row flags [H], column flags [W], then 4 results
  const DynParams& p = bt.s[blockIdx.z];
  if ((int)blockIdx.x >= p.num) return;  // the grid is as wide as the sample with the most instances
  const int i = blockIdx.x, which = blockIdx.y / kExtChunks, chunk = blockIdx.y % kExtChunks, tid = threadIdx.x;
  const int H = p.H, W = p.W;
  const int band = (H + kExtChunks - 1) / kExtChunks, r_lo = chunk * band, r_hi = min(r_lo + band, H);
  const int k_lo = r_lo * W, HW = max(r_hi, r_lo) * W;  // this workgroup scans pixels [k_lo, HW)
  int* rowf = sh;
  int* colf = sh + H;
  int* res = sh + H + W;
  for (int k = tid; k < H + W; k += 1024) sh[k] = 0;
  if (tid < 4) res[tid] = (tid & 1) ? 0x7fffffff : 0;  // low, top, right, left: max / min
  __syncthreads();
  const long long* sel = which ? p.idx_next : p.idx_last;
  const uint8_t* m = (which ? p.mask_next : p.mask_last) + dyn_row(sel, i, which ? p.rows_next : p.rows_last) * ((size_t)H * W);
  auto mark = [&](int k) { rowf[k / W] = 1; colf[k % W] = 1; };  // same-value stores: benign races
  if ((HW & 15) == 0 && (k_lo & 15) == 0 && (reinterpret_cast<size_t>(m) & 15) == 0) {
    // masks are mostly empty: scan 16 bytes per load, look at the bytes only where a word is non-zero
    const uint4* m16 = reinterpret_cast<const uint4*>(m);
    for (int k = k_lo / 16 + tid; k < HW / 16; k += 1024) {
      const uint4 v = m16[k];
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (w4[j])
          for (int t = 0; t < 4; ++t)
            if ((w4[j] >> (8 * t)) & 0xffu) mark(k * 16 + j * 4 + t);
    }
  } else {
    for (int k = k_lo + tid; k < HW; k += 1024)
      if (m[k]) mark(k);
  }
  __syncthreads();
  for (int r = 1 + tid; r < H; r += 1024)
    if (rowf[r]) { atomicMax(&res[0], r); atomicMin(&res[1], r); }
  for (int c = 1 + tid; c < W; c += 1024)
    if (colf[c]) { atomicMax(&res[2], c); atomicMin(&res[3], c); }
  __syncthreads();
  if (tid < 4) p.ext[((i * 2 + which) * kExtChunks + chunk) * 4 + tid] = res[tid];  // merged by instance_delta
}

MAL_DEV unsigned nz4(unsigned w) {  // bit k = byte k of w is non-zero
  return ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u);
}

// The same extents for W % 16 == 0 without the row / column flag arrays: a thread's 16 bytes lie in one row, so it knows
// its row and, from the non-zero bytes' bit mask, its first and last marked column; four wave-level min / max
// reductions, one LDS atomic per wave and quantity, one barrier.
__global__ __launch_bounds__(1024) void dyn_extents16_kernel(DynBatch bt) {
  __shared__ int res[4];
  const DynParams& p = bt.s[blockIdx.z];
  if ((int)blockIdx.x >= p.num) return;
  const int i = blockIdx.x, which = blockIdx.y / kExtChunks, chunk = blockIdx.y % kExtChunks, tid = threadIdx.x;
  const int H = p.H, W = p.W, W16 = W >> 4;
  const int band = (H + kExtChunks - 1) / kExtChunks, r_lo = chunk * band, r_hi = min(r_lo + band, H);
  if (tid < 4) res[tid] = (tid & 1) ? 0x7fffffff : 0;  // low, top, right, left: max / min
  __syncthreads();
  const long long* sel = which ? p.idx_next : p.idx_last;
  const uint4* m16 = reinterpret_cast<const uint4*>((which ? p.mask_next : p.mask_last) +
                                                    dyn_row(sel, i, which ? p.rows_next : p.rows_last) * ((size_t)H * W));
  int rmax = 0, rmin = 0x7fffffff, cmax = 0, cmin = 0x7fffffff;
  for (int k = r_lo * W16 + tid; k < max(r_hi, r_lo) * W16; k += (int)blockDim.x) {
    const uint4 v = m16[k];
    const unsigned w4[4] = {v.x, v.y, v.z, v.w};
    unsigned m = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) m |= nz4(w4[j]) << (4 * j);
    if (m) {
      const int row = k / W16, col0 = (k - row * W16) * 16;
      if (row >= 1) { rmax = max(rmax, row); rmin = min(rmin, row); }
      const unsigned m1 = col0 == 0 ? (m & ~1u) : m;  // column 0 is invisible, as row 0
      if (m1) { cmax = max(cmax, col0 + 31 - __clz((int)m1)); cmin = min(cmin, col0 + __ffs((int)m1) - 1); }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    rmax = max(rmax, __shfl_down(rmax, o, 64)); rmin = min(rmin, __shfl_down(rmin, o, 64));
    cmax = max(cmax, __shfl_down(cmax, o, 64)); cmin = min(cmin, __shfl_down(cmin, o, 64));
  }
  if ((tid & 63) == 0) {
    if (rmax > 0) { atomicMax(&res[0], rmax); atomicMin(&res[1], rmin); }
    if (cmax > 0) { atomicMax(&res[2], cmax); atomicMin(&res[3], cmin); }
  }
  __syncthreads();
  if (tid < 4) p.ext[((i * 2 + which) * kExtChunks + chunk) * 4 + tid] = res[tid];
}

// displacement of instance i from the band extents (dyn_utils.py:53-103): of (low_next-low_last, top_next-top_last) the one
// of larger magnitude (the first on a tie), halved and rounded half-to-even; columns alike; replace=1 zeroes |d| < 3
MAL_DEV void instance_delta(const DynParams& p, int i, int* dx_out, int* dy_out) {
  int el[4], en[4];  // low, top, right, left of (last, next): max / min over the row bands
  for (int which = 0; which < 2; ++which) {
    int* e = which ? en : el;
    for (int t = 0; t < 4; ++t) e[t] = (t & 1) ? 0x7fffffff : 0;
    for (int c = 0; c < kExtChunks; ++c)
      for (int t = 0; t < 4; ++t) {
        const int v = p.ext[((i * 2 + which) * kExtChunks + c) * 4 + t];
        e[t] = (t & 1) ? min(e[t], v) : max(e[t], v);
      }
    for (int t = 0; t < 4; ++t)
      if (e[t] == 0x7fffffff) e[t] = 0;  // nothing present: argmin over an all-"inf" row returns index 0
  }
  auto pick = [](int a, int b) {  // larger magnitude, the first on a tie; half, rounded half-to-even
    const int s = abs(b) > abs(a) ? b : a;
    return (int)rintf((float)s * 0.5f);
  };
  int dx = pick(en[0] - el[0], en[1] - el[1]);
  int dy = pick(en[2] - el[2], en[3] - el[3]);
  if (p.replace) { if (abs(dx) < 3) dx = 0; if (abs(dy) < 3) dy = 0; }
  *dx_out = dx; *dy_out = dy;
}

// the per-instance small data of a sample in shared memory: displacement [num][2], then mask rows [num][2].  Forward
// (from_ext): every workgroup derives the displacements from the band extents itself (a few hundred integers from the L2:
// no launch of its own for it) and the sample's first workgroup records them for the backward.
MAL_DEV void stage_instances(const DynParams& p, int* s_delta, bool from_ext) {
  if (from_ext) {
    for (int i = threadIdx.x; i < p.num; i += 256) {
      int dx, dy;
      instance_delta(p, i, &dx, &dy);
      s_delta[2 * i] = dx; s_delta[2 * i + 1] = dy;
      if (blockIdx.x == 0) { p.delta[2 * i] = dx; p.delta[2 * i + 1] = dy; }
    }
  }
  for (int k = threadIdx.x; k < p.num * 2; k += 256) {
    if (!from_ext) s_delta[k] = p.delta[k];
    const int i = k >> 1;
    const long long* sel = (k & 1) ? p.idx_next : p.idx_last;
    s_delta[2 * p.num + k] = (int)dyn_row(sel, i, (k & 1) ? p.rows_next : p.rows_last);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void dyn_synth_fwd_kernel(DynBatch bt) {
  extern __shared__ int s_delta[];  // [num][2] displacements, [num][2] mask rows
  const DynParams& p = bt.s[blockIdx.z];
  stage_instances(p, s_delta, true);
  const int* s_row = s_delta + 2 * p.num;
  const int H = p.H, W = p.W, HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const int r = pix / W, c = pix - r * W;
  bool region = false, any_l = false, any_n = false, bg_l = false, bg_n = false;
  float accl[4] = {0.f, 0.f, 0.f, 0.f}, accn[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < p.num; ++i) {
    const uint8_t* ml = p.mask_last + (size_t)s_row[2 * i] * HW;
    const uint8_t* mn = p.mask_next + (size_t)s_row[2 * i + 1] * HW;
    const bool a = ml[pix] != 0, b = mn[pix] != 0;
    region |= a | b; bg_l |= a & !b; bg_n |= b & !a;
    const int dx = s_delta[2 * i], dy = s_delta[2 * i + 1];
    {  // "last" moves by +d: destination p takes source p - d
      const int rs = r - dx, cs = c - dy;
      if (rs >= 0 && rs < H && cs >= 0 && cs < W && ml[rs * W + cs]) {
        any_l = true;
        for (int ch = 0; ch < p.C; ++ch) accl[ch] += p.img_last[(size_t)ch * HW + rs * W + cs];
      }
    }
    {  // "next" moves by -d
      const int rs = r + dx, cs = c + dy;
      if (rs >= 0 && rs < H && cs >= 0 && cs < W && mn[rs * W + cs]) {
        any_n = true;
        for (int ch = 0; ch < p.C; ++ch) accn[ch] += p.img_next[(size_t)ch * HW + rs * W + cs];
      }
    }
  }
  p.flags[pix] = (uint8_t)((region ? 1 : 0) | (any_l ? 2 : 0) | (any_n ? 4 : 0) | (bg_l ? 8 : 0) | (bg_n ? 16 : 0));
  if (p.prefilled && !region) return;  // ori_* hold the images already
  for (int ch = 0; ch < p.C; ++ch) {
    const float il = p.img_last[(size_t)ch * HW + pix], in = p.img_next[(size_t)ch * HW + pix];
    p.ori_last[(size_t)ch * HW + pix] = region ? (any_l ? accl[ch] : (bg_l ? in : il)) : il;
    p.ori_next[(size_t)ch * HW + pix] = region ? (any_n ? accn[ch] : (bg_n ? il : in)) : in;
  }
}

// REGION: the in-place form, first launch -- only pixels of the instances' region change (everywhere else the gradient
// IS the cotangent): their new values go to g_tmp_*, dyn_apply_region_kernel moves them once every gather has been served
template <bool REGION>
__global__ __launch_bounds__(256) void dyn_synth_bwd_kernel(DynBatch bt) {
  extern __shared__ int s_delta[];
  const DynParams& p = bt.s[blockIdx.z];
  stage_instances(p, s_delta, false);
  const int* s_row = s_delta + 2 * p.num;
  const int H = p.H, W = p.W, HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const int r = pix / W, c = pix - r * W;
  const int f = p.flags[pix];
  if (REGION && !(f & 1)) return;
  // where the two outputs at p read the two images AT p
  const bool ol_from_l = !(f & 1) || (!(f & 2) && !(f & 8)), ol_from_n = (f & 1) && !(f & 2) && (f & 8);
  const bool on_from_n = !(f & 1) || (!(f & 4) && !(f & 16)), on_from_l = (f & 1) && !(f & 4) && (f & 16);
  float gl[4] = {0.f, 0.f, 0.f, 0.f}, gn[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ch = 0; ch < p.C; ++ch) {
    const float a = p.g_ori_last[(size_t)ch * HW + pix], b = p.g_ori_next[(size_t)ch * HW + pix];
    if (ol_from_l) gl[ch] += a;
    if (on_from_l) gl[ch] += b;
    if (on_from_n) gn[ch] += b;
    if (ol_from_n) gn[ch] += a;
  }
  // where p was copied to: instance i took img_last(p) to p + d_i (if that lies in the replaced region)
  for (int i = 0; i < p.num; ++i) {
    const int dx = s_delta[2 * i], dy = s_delta[2 * i + 1];
    if (p.mask_last[(size_t)s_row[2 * i] * HW + pix]) {
      const int rd = r + dx, cd = c + dy;
      if (rd >= 0 && rd < H && cd >= 0 && cd < W && (p.flags[rd * W + cd] & 1))
        for (int ch = 0; ch < p.C; ++ch) gl[ch] += p.g_ori_last[(size_t)ch * HW + rd * W + cd];
    }
    if (p.mask_next[(size_t)s_row[2 * i + 1] * HW + pix]) {
      const int rd = r - dx, cd = c - dy;
      if (rd >= 0 && rd < H && cd >= 0 && cd < W && (p.flags[rd * W + cd] & 1))
        for (int ch = 0; ch < p.C; ++ch) gn[ch] += p.g_ori_next[(size_t)ch * HW + rd * W + cd];
    }
  }
  float* const out_l = REGION ? p.g_tmp_last : p.g_img_last;
  float* const out_n = REGION ? p.g_tmp_next : p.g_img_next;
  for (int ch = 0; ch < p.C; ++ch) {
    if (out_l) out_l[(size_t)ch * HW + pix] = gl[ch];
    if (out_n) out_n[(size_t)ch * HW + pix] = gn[ch];
  }
}

__global__ __launch_bounds__(256) void dyn_apply_region_kernel(DynBatch bt) {
  const DynParams& p = bt.s[blockIdx.z];
  const int HW = p.H * p.W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW || !(p.flags[pix] & 1)) return;
  for (int ch = 0; ch < p.C; ++ch) {
    p.g_img_last[(size_t)ch * HW + pix] = p.g_tmp_last[(size_t)ch * HW + pix];
    p.g_img_next[(size_t)ch * HW + pix] = p.g_tmp_next[(size_t)ch * HW + pix];
  }
}

// ---------------------------------------------------------------- four pixels per thread
// The kernels above are one memory round trip after another per wave (a byte per lane, the shifted byte behind a test):
// 25 us for work whose bytes need 5.  For W % 4 == 0 a thread owns four consecutive pixels of a row: mask bytes arrive
// as 32-bit words (the shifted ones as the two words that straddle them, funnel-shifted), every load of an instance is
// unconditional, the images move as float4 -- and only where no pixel of the quad lies in the instances' region.
// bytes row[cs .. cs+3] for any integer cs, bytes outside [0, W) read as zero; row 4-byte aligned, W % 4 == 0
MAL_DEV unsigned bytes4_at(const uint8_t* row, int cs, int W) {
  const unsigned* rw = reinterpret_cast<const unsigned*>(row);
  const int a0 = cs >> 2, sh = (cs & 3) * 8, wmax = (W >> 2) - 1;
  const unsigned l0 = rw[min(max(a0, 0), wmax)], l1 = rw[min(max(a0 + 1, 0), wmax)];
  const unsigned w0 = (a0 >= 0 && a0 <= wmax) ? l0 : 0u, w1 = (a0 + 1 >= 0 && a0 + 1 <= wmax) ? l1 : 0u;
  return sh ? (w0 >> sh) | (w1 << (32 - sh)) : w0;
}

// Branches around LOADS serialise a wave (one wait per basic block): the quad kernels load unconditionally from
// predicated addresses (a miss reads element 0 and contributes +0.0f, which changes no sum), behind WAVE-uniform guards
// (`__any`) so that the ~90 % of wavefronts without a pixel in any instance's region skip the work altogether.
MAL_DEV float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
MAL_DEV void st4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

template <int C>
__global__ __launch_bounds__(256) void dyn_synth_fwd4_kernel(DynBatch bt) {
  extern __shared__ int s_delta[];
  const DynParams& p = bt.s[blockIdx.z];
  stage_instances(p, s_delta, true);
  const int* s_row = s_delta + 2 * p.num;
  const int H = p.H, W = p.W, HW = H * W, W4 = W >> 2;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= (HW >> 2)) return;
  const int r = q / W4, c = (q - r * W4) * 4, pix = r * W + c;
  const float* __restrict__ il = p.img_last;
  const float* __restrict__ in = p.img_next;
  unsigned region = 0u, any_l = 0u, any_n = 0u, bg_l = 0u, bg_n = 0u;  // bit k = pixel pix + k
  float accl[4][C], accn[4][C];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int ch = 0; ch < C; ++ch) { accl[k][ch] = 0.f; accn[k][ch] = 0.f; }
  for (int i = 0; i < p.num; ++i) {
    const uint8_t* ml = p.mask_last + (size_t)s_row[2 * i] * HW;
    const uint8_t* mn = p.mask_next + (size_t)s_row[2 * i + 1] * HW;
    const int dx = s_delta[2 * i], dy = s_delta[2 * i + 1];
    const int rl = r - dx, rn = r + dx;  // source rows of the "last" copy (moves by +d) and of the "next" copy (-d)
    const unsigned wa = *reinterpret_cast<const unsigned*>(ml + pix), wb = *reinterpret_cast<const unsigned*>(mn + pix);
    const unsigned sl_raw = bytes4_at(ml + (size_t)min(max(rl, 0), H - 1) * W, c - dy, W);
    const unsigned sn_raw = bytes4_at(mn + (size_t)min(max(rn, 0), H - 1) * W, c + dy, W);
    const unsigned a4 = nz4(wa), b4 = nz4(wb);
    const unsigned sl = (rl >= 0 && rl < H) ? nz4(sl_raw) : 0u, sn = (rn >= 0 && rn < H) ? nz4(sn_raw) : 0u;
    region |= a4 | b4; bg_l |= a4 & ~b4; bg_n |= b4 & ~a4;
    any_l |= sl; any_n |= sn;
    if (__any((sl | sn) != 0u)) {
      const int bl = rl * W + (c - dy), bn = rn * W + (c + dy);
      float vl[4][C], vn[4][C];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
          vl[k][ch] = il[(size_t)ch * HW + (((sl >> k) & 1u) ? bl + k : 0)];
          vn[k][ch] = in[(size_t)ch * HW + (((sn >> k) & 1u) ? bn + k : 0)];
        }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
          accl[k][ch] += ((sl >> k) & 1u) ? vl[k][ch] : 0.f;
          accn[k][ch] += ((sn >> k) & 1u) ? vn[k][ch] : 0.f;
        }
    }
  }
  unsigned fw = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    fw |= ((((region >> k) & 1u)) | (((any_l >> k) & 1u) << 1) | (((any_n >> k) & 1u) << 2) | (((bg_l >> k) & 1u) << 3) |
           (((bg_n >> k) & 1u) << 4)) << (8 * k);
  *reinterpret_cast<unsigned*>(p.flags + pix) = fw;
  if (p.prefilled && !__any(region != 0u)) return;  // ori_* hold the images already
  float ol[C][4], on[C][4];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) {
    const float4 a = ld4(il + (size_t)ch * HW + pix), b = ld4(in + (size_t)ch * HW + pix);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool rg = (region >> k) & 1u;
      ol[ch][k] = rg ? (((any_l >> k) & 1u) ? accl[k][ch] : (((bg_l >> k) & 1u) ? bv[k] : av[k])) : av[k];
      on[ch][k] = rg ? (((any_n >> k) & 1u) ? accn[k][ch] : (((bg_n >> k) & 1u) ? av[k] : bv[k])) : bv[k];
    }
  }
  if (!p.prefilled) {
#pragma unroll
    for (int ch = 0; ch < C; ++ch) { st4(p.ori_last + (size_t)ch * HW + pix, ol[ch]); st4(p.ori_next + (size_t)ch * HW + pix, on[ch]); }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if ((region >> k) & 1u) {
#pragma unroll
      for (int ch = 0; ch < C; ++ch) { p.ori_last[(size_t)ch * HW + pix + k] = ol[ch][k]; p.ori_next[(size_t)ch * HW + pix + k] = on[ch][k]; }
    }
}

// the adjoint, same order of additions per pixel as dyn_synth_bwd_kernel
template <int C, bool REGION>
__global__ __launch_bounds__(256) void dyn_synth_bwd4_kernel(DynBatch bt) {
  extern __shared__ int s_delta[];
  const DynParams& p = bt.s[blockIdx.z];
  stage_instances(p, s_delta, false);
  const int* s_row = s_delta + 2 * p.num;
  const int H = p.H, W = p.W, HW = H * W, W4 = W >> 2;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= (HW >> 2)) return;
  const int r = q / W4, c = (q - r * W4) * 4, pix = r * W + c;
  const float* __restrict__ go_l = p.g_ori_last;
  const float* __restrict__ go_n = p.g_ori_next;
  const unsigned fw = *reinterpret_cast<const unsigned*>(p.flags + pix);
  float* const out_l = REGION ? p.g_tmp_last : p.g_img_last;
  float* const out_n = REGION ? p.g_tmp_next : p.g_img_next;
  const bool wave_region = __any((fw & 0x01010101u) != 0u);
  if (REGION && !wave_region) return;
  float gl[C][4], gn[C][4];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) {
    const float4 a4 = ld4(go_l + (size_t)ch * HW + pix), b4 = ld4(go_n + (size_t)ch * HW + pix);
    const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = (int)((fw >> (8 * k)) & 0xffu);
      const bool ol_from_l = !(f & 1) || (!(f & 2) && !(f & 8)), ol_from_n = (f & 1) && !(f & 2) && (f & 8);
      const bool on_from_n = !(f & 1) || (!(f & 4) && !(f & 16)), on_from_l = (f & 1) && !(f & 4) && (f & 16);
      float l = 0.f, n = 0.f;
      l += ol_from_l ? a[k] : 0.f;
      l += on_from_l ? b[k] : 0.f;
      n += on_from_n ? b[k] : 0.f;
      n += ol_from_n ? a[k] : 0.f;
      gl[ch][k] = l; gn[ch][k] = n;
    }
  }
  if (wave_region) {
    for (int i = 0; i < p.num; ++i) {
      const int dx = s_delta[2 * i], dy = s_delta[2 * i + 1];
      const unsigned ma = nz4(*reinterpret_cast<const unsigned*>(p.mask_last + (size_t)s_row[2 * i] * HW + pix));
      const unsigned mb = nz4(*reinterpret_cast<const unsigned*>(p.mask_next + (size_t)s_row[2 * i + 1] * HW + pix));
      // where pixel p was copied to: "last" to p + d, "next" to p - d, if that pixel lies in the replaced region
      const int rdl = r + dx, rdn = r - dx;
      const unsigned fl = bytes4_at(p.flags + (size_t)min(max(rdl, 0), H - 1) * W, c + dy, W);
      const unsigned fn = bytes4_at(p.flags + (size_t)min(max(rdn, 0), H - 1) * W, c - dy, W);
      unsigned hl = 0u, hn = 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hl |= (((fl >> (8 * k)) & 1u) & ((ma >> k) & 1u)) << k;
        hn |= (((fn >> (8 * k)) & 1u) & ((mb >> k) & 1u)) << k;
      }
      if (rdl < 0 || rdl >= H) hl = 0u;
      if (rdn < 0 || rdn >= H) hn = 0u;
      if (__any((hl | hn) != 0u)) {
        const int bl = rdl * W + (c + dy), bn = rdn * W + (c - dy);
        float vl[C][4], vn[C][4];
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            vl[ch][k] = go_l[(size_t)ch * HW + (((hl >> k) & 1u) ? bl + k : 0)];
            vn[ch][k] = go_n[(size_t)ch * HW + (((hn >> k) & 1u) ? bn + k : 0)];
          }
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            gl[ch][k] += ((hl >> k) & 1u) ? vl[ch][k] : 0.f;
            gn[ch][k] += ((hn >> k) & 1u) ? vn[ch][k] : 0.f;
          }
      }
    }
  }
  if (!REGION) {
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      if (out_l) st4(out_l + (size_t)ch * HW + pix, gl[ch]);
      if (out_n) st4(out_n + (size_t)ch * HW + pix, gn[ch]);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if ((fw >> (8 * k)) & 1u) {
#pragma unroll
      for (int ch = 0; ch < C; ++ch) { out_l[(size_t)ch * HW + pix + k] = gl[ch][k]; out_n[(size_t)ch * HW + pix + k] = gn[ch][k]; }
    }
}

template <int C>
__global__ __launch_bounds__(256) void dyn_apply_region4_kernel(DynBatch bt) {
  const DynParams& p = bt.s[blockIdx.z];
  const int HW = p.H * p.W;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= (HW >> 2)) return;
  const int pix = q * 4;
  const unsigned fw = *reinterpret_cast<const unsigned*>(p.flags + pix);
  if (!__any((fw & 0x01010101u) != 0u)) return;
  const float* __restrict__ tl = p.g_tmp_last;
  const float* __restrict__ tn = p.g_tmp_next;
  float4 vl[C], vn[C];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) { vl[ch] = ld4(tl + (size_t)ch * HW + pix); vn[ch] = ld4(tn + (size_t)ch * HW + pix); }
#pragma unroll
  for (int ch = 0; ch < C; ++ch) {
    const float a[4] = {vl[ch].x, vl[ch].y, vl[ch].z, vl[ch].w}, b[4] = {vn[ch].x, vn[ch].y, vn[ch].z, vn[ch].w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((fw >> (8 * k)) & 1u) { p.g_img_last[(size_t)ch * HW + pix + k] = a[k]; p.g_img_next[(size_t)ch * HW + pix + k] = b[k]; }
  }
}

}  // namespace mal
namespace mal { opt_t g_dyn_small_blocks{1}; }  // option "dyn_small_blocks" (A/B): 0 = 1024-thread workgroups in dyn_extents16_kernel

using namespace mal;

static bool aligned_to(const void* ptr, size_t a) { return ptr == nullptr || (reinterpret_cast<uintptr_t>(ptr) & (a - 1)) == 0; }

static int dyn_check(int num, int C, int H, int W) {
  if (num < 1 || num > MAL_MAX_INSTANCES || C < 1 || C > 4) return MAL_EINVAL;
  if (H < 2 || W < 2 || (double)H * W > 2.0e9 / 4) return MAL_ESHAPE;
  return MAL_OK;
}

extern "C" size_t mal_dyn_workspace_bytes(int num) {
  return num > 0 ? align256((size_t)num * 8 * kExtChunks * sizeof(int)) : 0;
}

static int dyn_fwd_chunk(const mal_dyn_item* it, int n, int C, int H, int W, int replace, hipStream_t st) {
  DynBatch bt = {};
  int max_num = 0;
  for (int k = 0; k < n; ++k) {
    const mal_dyn_item& a = it[k];
    int rc = dyn_check(a.num, C, H, W);
    if (rc) return rc;
    if (!a.mask_last || !a.mask_next || !a.img_last || !a.img_next || !a.ori_last || !a.ori_next || !a.delta || !a.flags ||
        !a.ws)
      return MAL_EINVAL;
    if (a.ws_bytes < mal_dyn_workspace_bytes(a.num)) return MAL_EWORKSPACE;
    DynParams& p = bt.s[k];
    p.mask_last = a.mask_last; p.mask_next = a.mask_next; p.num = a.num; p.C = C; p.H = H; p.W = W; p.replace = replace;
    p.idx_last = (const long long*)a.idx_last; p.idx_next = (const long long*)a.idx_next;
    p.rows_last = a.n_last; p.rows_next = a.n_next;
    p.img_last = a.img_last; p.img_next = a.img_next; p.ori_last = a.ori_last; p.ori_next = a.ori_next;
    p.ext = (int*)a.ws; p.delta = a.delta; p.flags = a.flags; p.prefilled = a.prefilled;
    max_num = a.num > max_num ? a.num : max_num;
  }
  bool wide = (W & 15) == 0;
  for (int k = 0; k < n && wide; ++k) wide = aligned_to(bt.s[k].mask_last, 16) && aligned_to(bt.s[k].mask_next, 16);
  // (256 threads: one wave per SIMD -- a 1024-thread workgroup needs four slots of 32 VGPRs on EVERY SIMD of a CU at once
  // and waited for a whole marching pass beside which the --temporal step runs it: 45 us instead of 13)
  if (wide)
    hipLaunchKernelGGL(dyn_extents16_kernel, dim3(max_num, 2 * kExtChunks, n), dim3(g_dyn_small_blocks ? 256 : 1024), 0, st, bt);
  else
    hipLaunchKernelGGL(dyn_extents_kernel, dim3(max_num, 2 * kExtChunks, n), dim3(1024), (size_t)(H + W + 4) * sizeof(int), st, bt);
  bool quad = (W & 3) == 0 && C == 3;
  for (int k = 0; k < n && quad; ++k) {
    const DynParams& q = bt.s[k];
    quad = aligned_to(q.mask_last, 4) && aligned_to(q.mask_next, 4) && aligned_to(q.flags, 4) && aligned_to(q.img_last, 16) &&
           aligned_to(q.img_next, 16) && aligned_to(q.ori_last, 16) && aligned_to(q.ori_next, 16);
  }
  if (quad)
    hipLaunchKernelGGL(dyn_synth_fwd4_kernel<3>, dim3((H * W / 4 + 255) / 256, 1, n), dim3(256), (size_t)max_num * 4 * sizeof(int), st, bt);
  else
    hipLaunchKernelGGL(dyn_synth_fwd_kernel, dim3((H * W + 255) / 256, 1, n), dim3(256), (size_t)max_num * 4 * sizeof(int), st, bt);
  return launch_status();
}

static int dyn_bwd_chunk(const mal_dyn_item* it, int n, int C, int H, int W, hipStream_t st) {
  DynBatch bt = {};
  int max_num = 0;
  int region = 0;  // 0: out of place; 1: in place through scratch + move; 2: region snapshots, straight into g_img_*
  for (int k = 0; k < n; ++k) {
    const mal_dyn_item& a = it[k];
    int rc = dyn_check(a.num, C, H, W);
    if (rc) return rc;
    if (!a.mask_last || !a.mask_next || !a.delta || !a.flags || !a.g_ori_last || !a.g_ori_next ||
        (!a.g_img_last && !a.g_img_next))
      return MAL_EINVAL;
    DynParams& p = bt.s[k];
    p.mask_last = a.mask_last; p.mask_next = a.mask_next; p.num = a.num; p.C = C; p.H = H; p.W = W;
    p.idx_last = (const long long*)a.idx_last; p.idx_next = (const long long*)a.idx_next;
    p.rows_last = a.n_last; p.rows_next = a.n_next;
    p.delta = a.delta; p.flags = a.flags;
    p.g_ori_last = a.g_ori_last; p.g_ori_next = a.g_ori_next; p.g_img_last = a.g_img_last; p.g_img_next = a.g_img_next;
    p.g_tmp_last = a.g_tmp_last; p.g_tmp_next = a.g_tmp_next;
    const bool inplace = a.g_img_last == a.g_ori_last || a.g_img_next == a.g_ori_next;
    if (inplace && (a.g_img_last != a.g_ori_last || a.g_img_next != a.g_ori_next || !a.g_tmp_last || !a.g_tmp_next)) return MAL_EINVAL;
    if (a.region_only) {  // the cotangents are region snapshots: the region kernels write straight into g_img_*
      if (inplace || !a.g_img_last || !a.g_img_next) return MAL_EINVAL;
      p.g_tmp_last = a.g_img_last; p.g_tmp_next = a.g_img_next;
    }
    const int form = a.region_only ? 2 : (inplace ? 1 : 0);
    if (k == 0) region = form;
    else if (region != form) return MAL_EINVAL;
    max_num = a.num > max_num ? a.num : max_num;
  }
  bool quad = (W & 3) == 0 && C == 3;
  for (int k = 0; k < n && quad; ++k) {
    const DynParams& q = bt.s[k];
    quad = aligned_to(q.mask_last, 4) && aligned_to(q.mask_next, 4) && aligned_to(q.flags, 4) && aligned_to(q.g_ori_last, 16) && aligned_to(q.g_ori_next, 16) && aligned_to(q.g_img_last, 16) &&
           aligned_to(q.g_img_next, 16) && aligned_to(q.g_tmp_last, 16) && aligned_to(q.g_tmp_next, 16);
  }
  const size_t lds = (size_t)max_num * 4 * sizeof(int);
  const dim3 grid((H * W + 255) / 256, 1, n), grid4((H * W / 4 + 255) / 256, 1, n);
  if (region && quad) {
    hipLaunchKernelGGL((dyn_synth_bwd4_kernel<3, true>), grid4, dim3(256), lds, st, bt);
    if (region == 1) hipLaunchKernelGGL(dyn_apply_region4_kernel<3>, grid4, dim3(256), 0, st, bt);
  } else if (region) {
    hipLaunchKernelGGL(dyn_synth_bwd_kernel<true>, grid, dim3(256), lds, st, bt);
    if (region == 1) hipLaunchKernelGGL(dyn_apply_region_kernel, grid, dim3(256), 0, st, bt);
  } else if (quad) {
    hipLaunchKernelGGL((dyn_synth_bwd4_kernel<3, false>), grid4, dim3(256), lds, st, bt);
  } else {
    hipLaunchKernelGGL(dyn_synth_bwd_kernel<false>, grid, dim3(256), lds, st, bt);
  }
  return launch_status();
}

extern "C" int mal_dyn_batch_fwd(const mal_dyn_item* items, int n_items, int C, int H, int W, int replace, void* stream) {
  if (!items || n_items < 1) return MAL_EINVAL;
  for (int o = 0; o < n_items; o += kDynBatch) {
    int rc = dyn_fwd_chunk(items + o, n_items - o < kDynBatch ? n_items - o : kDynBatch, C, H, W, replace, (hipStream_t)stream);
    if (rc) return rc;
  }
  return MAL_OK;
}

extern "C" int mal_dyn_batch_bwd(const mal_dyn_item* items, int n_items, int C, int H, int W, void* stream) {
  if (!items || n_items < 1) return MAL_EINVAL;
  for (int o = 0; o < n_items; o += kDynBatch) {
    int rc = dyn_bwd_chunk(items + o, n_items - o < kDynBatch ? n_items - o : kDynBatch, C, H, W, (hipStream_t)stream);
    if (rc) return rc;
  }
  return MAL_OK;
}

extern "C" int mal_dyn_instance_fwd(const uint8_t* mask_last, const uint8_t* mask_next, int num, const float* img_last,
                                    const float* img_next, int C, int H, int W, int replace, float* ori_last,
                                    float* ori_next, int32_t* delta, uint8_t* flags, void* ws, size_t ws_bytes,
                                    void* stream) {
  mal_dyn_item a = {};
  a.mask_last = mask_last; a.mask_next = mask_next; a.num = num; a.img_last = img_last; a.img_next = img_next;
  a.ori_last = ori_last; a.ori_next = ori_next; a.delta = delta; a.flags = flags; a.ws = ws; a.ws_bytes = ws_bytes;
  return dyn_fwd_chunk(&a, 1, C, H, W, replace, (hipStream_t)stream);
}

extern "C" int mal_dyn_instance_bwd(const uint8_t* mask_last, const uint8_t* mask_next, int num, const int32_t* delta,
                                    const uint8_t* flags, const float* g_ori_last, const float* g_ori_next, int C, int H,
                                    int W, float* g_img_last, float* g_img_next, void* stream) {
  mal_dyn_item a = {};
  a.mask_last = mask_last; a.mask_next = mask_next; a.num = num; a.delta = const_cast<int32_t*>(delta);
  a.flags = const_cast<uint8_t*>(flags); a.g_ori_last = g_ori_last; a.g_ori_next = g_ori_next;
  a.g_img_last = g_img_last; a.g_img_next = g_img_next;
  return dyn_bwd_chunk(&a, 1, C, H, W, (hipStream_t)stream);
}

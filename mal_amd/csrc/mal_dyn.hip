// N2: the temporal-hint producer's per-sample arithmetic -- manydepth/dyn_utils.py:6-119,
// `fill_dynamic_obj` + `generate_dynamic_instance` -- given the matched instance masks of the two warped
// frames.  (Mask2Former and the Hungarian matcher that produce the masks stay outside this library.)
//
// Upstream this is a TorchScript loop over instances with (num, 3, H, W) temporaries per call and per
// sample; here it is three launches per sample and no temporaries:
//   dyn_extents_kernel   per (instance, frame): which rows / columns >= 1 hold mask pixels (the reference
//                        weighs the mask by the row / column index, so index 0 is invisible) -> low, top,
//                        right, left; then, per instance, the displacement: of (low_next-low_last,
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

struct DynParams {
  const uint8_t* mask_last; const uint8_t* mask_next;  // (num,H,W) bytes, non-zero = set
  // instance i is row idx_*[i] of the mask tensor (the matcher's selection, dyn_utils.py:147-150, applied here instead of
  // by one gather launch per sample and frame); nullptr = row i
  const long long* idx_last; const long long* idx_next;
  int num, C, H, W, replace;
  const float* img_last; const float* img_next;         // (C,H,W)
  float* ori_last; float* ori_next;
  int* ext;       // [num][2][kExtChunks][4]: low, top, right, left of (last, next), per band of rows
  int* delta;     // [num][2]: row, column displacement of the "last" copy
  uint8_t* flags; // [H*W]: bit0 region, bit1 any_last, bit2 any_next, bit3 last's background is img_next, bit4 next's is img_last
  const float* g_ori_last; const float* g_ori_next; float* g_img_last; float* g_img_next;
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
  const uint8_t* m = (which ? p.mask_next : p.mask_last) + (size_t)(sel ? sel[i] : i) * ((size_t)H * W);
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
  if (tid < 4) p.ext[((i * 2 + which) * kExtChunks + chunk) * 4 + tid] = res[tid];  // merged by dyn_delta_kernel
}

__global__ void dyn_delta_kernel(DynBatch bt) {
  const DynParams& p = bt.s[blockIdx.z];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.num) return;
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
  p.delta[i * 2] = dx; p.delta[i * 2 + 1] = dy;
}

// the per-instance small data of a sample in shared memory: displacement [num][2], then mask rows [num][2]
MAL_DEV void stage_instances(const DynParams& p, int* s_delta) {
  for (int k = threadIdx.x; k < p.num * 2; k += 256) {
    s_delta[k] = p.delta[k];
    const int i = k >> 1;
    const long long* sel = (k & 1) ? p.idx_next : p.idx_last;
    s_delta[2 * p.num + k] = sel ? (int)sel[i] : i;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void dyn_synth_fwd_kernel(DynBatch bt) {
  extern __shared__ int s_delta[];  // [num][2] displacements, [num][2] mask rows
  const DynParams& p = bt.s[blockIdx.z];
  stage_instances(p, s_delta);
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
  for (int ch = 0; ch < p.C; ++ch) {
    const float il = p.img_last[(size_t)ch * HW + pix], in = p.img_next[(size_t)ch * HW + pix];
    p.ori_last[(size_t)ch * HW + pix] = region ? (any_l ? accl[ch] : (bg_l ? in : il)) : il;
    p.ori_next[(size_t)ch * HW + pix] = region ? (any_n ? accn[ch] : (bg_n ? il : in)) : in;
  }
}

__global__ __launch_bounds__(256) void dyn_synth_bwd_kernel(DynBatch bt) {
  extern __shared__ int s_delta[];
  const DynParams& p = bt.s[blockIdx.z];
  stage_instances(p, s_delta);
  const int* s_row = s_delta + 2 * p.num;
  const int H = p.H, W = p.W, HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const int r = pix / W, c = pix - r * W;
  const int f = p.flags[pix];
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
  for (int ch = 0; ch < p.C; ++ch) {
    if (p.g_img_last) p.g_img_last[(size_t)ch * HW + pix] = gl[ch];
    if (p.g_img_next) p.g_img_next[(size_t)ch * HW + pix] = gn[ch];
  }
}

}  // namespace mal

using namespace mal;

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
    p.img_last = a.img_last; p.img_next = a.img_next; p.ori_last = a.ori_last; p.ori_next = a.ori_next;
    p.ext = (int*)a.ws; p.delta = a.delta; p.flags = a.flags;
    max_num = a.num > max_num ? a.num : max_num;
  }
  hipLaunchKernelGGL(dyn_extents_kernel, dim3(max_num, 2 * kExtChunks, n), dim3(1024), (size_t)(H + W + 4) * sizeof(int), st, bt);
  hipLaunchKernelGGL(dyn_delta_kernel, dim3(1, 1, n), dim3(64), 0, st, bt);
  hipLaunchKernelGGL(dyn_synth_fwd_kernel, dim3((H * W + 255) / 256, 1, n), dim3(256), (size_t)max_num * 4 * sizeof(int), st, bt);
  return launch_status();
}

static int dyn_bwd_chunk(const mal_dyn_item* it, int n, int C, int H, int W, hipStream_t st) {
  DynBatch bt = {};
  int max_num = 0;
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
    p.delta = a.delta; p.flags = a.flags;
    p.g_ori_last = a.g_ori_last; p.g_ori_next = a.g_ori_next; p.g_img_last = a.g_img_last; p.g_img_next = a.g_img_next;
    max_num = a.num > max_num ? a.num : max_num;
  }
  hipLaunchKernelGGL(dyn_synth_bwd_kernel, dim3((H * W + 255) / 256, 1, n), dim3(256), (size_t)max_num * 4 * sizeof(int), st, bt);
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

// Internal interface of the marching fused-pass kernels (mal_march.hip), shared with the
// whole-step launch list (mal_step.hip).
#pragma once
#include "mal_common.h"
#include "mal_device.h"
#include "mal_pose.h"

namespace mal {

struct MarchParams {
  const float* disp; const float* disp2; const float* K; const float* invK;
  const float* T[2]; const float* src[2];
  const float* target; const float* ident; const float* noise; const float* ext_mask; const float* sample_scale;
  const float* mono_depth; const float* mono_reproj; const float* ens_reproj;
  // whole-step launch list (mal_step.hip): the teacher's depth is re-derived from its disparity and
  // the matching mask (trainer.py:1066-1076) is formed in place, so neither map touches HBM
  const float* mono_disp; const float* lowest_cost; float* cmask_out;
  // whole-step list: with g_distil == nullptr the two epilogue gradients leave as ONE map,
  // merge_cons * d cons + merge_distil * d distil (their loss weights do not depend on the data)
  float merge_cons, merge_distil;
  // --learn_ens (whole-step list, epilogue passes): the learnt ensemble's disparity (its depth replaces (mono + multi) / 2
  // as the distillation target of the pixels the ensemble wins) and the map that receives merge_distil * d distil / d it
  const float* ens_disp; float* g_ens;
  // --dual_distil (whole-step list, two-way distillation): g_ens receives merge_distil * d distil / d mono_disp instead
  // (the teacher's depth keeps its graph where it wins the argmin, loss_utils.py:231-234); ens_disp must be NULL
  int dual_distil;
  // generic gradient passes only (march_launch leaves the specialisations when either is set): avg = --avg_reprojection, the
  // MEAN over the two candidates instead of their min (dualrefine/trainer.py:579-583; both take half of every gradient);
  // no_ssim = --no_ssim, r = mean_c |t - p| (manydepth/trainer.py:1217-1218, dualrefine/trainer.py:493-494)
  int avg, no_ssim;
  // per-sample camera block [B][40]: P_f = (K T_f)[:3,:], inv_K[:3,:3]; march_launch fills it unless cam_ready
  float* cam; int cam_ready;
  int sample_scale_is_mask;  // sample_scale holds the augmentation mask: the scale is 1 - mask
  int B, H, W; float min_disp, range, eps; int convention;
  float* min_reproj; float* g_reproj; float* g_cons; float* g_distil; float* cons_target; float* depth_out;
  double* block_sums; float* block_gP;
  // One-row halo of the gradient passes (nullable = two-row halo, every gradient row complete in g_reproj): boundary
  // scratch rows [B][segs][2][W]; row 0 / 1 of segment s = what the task ABOVE / BELOW contributes to the gradient of the
  // segment's first / last row (d / d disp, unnormalised like g_reproj).  The consumer adds them: march_boundary_add.
  float* bnd;
  int strips, segs, rows, ntasks, per_xcd;
  int packed;
  int flip_odd;  // odd row segments walk bottom-up: both tasks that share a segment boundary reach it together (L2 serves the halo)
  int debug;  // experiments only (mal_set_option("debug")): bit 0 = taps read the pixel's own address
  // ---- temporal hint (loss_utils.py:84-88: r(syn_f, target) joins the per-pixel min), whole-step list
  // forward-only pass that precedes the producer: the warped images as planar (B,3,H,W) (what image_synthesis reads,
  // dyn_utils.py:127-128) and the winner among the two warped candidates
  float* color_out[2]; unsigned char* argmin_out;
  // ... with the automask (MAL_F_AUTOMASK on that pass) it also leaves what the step needs where NO synthesised candidate
  // can win: second copies of the min / winner (the fused sweep over the synthesised pair overwrites them at the pixels it
  // re-decides; the first copies stay its read-only inputs) and the automask weight; block_sums = the sums over the two
  // warped candidates, which the fused sweep corrects by per-task differences
  float* min_reproj2; unsigned char* argmin_out2; float* weight_out;
  int color_out_stride;  // floats between two samples of color_out (0 = 3*H*W: contiguous (B,3,H,W))
  // a second copy, contiguous (B,3,H,W) each, nullable: the buffers the synthesised images are made in -- the producer
  // then overwrites only the pixels its instances touch instead of copying every sample (dyn_utils.py:127-128)
  float* color_out2[2];
  // TEMPORAL gradient pass (runs after the producer's backward): the decision of the four-way min and the automask are
  // TAKEN from the materialised-candidate kernels (forced_arg: winner 0..3, forced_w: automask weight), and the
  // gradient that reaches the warped images through syn (dyn_utils.py:145-146,163-164), g_color[f] planar (B,3,H,W),
  // joins d loss / d warped colour before the chain rule through the warp
  const float* forced_w; const unsigned char* forced_arg; const float* g_color[2];
  // ... as a CORRECTION of what an exporting gradient pass (color_out with MAL_F_GRAD) left: (B,H,W) bytes, bit 0 = the
  // pixel can differ between syn and the warped image; tasks that keep away from every marked pixel return at once (their
  // gradient rows, boundary rows and pose partials stand).  nullptr = every task runs.
  const unsigned char* region;
  // ... and, running in the backward call where the loss scalars exist already, it finishes the teacher's disparity
  // gradient itself instead of leaving the unnormalised map to the assembly kernel (fin_out nullable):
  //   fin_out = coefs[0] g_total * G + coefs[4] g_total * (fin_gn / (mean_b + 1e-7) - corr_b),  mean / corr in fin_stats
  const float* fin_gn; const float* fin_coefs; const double* fin_stats; const float* fin_g_total; float* fin_out;
  // Parity instrumentation (tests only; DBG instantiations of the gradient passes): the per-pixel DECISIONS the pass
  // took, as kDecPlanes uint32 planes of B*H*W each (include/mal_hip.h, MAL_DEC_*).  nullptr = the production kernels.
  unsigned* dbg;
  // FRAMED gradient pass (DualRefine's pose-update losses, dualrefine/trainer.py:457-480,699-767: the two candidates of the
  // min are warped with two DIFFERENT disparities -- frame -1 with disp, frame +1 with disp2): g_reproj receives d / d disp
  // (what frame -1's candidate contributes), g_reproj2 d / d disp2 (frame +1's); bnd2 = the boundary scratch rows of g_reproj2
  int framed; float* g_reproj2; float* bnd2;
};
constexpr int kDecPlanes = MAL_DEC_PLANES;

constexpr int kCamFloats = 40;  // P interleaved over the two frames [12][2], inv_K 3x3, 7 pad

// entry `lane` (< kCamFloats) of sample b's camera block: P_f = (K T_f)[:3,:] with ATen's bmm association, inv_K[:3,:3]
MAL_DEV void cam_fill(const float* K, const float* T0, const float* T1, const float* invK, float* cam, int b, int lane) {
  float mine = 0.f;
  if (lane < 24) {
    const int f = lane & 1, e = lane >> 1, i = e >> 2, j = e & 3;
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

// zero-initialised parameter block with the depth range / convention filled in
MarchParams march_params(int B, int H, int W, float min_depth, float max_depth, float eps, int convention);
// fills the task decomposition for `flags` (MAL_F_*), launches the matching instantiation on `st`
// (bracketed by the one-shot profile events if armed).  Partials go to p.block_sums / p.block_gP.
int march_launch(MarchParams& p, int flags, hipStream_t st);
// the decomposition march_launch picks for (B,H,W) and `flags` on the current device
void march_geometry(int B, int H, int W, int flags, int* strips, int* segs, int* rows);
// floats of a boundary scratch buffer (MarchParams::bnd) that fits every decomposition of (B,H,W)
inline size_t march_bnd_floats(int B, int H, int W) { return (size_t)B * ((size_t)H / 8 + 1) * 2 * (size_t)W; }
// what the boundary scratch rows add to pixel (b, y, x) of the gradient map (0 away from segment boundaries); y, rows,
// segs wave-uniform in the callers
MAL_DEV float march_boundary_term(const float* bnd, int b, int y, int x, int H, int W, int rows, int segs) {
  const int sq = y / rows, within = y - sq * rows;
  const int last = min(rows, H - sq * rows) - 1;
  float v = 0.f;
  if (within == 0 && sq > 0) v = bnd[((size_t)(b * segs + sq) * 2 + 0) * W + x];
  else if (within == last && sq < segs - 1) v = bnd[((size_t)(b * segs + sq) * 2 + 1) * W + x];
  return v;
}
// the scratch row that completes image row y of sample b (nullptr away from segment boundaries); wave-uniform
MAL_DEV const float* march_boundary_row(const float* bnd, int b, int y, int H, int W, int rows, int segs) {
  const int sq = y / rows, within = y - sq * rows;
  const int last = min(rows, H - sq * rows) - 1;
  if (within == 0 && sq > 0) return bnd + ((size_t)(b * segs + sq) * 2 + 0) * W;
  if (within == last && sq < segs - 1) return bnd + ((size_t)(b * segs + sq) * 2 + 1) * W;
  return nullptr;
}
// tasks per sample of pack_identity_launch's decomposition (the per-task smoothness partials are laid out by it)
int pack_identity_tasks_per_sample(int H, int W);
// min_f r(src_f, target) of the RAW sources -> ident (B,1,H,W) (the identity term of
// manydepth/loss_utils.py:92-101), and in the same sweep the two planar sources repacked as (B,H,W,kTexel) texels
// (packed0/1 nullable together), optionally the target too.  With `poses` (the whole-step launch list) B extra
// workgroups of the same launch compose the poses of both frames (layers.py:26-100), fill the camera block of
// the marching passes and reset the completion counter of the step's last kernel.
struct StepPoses { PoseParams pose; const float* K; const float* invK; float* cam; unsigned* ticket; };
// Tie-break noise of the automask (loss_utils.py:105-106: identity + 1e-5 * randn) generated in the sweep and ADDED to
// the identity map it writes: N(0,1) from Philox4x32-10 keyed by `seed`, counter = (pixel group, step) with
// step = *counter (device, advanced by the step's last kernel so graph replays draw fresh noise) or `step`.
struct TieNoise { int on; unsigned long long seed, step; const unsigned long long* counter; float* noise_out /*nullable*/; };
// Edge-aware smoothness (get_smooth_loss of the mean-normalised disparity, layers.py:210-223 + loss_utils.py:119-121)
// of n <= 2 disparity maps in the identity/packing sweep: gn[m] = d loss / d normalised-disp map out (unnormalised by
// the sample's 1/(mean+1e-7)), partials [task][m][4] = sum |dx d| w, sum |dy d| w, sum gn*disp, sum disp;
// dec[m] (tests, nullable): decision planes receiving the signs taken (MAL_DEC_SMOOTH_X / _Y).
struct SmoothParams { int n; const float* disp[2]; float* gn[2]; double* partials; unsigned* dec[2]; };
// DualRefine's one-call step (mal_dr_step.hip): what its prologue does -- the camera block (K T)[:3,:] of both frames + inv_K of
// sample b for every slot's poses (slot = deq iteration; pu_slot: the pose-update pass) and, with MAL_DR_NOISE_PHILOX, every
// slot's N(0,1) map -- as extra workgroups of the identity / packing sweep instead of a launch of its own behind it (neither
// needs the other).  noise_blocks: 64-pixel-group workgroups per slot (0: no noise), then B camera workgroups per slot.
constexpr int kDrSlotsMax = 5;  // MAL_DR_MAX_ITERS + the pose-update pass
struct DrExtra {
  const float* K; const float* invK; const float* T[kDrSlotsMax][2]; float* cam[kDrSlotsMax]; float* noise[kDrSlotsMax];
  unsigned long long seed, step; const unsigned long long* counter;
  int slots, pu_slot, noise_blocks, B, H, W; unsigned* ticket;
};
// the arguments of one fused sweep over a materialised candidate pair (photo_march_fused_more, mal_photo_march.hip), for the
// batched launch photo_march_fused_more_n
struct FusedMoreArgs {
  const float* target; const float* cand0; const float* cand1; int idx0; const float* ident; const float* noise;
  const float* prev_min; const uint8_t* prev_arg; float* min_reproj; uint8_t* argmin; float* weight_out; double* block_sums;
  float* g_cand0; float* g_cand1; const uint8_t* region; float* g_region0; float* g_region1; const float* orig0; const float* orig1;
  size_t orig_stride; int weight_given;
};
int photo_march_fused_more_n(int n, const FusedMoreArgs* a, int B, int H, int W, int* per_sample_out, hipStream_t st);
// The four-scale step (mal_step_ms.hip): its per-scale tie-break noise maps (MAL_STEP_NOISE_PHILOX) and the bilinear
// upsampling of the lower scales' disparities to full resolution (trainer.py:1094-1096) -- neither reads anything the
// identity / packing sweep writes -- as extra workgroups of that sweep's launch instead of two launches behind it.
struct UpMaps { const float* src[2 * MAL_MS_MAX_SCALES]; float* dst[2 * MAL_MS_MAX_SCALES]; int h[2 * MAL_MS_MAX_SCALES], w[2 * MAL_MS_MAX_SCALES]; };
struct MsExtra {
  unsigned long long seed, step; const unsigned long long* counter; unsigned mult; int n_noise; float* noise[MAL_MS_MAX_SCALES];
  UpMaps up; int n_up;            // maps to upsample, four output pixels per thread (W % 4 == 0, 16-byte aligned destinations)
  int B, H, W, noise_blocks, up_blocks;  // 64-thread workgroups per noise map / per upsampled map
};
int pack_identity_launch(const float* target, const float* src0, const float* src1, int B, int H, int W,
                         float* packed0, float* packed1, float* packed_target, float* ident, hipStream_t st,
                         const StepPoses* poses = nullptr, const TieNoise* noise = nullptr,
                         const SmoothParams* smooth = nullptr, int* tasks_per_sample = nullptr, bool texel_in = false,
                         int variant = 0 /* bit 0 --no_ssim, bit 1 --avg_reprojection: the identity term follows */,
                         const DrExtra* dr = nullptr, const MsExtra* ms = nullptr);

// Philox4x32-10 (Salmon et al., SC'11; the generator behind torch's device randn), one block of four 32-bit words
MAL_DEV void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// the four N(0,1) values of pixel group `group` (= index of the pixel in row (y & ~3) of its column): Box-Muller on
// the two word pairs; value j belongs to row (y & ~3) + j
MAL_DEV void tie_noise4(unsigned long long seed, unsigned long long step, unsigned group, float (&n)[4]) {
  unsigned o[4];
  philox4x32_10(group, (unsigned)step, (unsigned)(step >> 32), 0x4d414cu, (unsigned)seed, (unsigned)(seed >> 32), o);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)o[2 * h] + 1.0f) * 2.3283064365386963e-10f;   // (0, 1]
    const float u2 = (float)o[2 * h + 1] * 2.3283064365386963e-10f;       // [0, 1]
    const float r = __builtin_sqrtf(-2.0f * __logf(u1));
    n[2 * h] = r * __cosf(6.283185307179586f * u2);
    n[2 * h + 1] = r * __sinf(6.283185307179586f * u2);
  }
}

// ---- bilinear upsampling (align_corners=False), as mal_step_ms.hip applies it
// ATen's area_pixel_compute_source_index: src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out (float);
// i0 = (int)src, i1 = i0 + (i0 < in - 1), lambda1 = src - i0, lambda0 = 1 - lambda1.
struct Tap { int i0, i1; float l0, l1; };
MAL_DEV Tap tap_of(int dst, float scale, int n_in) {
  float src = fma_(scale, (float)dst + 0.5f, -0.5f);  // contracted in ATen's builds (exact either way for power-of-two factors)
  src = src < 0.f ? 0.f : src;
  Tap t;
  t.i0 = (int)src;
  t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.0f - t.l1;
  return t;
}
// h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d) as ATen's builds contract it (device and host kernels alike, measured
// bit for bit by scripts/upsample_probe.py): the first product of each sum becomes the fused multiply-add
MAL_DEV float bilinear_value(float h0, float h1, float w0, float w1, float a, float b, float c, float d) {
  const float top = fma_(w0, a, w1 * b), bot = fma_(w0, c, w1 * d);
  return fma_(h0, top, h1 * bot);
}
// V consecutive output pixels (row `row` = b * H + y, columns x0 ..) of map k
template <int V>
MAL_DEV void upsample_pixels(const UpMaps& m, int k, int row, int x0, int H, int W) {
  const int h = m.h[k], w = m.w[k];
  const int b = row / H, y = row - b * H;
  const Tap ty = tap_of(y, (float)h / (float)H, h);
  const float* s0 = m.src[k] + (size_t)b * h * w + ty.i0 * w;
  const float* s1 = m.src[k] + (size_t)b * h * w + ty.i1 * w;
  float o[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const Tap tx = tap_of(x0 + v, (float)w / (float)W, w);
    o[v] = bilinear_value(ty.l0, ty.l1, tx.l0, tx.l1, s0[tx.i0], s0[tx.i1], s1[tx.i0], s1[tx.i1]);
  }
  float* d = m.dst[k] + (size_t)row * W + x0;
  if (V == 4) *reinterpret_cast<float4*>(d) = make_float4(o[0], o[1], o[2], o[3]);
  else d[0] = o[0];
}
// workgroup e (64 threads) of an MsExtra: the noise maps' groups first (map e / noise_blocks; numbering of
// tiebreak_noise_kernel: step number (counter ? *counter : step) * mult + map), then the upsampled maps' quads
MAL_DEV void ms_extra_block(const MsExtra& p, int e, int tid) {
  const int nb = p.n_noise * p.noise_blocks;
  if (e < nb) {
    const int k = e / p.noise_blocks, r = e - k * p.noise_blocks;
    const int H = p.H, W = p.W, H4 = (H + 3) >> 2;
    const size_t i = (size_t)r * 64 + tid;
    if (i >= (size_t)p.B * H4 * W) return;
    const int x = (int)(i % W), row = (int)(i / W), y4 = row % H4, b = row / H4, y0 = y4 * 4;
    const unsigned long long st = (p.counter ? *p.counter : p.step) * p.mult + (unsigned)k;
    float n[4];
    tie_noise4(p.seed, st, (unsigned)(b * H * W + y0 * W + x), n);
    float* o = p.noise[k] + (size_t)b * H * W + (size_t)y0 * W + x;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (y0 + j < H) o[(size_t)j * W] = n[j];
    return;
  }
  e -= nb;
  const int k = e / p.up_blocks, r = e - k * p.up_blocks;
  const int W4 = p.W >> 2;
  const size_t q = (size_t)r * 64 + tid;  // quad index inside the map: row-major over (B*H, W/4)
  if (q >= (size_t)p.B * p.H * W4) return;
  upsample_pixels<4>(p.up, k, (int)(q / W4), (int)(q % W4) * 4, p.H, p.W);
}

// workgroup e (64 threads) of a DrExtra: slot e / (noise_blocks + B); the slot's noise groups first, then its camera blocks.
// MAL_DR_NOISE_PHILOX's numbering: key seed (the pose-update pass: seed ^ MAL_DR_POSE_NOISE_KEY), step number
// (counter ? *counter : step) * MAL_DR_MAX_ITERS + slot (the pose-update pass: + 0), value layout of tiebreak_noise_kernel
MAL_DEV void dr_extra_block(const DrExtra& p, int e, int tid) {
  const int per = p.noise_blocks + p.B, it = e / per, r = e - it * per;
  if (e == 0 && tid < 4) p.ticket[tid] = 0u;
  if (r >= p.noise_blocks) {
    cam_fill(p.K, p.T[it][0], p.T[it][1], p.invK, p.cam[it], r - p.noise_blocks, tid);
    return;
  }
  const int H = p.H, W = p.W, H4 = (H + 3) >> 2;
  const size_t i = (size_t)r * 64 + tid;
  if (i >= (size_t)p.B * H4 * W) return;
  const int x = (int)(i % W), row = (int)(i / W), y4 = row % H4, b = row / H4, y0 = y4 * 4;
  const bool pu = it == p.pu_slot;
  const unsigned long long st = (p.counter ? *p.counter : p.step) * (unsigned long long)MAL_DR_MAX_ITERS + (pu ? 0 : it);
  float n[4];
  tie_noise4(pu ? p.seed ^ MAL_DR_POSE_NOISE_KEY : p.seed, st, (unsigned)(b * H * W + y0 * W + x), n);
  float* o = p.noise[it] + (size_t)b * H * W + (size_t)y0 * W + x;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (y0 + j < H) o[(size_t)j * W] = n[j];
}

}  // namespace mal

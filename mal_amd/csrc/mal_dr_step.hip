// mal_dr_loss_fwd / _bwd: DualRefine's loss loops (dualrefine/trainer.py:395-451 generate_images_pred, :530-633
// compute_losses, the f_thres > 0 branch) over the deq iterations of scale 0 as ONE host call per direction.
//
// Through the operator-level API (mal_amd/dualrefine.py) the same step is ~75 launches, most of them 4 us of scalar glue
// between two 48 us marching passes (rocprofv3, B=8 192x640: 0.36 ms per step of which 0.10 ms are the passes).  Here:
//   forward   1 prologue: camera blocks of every iteration's poses + (MAL_DR_NOISE_PHILOX) the tie-break noise maps
//             1 identity term min_f r(src_f, target) (dualrefine/trainer.py:563-573) + texel packing of the three images
//             1 edge-aware smoothness sweep over all iterations' disparities (:620-624)
//             per deq iteration ONE marching pass with convention B (Project3D 2(u+0.5)/W-1, align_corners=False):
//               warp + SSIM/L1 + min over the two frames + automask against identity + 1e-5 noise_it (:575-592),
//               for it > 0 times consistency_mask (:593-597) and with the consistency term against iteration 0's
//               depth in its epilogue (:606-618); gradient w.r.t. the iteration's disparity and both poses
//             1 reduction: sums, pose gradients, loss scalars (upstream's running `loss` is added to `total` once per
//               iteration, so iteration it weighs (n - it), :624-631), coefficients of the backward
//   backward  1 assembly: d total / d disp_it, d total / d T (the 4x4 matrices the trainer hands in: refined or not,
//             detached or not is the caller's business -- a NULL output pointer is a detached pose)
// MAL_DR_POSE_UPDATE (round 5): the pose-update losses (dualrefine/trainer.py:457-480,699-767) as ONE more marching pass of
// the same call -- slot n_iters of every per-iteration array below.  Its two candidates are warped with two different
// disparities (MarchParams::framed: frame -1 with the last iteration's disparity under the refined pose, frame +1 as
// ("color", 1, 0, 0) was, with iteration 0's), so it leaves two gradient maps; no consistency mask, no smoothness, its own
// tie-break noise, its own cotangent (g_pu_total: upstream adds the term after the division by len(scales), :337-343).
#include "mal_march.h"
#include "mal_device.h"

namespace mal {

extern opt_t g_march_halo1;  // mal_step.hip
int smooth_march_sweep_batch(int n, const float* const* disp, const float* const* img, int B, const int* H, const int* W,
                             float* const* gn, double* const* partial, hipStream_t st, int* per_sample);

constexpr int kDrIt = MAL_DR_MAX_ITERS;
constexpr int kDrSlots = kDrIt + 1;  // + the pose-update pass (slot n_iters)

struct DrWs {
  float* packed[3]; float* ident; float* cam[kDrSlots];
  float* G_r[kDrSlots]; float* G_c[kDrIt]; float* gn[kDrIt]; float* bnd[kDrSlots];
  double* bs[kDrSlots]; float* bgP[kDrSlots]; double* sm[kDrIt];
  float* G_r2; float* bnd2;  // pose-update pass: d / d (frame +1's disparity)
  double* ps;      // [slot][B][8]
  double* stats;   // [2][it][B]: mean, mean-coupling term of the smoothness
  float* gT;       // [slot][2][B*16]
  float* coefs;    // [slot][4]: cR, cC, cS, weight
  float* noise[kDrSlots];  // MAL_DR_NOISE_PHILOX: the iteration's N(0,1) map
  unsigned* ticket;
  size_t bytes;
};

static DrWs carve_dr(void* base, int B, int H, int W, int n) {
  DrWs w = {};
  char* p = (char*)base;
  size_t o = 0;
  const size_t HW = (size_t)H * W, map = B * HW * sizeof(float), nb = ws_blocks(B, H, W);
  auto take = [&](size_t bytes) { char* r = p + o; o += align256(bytes); return r; };
  for (int i = 0; i < 3; ++i) w.packed[i] = (float*)take(B * HW * kTexel * sizeof(float));
  w.ident = (float*)take(map);
  for (int it = 0; it < n; ++it) {
    w.cam[it] = (float*)take((size_t)B * kCamFloats * 4);
    w.G_r[it] = (float*)take(map); w.G_c[it] = (float*)take(map); w.gn[it] = (float*)take(map);
    w.bnd[it] = (float*)take(march_bnd_floats(B, H, W) * sizeof(float));
    w.bs[it] = (double*)take(nb * 8 * 8); w.bgP[it] = (float*)take(nb * 24 * 4); w.sm[it] = (double*)take(nb * 8 * 8);
    w.noise[it] = (float*)take(map);
  }
  w.ps = (double*)take((size_t)kDrSlots * B * 8 * 8);
  w.stats = (double*)take((size_t)2 * kDrIt * B * 8);
  w.gT = (float*)take((size_t)kDrSlots * 2 * B * 16 * 4);
  w.coefs = (float*)take(kDrSlots * 4 * 4);
  w.ticket = (unsigned*)take(16);
  for (int it = n; it < n + 1; ++it) {  // the pose-update pass's slot (behind everything an earlier layout had: texels_from)
    w.cam[it] = (float*)take((size_t)B * kCamFloats * 4);
    w.G_r[it] = (float*)take(map); w.G_r2 = (float*)take(map);
    w.bnd[it] = (float*)take(march_bnd_floats(B, H, W) * sizeof(float)); w.bnd2 = (float*)take(march_bnd_floats(B, H, W) * sizeof(float));
    w.bs[it] = (double*)take(nb * 8 * 8); w.bgP[it] = (float*)take(nb * 24 * 4);
    w.noise[it] = (float*)take(map);
  }
  w.bytes = o;
  return w;
}

struct DrFinal {
  const double* bs[kDrSlots]; const float* bgP[kDrSlots]; const double* sm[kDrIt];
  const float* K;
  int per_sample, per_sample_sm, sm_stride, B, H, W, n;
  int pu;  // 1: slot n holds the pose-update pass (sums and pose partials only)
  int hs, ws;  // size of the maps the smoothness term is taken on (H >> scale, W >> scale)
  float smooth_weight;
  double* ps; double* stats; float* gT; float* losses; float* coefs; float* loss_total; float* pu_loss_total; unsigned* ticket;
  unsigned long long* noise_counter;  // nullable: advanced by one when every reader of this step has finished
};

// The step's first launch.  blockIdx.y = iteration; blockIdx.x < noise_blocks: the iteration's N(0,1) map from Philox
// (MAL_DR_NOISE_PHILOX; step number = (counter ? *counter : step) * kDrIt + it, value layout of tiebreak_noise_kernel, so
// mal_tiebreak_noise reproduces a map); the B workgroups after them: the camera block (K T)[:3,:] of both frames +
// inv_K[:3,:3] of sample b for that iteration's poses; workgroup (0,0) also clears the completion ticket.
struct DrPrologue {
  const float* K; const float* invK; const float* T[kDrSlots][2]; float* cam[kDrSlots]; float* noise[kDrSlots];
  unsigned long long seed, step; const unsigned long long* counter;
  int noise_blocks, B, H, W; unsigned* ticket;
  int pu_slot;  // >= 0: that slot is the pose-update pass (its noise: key seed ^ MAL_DR_POSE_NOISE_KEY, step number step * kDrIt)
};
__global__ __launch_bounds__(256) void dr_prologue_kernel(DrPrologue p) {
  const int it = blockIdx.y, tid = threadIdx.x;
  if (blockIdx.x == 0 && it == 0 && tid < 4) p.ticket[tid] = 0u;
  if ((int)blockIdx.x >= p.noise_blocks) {
    if (tid < 64) cam_fill(p.K, p.T[it][0], p.T[it][1], p.invK, p.cam[it], (int)blockIdx.x - p.noise_blocks, tid);
    return;
  }
  const int H = p.H, W = p.W, H4 = (H + 3) >> 2;
  const size_t i = blockIdx.x * (size_t)256 + tid;
  if (i >= (size_t)p.B * H4 * W) return;
  const int x = (int)(i % W), row = (int)(i / W), y4 = row % H4, b = row / H4, y0 = y4 * 4;
  const bool pu = it == p.pu_slot;
  const unsigned long long st = (p.counter ? *p.counter : p.step) * (unsigned long long)kDrIt + (pu ? 0 : it);
  float n[4];
  tie_noise4(pu ? p.seed ^ MAL_DR_POSE_NOISE_KEY : p.seed, st, (unsigned)(b * H * W + y0 * W + x), n);
  float* o = p.noise[it] + (size_t)b * H * W + (size_t)y0 * W + x;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (y0 + j < H) o[(size_t)j * W] = n[j];
}

// 2 n B workgroups: [0, nB): ps[it][b][j] = sum over the sample's tasks of the marching pass's partials (j < 4) and of the
// smoothness sweep's (j >= 4); [nB, 2nB): gT[it][f][b] = K_b^T [gP_fb ; 0].  The workgroup that finishes last forms the loss
// scalars and the coefficients of the backward (fixed summation order, no floating-point atomics).
__global__ __launch_bounds__(256) void dr_final_kernel(DrFinal p) {
  __shared__ double s_part[256];
  __shared__ double s_gP[24];
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, B = p.B, HW = p.H * p.W, nB = (p.n + p.pu) * B;
  if ((int)blockIdx.x < nB) {
    const int it = blockIdx.x / B, b = blockIdx.x - it * B;
    const int j = tid & 7, sub = tid >> 3;
    double acc = 0.0;
    if (j < 4) {
      const double* q = p.bs[it] + (size_t)b * p.per_sample * 8 + j;
#pragma unroll 8
      for (int t = sub; t < p.per_sample; t += 32) acc += q[(size_t)t * 8];
    } else if (it < p.n) {
      const double* q = p.sm[it] + (size_t)b * p.per_sample_sm * p.sm_stride + (j - 4);
#pragma unroll 8
      for (int t = sub; t < p.per_sample_sm; t += 32) acc += q[(size_t)t * p.sm_stride];
    }
    s_part[tid] = acc;
    __syncthreads();
    if (tid < 8) {
      double a = 0.0;
      for (int k = 0; k < 32; ++k) a += s_part[k * 8 + tid];
      p.ps[((size_t)it * B + b) * 8 + tid] = a;
    }
  } else {
    const int i2 = blockIdx.x - nB, it = i2 / B, b = i2 - it * B;
    const int v = tid % 24, sub = tid / 24;
    double acc = 0.0;
    if (sub < 10) {
#pragma unroll 8
      for (int t = sub; t < p.per_sample; t += 10) acc += (double)p.bgP[it][((size_t)b * p.per_sample + t) * 24 + v];
    }
    s_part[tid] = acc;
    __syncthreads();
    if (tid < 24) {
      double a = 0.0;
      for (int k = 0; k < 10; ++k) a += s_part[k * 24 + tid];
      s_gP[tid] = a;
    }
    __syncthreads();
    if (tid < 32) {
      const int f = tid >> 4, e = tid & 15, k = e >> 2, j = e & 3;
      const float* Kb = p.K + b * 16;
      double a = 0.0;
      for (int i = 0; i < 3; ++i) a += (double)Kb[i * 4 + k] * s_gP[f * 12 + i * 4 + j];
      p.gT[((size_t)it * 2 + f) * B * 16 + b * 16 + e] = (float)a;
    }
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(p.ticket, 1u);
  __syncthreads();
  if (s_last != gridDim.x - 1) return;
  __threadfence();
  // per-sample statistics of the smoothness gradient (mean and the mean-coupling term), then the scalars: thread it
  const int HWs = p.hs * p.ws;
  for (int s = tid; s < p.n * B; s += 256) {
    const double* q = p.ps + (size_t)s * 8;
    const double mean = q[7] / (double)HWs;
    const double m = (double)((float)mean + 1e-7f);
    p.stats[s] = mean;
    p.stats[kDrIt * B + s] = q[6] / ((double)HWs * m * m);
  }
  __syncthreads();
  if (tid < p.n) {
    const int it = tid;
    double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b) {
      const double* q = p.ps + ((size_t)it * B + b) * 8;
      for (int j = 0; j < 8; ++j) {
        double v = q[j];
        if (j == 4 || j == 5) v = v * (double)div_(1.0f, (float)(q[7] / (double)HWs) + 1e-7f);
        tot[j] += v;
      }
    }
    const double N = (double)B * HW, Nx = (double)B * p.hs * (p.ws - 1), Ny = (double)B * (p.hs - 1) * p.ws;
    p.losses[it * 4 + 0] = (float)(tot[0] / (tot[1] + 1e-7));
    p.losses[it * 4 + 1] = it > 0 ? (float)(tot[2] / N) : 0.0f;
    p.losses[it * 4 + 2] = (float)(tot[4] / Nx + tot[5] / Ny);
    const float wt = (float)(p.n - it);  // upstream adds its RUNNING loss to the total once per iteration (:624-631)
    p.coefs[it * 4 + 0] = (float)((double)wt / (tot[1] + 1e-7));
    p.coefs[it * 4 + 1] = wt;                       // the consistency map already carries 1/N
    p.coefs[it * 4 + 2] = wt * p.smooth_weight;
    p.coefs[it * 4 + 3] = wt;
  }
  if (p.pu && tid == p.n) {  // the pose-update term: masked mean over its own pass, weight 1 (dualrefine/trainer.py:750-758)
    double t0 = 0.0, t1 = 0.0;
    for (int b = 0; b < B; ++b) {
      const double* q = p.ps + ((size_t)p.n * B + b) * 8;
      t0 += q[0]; t1 += q[1];
    }
    p.losses[4 * kDrIt + 2] = (float)(t0 / (t1 + 1e-7));
    if (p.pu_loss_total) *p.pu_loss_total = p.losses[4 * kDrIt + 2];
    p.coefs[p.n * 4 + 0] = (float)(1.0 / (t1 + 1e-7));
  }
  __threadfence();
  __syncthreads();
  if (tid != 0) return;
  // upstream's fp32 arithmetic: loss += reproj + consistency; loss += w * smooth; total += loss   (one running tensor)
  float total = 0.f, running = 0.f;
  for (int k = 0; k < p.n; ++k) {
    running = (running + p.losses[k * 4 + 0]) + p.losses[k * 4 + 1];
    running = running + p.smooth_weight * p.losses[k * 4 + 2];
    p.losses[k * 4 + 3] = running;  // the running loss after iteration k
    total = total + running;
  }
  p.losses[4 * kDrIt] = total;       // num_scales = 1
  p.losses[4 * kDrIt + 1] = running; // what every "loss/0_it" entry reads upstream (one aliased tensor)
  if (p.loss_total) *p.loss_total = total;
  if (p.noise_counter) *p.noise_counter += 1ull;  // every kernel of this step that reads it has finished (stream order)
}

struct DrAssemble {
  const float* G_r[kDrSlots]; const float* G_c[kDrIt]; const float* gn[kDrIt]; const float* bnd[kDrSlots];
  float* g_disp[kDrSlots]; float* g_T[kDrSlots][2];
  const float* G_r2; const float* bnd2; float* g_disp2; const float* g_pu_total;  // pose-update pass (slot n, grid y = n)
  int pu_on;          // the pose-update pass ran
  int pu_into[4];     // its gradient w.r.t. {frame -1's disparity, frame +1's, T_m1, T_p1} joins iteration k's output: k + 1 (0: own output)
  float* g_disp_lo[kDrIt];  // scale > 0: the smoothness term's gradient at the scale's own size (then not part of g_disp)
  const float* gT; const float* coefs; const double* stats; const float* g_total;
  int B, H, W, n, rows, segs, hs, ws;
};

// blockIdx.y = iteration; one image row per workgroup round (the boundary test of the one-row halo is wave-uniform);
// workgroup 0 of each iteration also scales the pose gradients
__global__ __launch_bounds__(256) void dr_assemble_kernel(DrAssemble p) {
  const int it = blockIdx.y, B = p.B, H = p.H, W = p.W;
  const bool pu = it == p.n;
  const float g = pu ? (p.g_pu_total ? *p.g_pu_total : 1.0f) : (p.g_total ? *p.g_total : 1.0f);
  const float cR = p.coefs[it * 4 + 0] * g, cC = pu ? 0.f : p.coefs[it * 4 + 1] * g, cS = pu ? 0.f : p.coefs[it * 4 + 2] * g;
  // the pose-update term's coefficient, for what it adds into THIS iteration's outputs (pu_into)
  const float cP = (p.pu_on && !pu) ? p.coefs[p.n * 4 + 0] * (p.g_pu_total ? *p.g_pu_total : 1.0f) : 0.f;
  if (blockIdx.x == 0) {
    for (int f = 0; f < 2; ++f) {
      float* out = p.g_T[it][f];
      if (!out) continue;
      const float* in = p.gT + ((size_t)it * 2 + f) * B * 16;
      const float* inP = (p.pu_on && !pu && p.pu_into[2 + f] == it + 1) ? p.gT + ((size_t)p.n * 2 + f) * B * 16 : nullptr;
      for (int i = threadIdx.x; i < B * 16; i += 256) out[i] = inP ? fma_(inP[i], cP, in[i] * cR) : in[i] * cR;
    }
  }
  if (pu) {  // two maps, the reprojection term only: d / d (frame -1's disparity), d / d (frame +1's)
    for (int m = 0; m < 2; ++m) {
      float* o2 = m ? p.g_disp2 : p.g_disp[it];
      if (!o2) continue;
      const float* G = m ? p.G_r2 : p.G_r[it];
      const float* bn = m ? p.bnd2 : p.bnd[it];
      for (int row = blockIdx.x; row < B * H; row += gridDim.x) {
        const int b = row / H, y = row - b * H;
        const float* brow = bn ? march_boundary_row(bn, b, y, H, W, p.rows, p.segs) : nullptr;
        const size_t r0 = (size_t)row * W;
        for (int x = threadIdx.x; x < W; x += 256) o2[r0 + x] = cR * (brow ? G[r0 + x] + brow[x] : G[r0 + x]);
      }
    }
    return;
  }
  float* out = p.g_disp[it];
  float* lo = p.g_disp_lo[it];
  const float* gn = p.gn[it];
  if (lo) {  // the smoothness term of a lower scale: its own, smaller map
    for (int row = blockIdx.x; row < B * p.hs; row += gridDim.x) {
      const int b = row / p.hs;
      const float inv = div_(1.0f, (float)p.stats[it * B + b] + 1e-7f), corr = (float)p.stats[kDrIt * B + it * B + b];
      const size_t r0 = (size_t)row * p.ws;
      for (int x = threadIdx.x; x < p.ws; x += 256) lo[r0 + x] = cS * (gn[r0 + x] * inv - corr);
    }
  }
  if (!out) return;
  const bool with_smooth = p.hs == H && p.ws == W;  // scale 0: all three terms live on the same map
  const float* G_r = p.G_r[it];
  const float* G_c = it > 0 ? p.G_c[it] : nullptr;
  // the pose-update pass's maps that are gradients w.r.t. THIS iteration's disparity (wave-uniform)
  const bool pa = p.pu_on && p.pu_into[0] == it + 1, pb = p.pu_on && p.pu_into[1] == it + 1;
  for (int row = blockIdx.x; row < B * H; row += gridDim.x) {
    const int b = row / H, y = row - b * H;
    const float inv = div_(1.0f, (float)p.stats[it * B + b] + 1e-7f), corr = (float)p.stats[kDrIt * B + it * B + b];
    const float* brow = p.bnd[it] ? march_boundary_row(p.bnd[it], b, y, H, W, p.rows, p.segs) : nullptr;
    const float* browA = (pa && p.bnd[p.n]) ? march_boundary_row(p.bnd[p.n], b, y, H, W, p.rows, p.segs) : nullptr;
    const float* browB = (pb && p.bnd2) ? march_boundary_row(p.bnd2, b, y, H, W, p.rows, p.segs) : nullptr;
    const size_t r0 = (size_t)row * W;
    for (int x = threadIdx.x; x < W; x += 256) {
      const size_t i = r0 + x;
      const float G = brow ? G_r[i] + brow[x] : G_r[i];
      float v = with_smooth ? cS * (gn[i] * inv - corr) : 0.f;
      if (G_c) v = fma_(cC, G_c[i], v);
      v = fma_(cR, G, v);
      if (pa) v = fma_(cP, browA ? p.G_r[p.n][i] + browA[x] : p.G_r[p.n][i], v);
      if (pb) v = fma_(cP, browB ? p.G_r2[i] + browB[x] : p.G_r2[i], v);
      out[i] = v;
    }
  }
}

}  // namespace mal

using namespace mal;

static int dr_check(const mal_dr_args* a) {
  if (!a) return MAL_EINVAL;
  int rc = check_shape(a->B, a->H, a->W);
  if (rc) return rc;
  if (a->n_iters < 1 || a->n_iters > kDrIt) return MAL_EINVAL;
  if (a->scale < 0 || a->scale > 3) return MAL_EINVAL;
  if (a->scale > 0) {
    const int f = 1 << a->scale;
    if (a->H % f || a->W % f || (a->H >> a->scale) < 2 || (a->W >> a->scale) < 2) return MAL_ESHAPE;
    if (!a->color0_s) return MAL_EINVAL;
    for (int it = 0; it < a->n_iters; ++it)
      if (!a->disp_lo[it]) return MAL_EINVAL;
  }
  if (!a->color0 || !a->color_m1 || !a->color_p1 || !a->K || !a->inv_K || !a->losses || !a->ws) return MAL_EINVAL;
  for (int it = 0; it < a->n_iters; ++it)
    if (!a->disp[it] || !a->T_m1[it] || !a->T_p1[it]) return MAL_EINVAL;
  // the motion mask multiplies the automask of every iteration > 0 (dualrefine/trainer.py:593-597) unless switched off
  if (a->n_iters > 1 && !(a->flags & MAL_DR_NO_MOTION_MASK) && !a->consistency_mask) return MAL_EINVAL;
  if (a->flags & MAL_DR_NOISE_PHILOX)
    for (int it = 0; it < a->n_iters; ++it)
      if (a->noise[it]) return MAL_EINVAL;
  if (a->flags & MAL_DR_POSE_UPDATE) {  // rides on the full-resolution call
    if (a->scale != 0 || !a->pu_disp_m1 || !a->pu_disp_p1 || !a->pu_T_m1 || !a->pu_T_p1) return MAL_EINVAL;
    if ((a->flags & MAL_DR_NOISE_PHILOX) && a->pu_noise) return MAL_EINVAL;
  }
  if (a->ws_bytes < carve_dr(nullptr, a->B, a->H, a->W, a->n_iters).bytes) return MAL_EWORKSPACE;
  return MAL_OK;
}

extern "C" size_t mal_dr_workspace_bytes(int B, int H, int W, int n_iters) {
  if (B <= 0 || H <= 0 || W <= 0 || n_iters < 1 || n_iters > kDrIt) return 0;
  return carve_dr(nullptr, B, H, W, n_iters).bytes;
}

extern "C" int mal_dr_loss_fwd(const mal_dr_args* a) {
  int rc = dr_check(a);
  if (rc) return rc;
  const int B = a->B, H = a->H, W = a->W, n = a->n_iters;
  DrWs w = carve_dr(a->ws, B, H, W, n);
  hipStream_t st = (hipStream_t)a->stream;
  const bool automask = !(a->flags & MAL_DR_NO_AUTOMASK);
  const int pu = (a->flags & MAL_DR_POSE_UPDATE) ? 1 : 0;
  const bool philox = automask && (a->flags & MAL_DR_NOISE_PHILOX);
  // identity term + texel packing (no noise here: every iteration adds its own).  Up to two iterations (the shipped
  // n_losses = 1): their edge-aware smoothness rides on this sweep, which holds the target rows anyway; more: one batched
  // smoothness sweep below
  const int hs = H >> a->scale, wsz = W >> a->scale;
  const bool smooth_fused = n <= 2 && a->scale == 0;
  const int variant = ((a->flags & MAL_DR_NO_SSIM) ? 1 : 0) | ((a->flags & MAL_DR_AVG) ? 2 : 0);
  int per_sample_sm = 1;
  const bool reuse = a->texels_from != nullptr && !smooth_fused;
  if (reuse) {  // texels and identity term of an earlier call of this step (its workspace starts with them, whatever its n_iters)
    const DrWs w0 = carve_dr(const_cast<void*>(a->texels_from), B, H, W, 1);
    for (int i = 0; i < 3; ++i) w.packed[i] = w0.packed[i];
    w.ident = w0.ident;
  } else {
    SmoothParams sm = {};
    sm.n = n; sm.partials = w.sm[0];
    for (int it = 0; it < n && smooth_fused; ++it) { sm.disp[it] = a->disp[it]; sm.gn[it] = w.gn[it]; }
    // ... and the prologue (camera blocks, noise maps: nothing the sweep reads or writes) as the same launch's last workgroups
    DrExtra x = {};
    x.K = a->K; x.invK = a->inv_K; x.seed = a->noise_seed; x.step = a->noise_step;
    x.counter = (const unsigned long long*)a->noise_counter;
    x.B = B; x.H = H; x.W = W; x.ticket = w.ticket; x.slots = n + pu; x.pu_slot = pu ? n : -1;
    x.noise_blocks = philox ? (int)(((size_t)B * ((H + 3) / 4) * W + 63) / 64) : 0;
    for (int it = 0; it < n; ++it) { x.T[it][0] = a->T_m1[it]; x.T[it][1] = a->T_p1[it]; x.cam[it] = w.cam[it]; x.noise[it] = w.noise[it]; }
    if (pu) { x.T[n][0] = a->pu_T_m1; x.T[n][1] = a->pu_T_p1; x.cam[n] = w.cam[n]; x.noise[n] = w.noise[n]; }
    rc = pack_identity_launch(a->color0, a->color_m1, a->color_p1, B, H, W, w.packed[1], w.packed[2], w.packed[0], w.ident, st,
                              nullptr, nullptr, smooth_fused ? &sm : nullptr, &per_sample_sm, false, variant, &x);
    if (rc) return rc;
  }
  if (reuse) {  // (no sweep in this call: the prologue is a launch of its own)
    DrPrologue q = {};
    q.K = a->K; q.invK = a->inv_K; q.seed = a->noise_seed; q.step = a->noise_step;
    q.counter = (const unsigned long long*)a->noise_counter;
    q.B = B; q.H = H; q.W = W; q.ticket = w.ticket;
    q.noise_blocks = philox ? (int)(((size_t)B * ((H + 3) / 4) * W + 255) / 256) : 0;
    for (int it = 0; it < n; ++it) { q.T[it][0] = a->T_m1[it]; q.T[it][1] = a->T_p1[it]; q.cam[it] = w.cam[it]; q.noise[it] = w.noise[it]; }
    q.pu_slot = pu ? n : -1;
    if (pu) { q.T[n][0] = a->pu_T_m1; q.T[n][1] = a->pu_T_p1; q.cam[n] = w.cam[n]; q.noise[n] = w.noise[n]; }
    hipLaunchKernelGGL(dr_prologue_kernel, dim3((unsigned)(q.noise_blocks + B), (unsigned)(n + pu)), dim3(256), 0, st, q);
    rc = launch_status();
    if (rc) return rc;
  }
  const int packed = MAL_F_SRC_PACKED | MAL_F_TGT_PACKED;
  const float merge_cons = (float)(1.0 / ((double)B * H * W));
  int per_sample = 1;
  for (int it = 0; it < n; ++it) {
    MarchParams p = march_params(B, H, W, a->min_depth, a->max_depth, 1e-7f, 1);  // convention B
    p.disp = a->disp[it]; p.K = a->K; p.invK = a->inv_K; p.T[0] = a->T_m1[it]; p.T[1] = a->T_p1[it];
    p.src[0] = w.packed[1]; p.src[1] = w.packed[2]; p.target = w.packed[0];
    p.ident = w.ident; p.noise = automask ? (philox ? w.noise[it] : a->noise[it]) : nullptr;
    p.g_reproj = w.G_r[it]; p.block_sums = w.bs[it]; p.block_gP = w.bgP[it];
    p.bnd = g_march_halo1 ? w.bnd[it] : nullptr;
    p.cam = w.cam[it]; p.cam_ready = 1;  // the poses differ per iteration: its own camera block (dr_prologue_kernel)
    p.avg = (a->flags & MAL_DR_AVG) ? 1 : 0; p.no_ssim = (a->flags & MAL_DR_NO_SSIM) ? 1 : 0;
    int flags = MAL_F_GRAD | MAL_F_POSE_GRAD | (automask ? MAL_F_AUTOMASK : 0) | packed;
    if (it > 0) {  // x consistency_mask, consistency term against iteration 0's depth (its disparity, no gradient to it)
      p.ext_mask = (a->flags & MAL_DR_NO_MOTION_MASK) ? nullptr : a->consistency_mask;
      p.mono_disp = a->disp[0]; p.mono_reproj = w.ident;  // the distillation selection has weight 0: any map serves
      p.g_cons = w.G_c[it]; p.g_distil = nullptr; p.merge_cons = merge_cons; p.merge_distil = 0.f;
      flags |= MAL_F_EPILOGUE;
    }
    p.dbg = a->dec[it];
    rc = march_launch(p, flags, st);
    if (rc) return rc;
    per_sample = p.strips * p.segs;
  }
  if (pu) {
    // the pose-update losses (dualrefine/trainer.py:457-480,699-767): min over {frame -1 under the refined pose with the last
    // iteration's disparity, frame +1 as iteration 0 warped it}, automask with a noise draw of its own, no mask, no epilogue
    MarchParams p = march_params(B, H, W, a->min_depth, a->max_depth, 1e-7f, 1);
    p.disp = a->pu_disp_m1; p.disp2 = a->pu_disp_p1; p.framed = 1;
    p.K = a->K; p.invK = a->inv_K; p.T[0] = a->pu_T_m1; p.T[1] = a->pu_T_p1;
    p.src[0] = w.packed[1]; p.src[1] = w.packed[2]; p.target = w.packed[0];
    p.ident = w.ident; p.noise = automask ? (philox ? w.noise[n] : a->pu_noise) : nullptr;
    p.g_reproj = w.G_r[n]; p.g_reproj2 = w.G_r2; p.block_sums = w.bs[n]; p.block_gP = w.bgP[n];
    p.bnd = g_march_halo1 ? w.bnd[n] : nullptr; p.bnd2 = g_march_halo1 ? w.bnd2 : nullptr;
    p.cam = w.cam[n]; p.cam_ready = 1;
    p.avg = (a->flags & MAL_DR_AVG) ? 1 : 0; p.no_ssim = (a->flags & MAL_DR_NO_SSIM) ? 1 : 0;
    p.dbg = a->pu_dec;
    rc = march_launch(p, MAL_F_GRAD | MAL_F_POSE_GRAD | (automask ? MAL_F_AUTOMASK : 0) | packed, st);
    if (rc) return rc;
    per_sample = p.strips * p.segs;
  }
  DrFinal fin = {};
  if (!smooth_fused) {
    const float *sd[kDrIt], *si[kDrIt];
    float* sg[kDrIt];
    double* sp[kDrIt];
    int sh[kDrIt], sw[kDrIt], per[kDrIt];
    for (int it = 0; it < n; ++it) {
      sd[it] = a->scale ? a->disp_lo[it] : a->disp[it]; si[it] = a->scale ? a->color0_s : a->color0;
      sg[it] = w.gn[it]; sp[it] = w.sm[it]; sh[it] = hs; sw[it] = wsz;
    }
    rc = smooth_march_sweep_batch(n, sd, si, B, sh, sw, sg, sp, st, per);
    if (rc) return rc;
    fin.per_sample_sm = per[0]; fin.sm_stride = 4;
    for (int it = 0; it < n; ++it) fin.sm[it] = w.sm[it];
  } else {
    fin.per_sample_sm = per_sample_sm; fin.sm_stride = 8;   // [task][map][4] of the first sweep
    for (int it = 0; it < n; ++it) fin.sm[it] = w.sm[0] + it * 4;
  }
  for (int it = 0; it < n + pu; ++it) { fin.bs[it] = w.bs[it]; fin.bgP[it] = w.bgP[it]; }
  fin.pu = pu; fin.pu_loss_total = pu ? a->pu_loss_total : nullptr;
  fin.K = a->K; fin.per_sample = per_sample; fin.B = B; fin.H = H; fin.W = W; fin.n = n; fin.smooth_weight = a->smooth_weight;
  fin.hs = hs; fin.ws = wsz;
  fin.ps = w.ps; fin.stats = w.stats; fin.gT = w.gT; fin.losses = a->losses; fin.coefs = w.coefs; fin.loss_total = a->loss_total;
  fin.ticket = w.ticket;
  fin.noise_counter = philox ? (unsigned long long*)a->noise_counter : nullptr;
  hipLaunchKernelGGL(dr_final_kernel, dim3(2 * (n + pu) * B), dim3(256), 0, st, fin);
  return launch_status();
}

extern "C" int mal_dr_loss_bwd(const mal_dr_args* a) {
  int rc = dr_check(a);
  if (rc) return rc;
  const int B = a->B, H = a->H, W = a->W, n = a->n_iters;
  DrWs w = carve_dr(a->ws, B, H, W, n);
  DrAssemble p = {};
  for (int it = 0; it < n; ++it) {
    p.G_r[it] = w.G_r[it]; p.G_c[it] = w.G_c[it]; p.gn[it] = w.gn[it]; p.bnd[it] = g_march_halo1 ? w.bnd[it] : nullptr;
    p.g_disp[it] = a->g_disp[it]; p.g_T[it][0] = a->g_T_m1[it]; p.g_T[it][1] = a->g_T_p1[it];
    p.g_disp_lo[it] = a->scale ? a->g_disp_lo[it] : nullptr;
  }
  const int pu = (a->flags & MAL_DR_POSE_UPDATE) ? 1 : 0;
  if (pu) {
    p.G_r[n] = w.G_r[n]; p.G_r2 = w.G_r2; p.bnd[n] = g_march_halo1 ? w.bnd[n] : nullptr; p.bnd2 = g_march_halo1 ? w.bnd2 : nullptr;
    p.g_disp[n] = a->g_pu_disp_m1; p.g_disp2 = a->g_pu_disp_p1; p.g_T[n][0] = a->g_pu_T_m1; p.g_T[n][1] = a->g_pu_T_p1;
    p.g_pu_total = a->g_pu_total;
    p.pu_on = 1;
    const int into[4] = {a->pu_disp_m1_into, a->pu_disp_p1_into, a->pu_T_m1_into, a->pu_T_p1_into};
    float* const own[4] = {a->g_pu_disp_m1, a->g_pu_disp_p1, a->g_pu_T_m1, a->g_pu_T_p1};
    for (int k = 0; k < 4; ++k) {
      if (into[k] < 0 || into[k] > n || (into[k] && own[k])) return MAL_EINVAL;  // one destination per gradient
      if (into[k] && a->scale) return MAL_EINVAL;
      p.pu_into[k] = into[k];
    }
  }
  p.hs = H >> a->scale; p.ws = W >> a->scale;
  p.gT = w.gT; p.coefs = w.coefs; p.stats = w.stats; p.g_total = a->g_total;
  p.B = B; p.H = H; p.W = W; p.n = n;
  march_geometry(B, H, W, MAL_F_GRAD, nullptr, &p.segs, &p.rows);
  unsigned rowsg = (unsigned)(B * H);
  if (rowsg > 4096) rowsg = 4096;
  hipLaunchKernelGGL(dr_assemble_kernel, dim3(rowsg, (unsigned)(n + pu)), dim3(256), 0, (hipStream_t)a->stream, p);
  return launch_status();
}

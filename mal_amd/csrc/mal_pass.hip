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
#include "experiments/pass_tiled.inc"
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
#include "experiments/pass_tiled_launch.inc"
#endif  // MAL_EXPERIMENTS

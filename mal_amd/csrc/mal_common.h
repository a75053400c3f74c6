// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <atomic>
#include "../../include/mal_hip.h"

namespace mal {
// A process-wide option (mal_set_option): written under a mutex, read by the launch functions of any thread -- an atomic, so
// that a concurrent mal_set_option is a defined (if ill-advised) thing to do.  They are same-box A/B switches, not per-call
// state: set them before the first launch (a step reads some of them in both of its calls).
typedef std::atomic<int> opt_t;

// tile geometry of the fused / photometric kernels: 64x16 output pixels per 256-thread
// workgroup; one wave owns one 64-pixel row segment per step, so every global access of the
// (B,C,H,W) planes is a 256-byte coalesced row piece.
constexpr int kTW = 64;
constexpr int kTH = 16;
constexpr int kThreads = 256;

struct TileGrid {
  int tiles_x, tiles_y, B;
  int blocks() const { return tiles_x * tiles_y * B; }
};

inline TileGrid tile_grid(int B, int H, int W) {
  TileGrid g;
  g.tiles_x = (W + kTW - 1) / kTW;
  g.tiles_y = (H + kTH - 1) / kTH;
  g.B = B;
  return g;
}

// workspace layout (all offsets 256-byte aligned):
//   [0]            double block_sums[blocks][8]
//   [..]           float  block_gP[blocks][2][12]
//   [..]           double scratch[4096]         (generic two-stage reductions)
//   [..]           float  cam[B][40]            (marching kernel: P of both frames + inv_K per sample)
struct Workspace {
  double* block_sums;
  float* block_gP;
  double* scratch;
  float* cam;
  float* bnd;   // boundary scratch rows of a marching gradient pass with the one-row halo (mal_march.h), [B][H/8+1][2][W]
  size_t bytes;
};
// the fold of those scratch rows into the pass's gradient map, done by extra workgroups of pass_finalize_kernel
struct BoundaryFold { const float* bnd; float* g; int B, H, W, rows, segs; };

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

// capacity of the per-workgroup partial arrays: the marching kernel's 60-column strips x 8-row
// segments (>= the 64x16 tiles of the LDS-tiled kernel and the 1024-pixel chunks of the others)
inline size_t ws_blocks(int B, int H, int W) {
  return (size_t)B * (size_t)((W + 59) / 60) * (size_t)((H + 7) / 8);
}

inline Workspace carve(void* base, int B, int H, int W) {
  size_t nb = ws_blocks(B, H, W);
  size_t o = 0;
  Workspace w;
  char* p = (char*)base;
  w.block_sums = (double*)(p + o); o += align256(nb * 8 * sizeof(double));
  w.block_gP = (float*)(p + o);    o += align256(nb * 24 * sizeof(float));
  w.scratch = (double*)(p + o);    o += align256(4096 * sizeof(double));
  w.cam = (float*)(p + o);         o += align256((size_t)B * 40 * sizeof(float));
  w.bnd = (float*)(p + o);         o += align256((size_t)B * ((size_t)H / 8 + 1) * 2 * (size_t)W * sizeof(float));
  w.bytes = o;
  return w;
}

inline int check_shape(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return MAL_EINVAL;
  if (H < 2 || W < 2) return MAL_ESHAPE;
  if ((double)B * 4.0 * (double)H * (double)W > 2.0e9) return MAL_ESHAPE;  // int32 indexing
  return MAL_OK;
}

// mal_pass.hip: fixed-order second stage shared by both fused-pass formulations
int launch_pass_finalize(const double* block_sums, const float* block_gP, const float* K, int nblocks,
                         int blocks_per_sample, int B, double* sums, float* gT0, float* gT1, hipStream_t st,
                         int nsums = 4, const BoundaryFold* fold = nullptr);

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MAL_OK : MAL_ELAUNCH;
}

}  // namespace mal

// a16: transformation_from_parameters (manydepth/layers.py:26-100) as ONE launch for all
// frames and samples, forward and backward.  Upstream this is ~40 tiny ATen launches per pose
// (and as many again in autograd's backward); on a HIP-graph'ed step those launches cost more
// device time than the fused photometric kernel.
//
//   angle = |v|, axis = v / (angle + 1e-7), ca = cos, sa = sin, C = 1 - ca
//   R = Rodrigues(axis, angle)                                     (layers.py:61-100)
//   M = T(t) R  = [R t; 0 1]            (invert = 0)                (layers.py:38-41)
//   M = R^T T(-t) = [R^T, -R^T t; 0 1]  (invert = 1)
#include "mal_pose.h"

namespace mal {

__global__ void pose_fwd_kernel(PoseParams p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.B * p.F) return;
  pose_fwd_one(p, i / p.B, i % p.B);
}

__global__ void pose_bwd_kernel(PoseParams p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.B * p.F) return;
  pose_bwd_one(p, i / p.B, i % p.B);
}

}  // namespace mal

using namespace mal;

static int fill_pose(PoseParams& p, const float* const* axisangle, const float* const* translation, const int* invert,
                     int B, int F) {
  if (B <= 0 || F < 1 || F > MAL_MAX_FRAMES || !axisangle || !translation || !invert) return MAL_EINVAL;
  p.B = B; p.F = F;
  for (int f = 0; f < MAL_MAX_FRAMES; ++f) {
    p.axisangle[f] = p.translation[f] = p.gT[f] = nullptr;
    p.T[f] = p.g_axisangle[f] = p.g_translation[f] = nullptr;
    p.invert[f] = 0;
  }
  for (int f = 0; f < F; ++f) {
    if (!axisangle[f] || !translation[f]) return MAL_EINVAL;
    p.axisangle[f] = axisangle[f]; p.translation[f] = translation[f]; p.invert[f] = invert[f];
  }
  return MAL_OK;
}

extern "C" int mal_pose_fwd(const float* const* axisangle, const float* const* translation, const int* invert, int B,
                            int F, float* const* T, void* stream) {
  PoseParams p;
  int rc = fill_pose(p, axisangle, translation, invert, B, F);
  if (rc) return rc;
  if (!T) return MAL_EINVAL;
  for (int f = 0; f < F; ++f) { if (!T[f]) return MAL_EINVAL; p.T[f] = T[f]; }
  hipLaunchKernelGGL(pose_fwd_kernel, dim3((B * F + 63) / 64), dim3(64), 0, (hipStream_t)stream, p);
  return launch_status();
}

extern "C" int mal_pose_bwd(const float* const* axisangle, const float* const* translation, const int* invert,
                            const float* const* g_T, int B, int F, float* const* g_axisangle,
                            float* const* g_translation, void* stream) {
  PoseParams p;
  int rc = fill_pose(p, axisangle, translation, invert, B, F);
  if (rc) return rc;
  if (!g_T || !g_axisangle || !g_translation) return MAL_EINVAL;
  for (int f = 0; f < F; ++f) {
    if (!g_T[f]) return MAL_EINVAL;
    p.gT[f] = g_T[f]; p.g_axisangle[f] = g_axisangle[f]; p.g_translation[f] = g_translation[f];
  }
  hipLaunchKernelGGL(pose_bwd_kernel, dim3((B * F + 63) / 64), dim3(64), 0, (hipStream_t)stream, p);
  return launch_status();
}

// a16: transformation_from_parameters (manydepth/layers.py:26-100), forward and backward, as device
// functions of one (frame, sample) pose -- shared by the stand-alone kernels (mal_pose.hip) and the
// whole-step launch list (mal_step.hip).
#pragma once
#include "mal_common.h"
#include "mal_device.h"

namespace mal {

struct PoseParams {
  const float* axisangle[MAL_MAX_FRAMES];    // (B,3) each (the reference's (B,1,3), contiguous)
  const float* translation[MAL_MAX_FRAMES];  // (B,3)
  int invert[MAL_MAX_FRAMES];
  float* T[MAL_MAX_FRAMES];                  // (B,16) out
  const float* gT[MAL_MAX_FRAMES];           // (B,16) in   (backward)
  float* g_axisangle[MAL_MAX_FRAMES];        // (B,3) out   (backward)
  float* g_translation[MAL_MAX_FRAMES];
  int B, F;
};

struct Rod { float x, y, z, ca, sa, C, angle; float R[9]; };

MAL_DEV Rod rodrigues(const float* v) {
  Rod r;
  r.angle = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float inv = div_safe_(1.0f, r.angle + 1e-7f);
  r.x = v[0] * inv; r.y = v[1] * inv; r.z = v[2] * inv;
  r.ca = cosf(r.angle); r.sa = sinf(r.angle); r.C = 1.0f - r.ca;
  const float xs = r.x * r.sa, ys = r.y * r.sa, zs = r.z * r.sa;
  const float xC = r.x * r.C, yC = r.y * r.C, zC = r.z * r.C;
  const float xyC = r.x * yC, yzC = r.y * zC, zxC = r.z * xC;
  r.R[0] = r.x * xC + r.ca; r.R[1] = xyC - zs;        r.R[2] = zxC + ys;
  r.R[3] = xyC + zs;        r.R[4] = r.y * yC + r.ca; r.R[5] = yzC - xs;
  r.R[6] = zxC - ys;        r.R[7] = yzC + xs;        r.R[8] = r.z * zC + r.ca;
  return r;
}

MAL_DEV void pose_fwd_one(const PoseParams& p, int f, int b) {
  const float* v = p.axisangle[f] + b * 3;
  const float* t = p.translation[f] + b * 3;
  const Rod r = rodrigues(v);
  float* M = p.T[f] + b * 16;
  if (!p.invert[f]) {
    for (int a = 0; a < 3; ++a) {
      for (int c = 0; c < 3; ++c) M[a * 4 + c] = r.R[a * 3 + c];
      M[a * 4 + 3] = t[a];
    }
  } else {
    for (int a = 0; a < 3; ++a) {
      for (int c = 0; c < 3; ++c) M[a * 4 + c] = r.R[c * 3 + a];
      // row a of R^T times (-t), accumulated in k order like the 4x4 matmul upstream
      float acc = r.R[0 * 3 + a] * (-t[0]);
      acc = fma_(r.R[1 * 3 + a], -t[1], acc);
      acc = fma_(r.R[2 * 3 + a], -t[2], acc);
      M[a * 4 + 3] = acc;
    }
  }
  M[12] = 0.f; M[13] = 0.f; M[14] = 0.f; M[15] = 1.f;
}

MAL_DEV void pose_bwd_one(const PoseParams& p, int f, int b) {
  const float* v = p.axisangle[f] + b * 3;
  const float* t = p.translation[f] + b * 3;
  const float* g = p.gT[f] + b * 16;
  const Rod r = rodrigues(v);
  float gR[9], gt[3];
  if (!p.invert[f]) {
    for (int a = 0; a < 3; ++a) {
      for (int c = 0; c < 3; ++c) gR[a * 3 + c] = g[a * 4 + c];
      gt[a] = g[a * 4 + 3];
    }
  } else {
    // M[:3,:3] = N = R^T ; M[:3,3] = N (-t)
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) gR[c * 3 + a] = g[a * 4 + c] + g[a * 4 + 3] * (-t[c]);  // gN[a][c] -> gR[c][a]
    for (int c = 0; c < 3; ++c)
      gt[c] = -(r.R[c * 3 + 0] * g[0 * 4 + 3] + r.R[c * 3 + 1] * g[1 * 4 + 3] + r.R[c * 3 + 2] * g[2 * 4 + 3]);
  }
  const float x = r.x, y = r.y, z = r.z, s = r.sa, C = r.C;
  // R as a function of (x, y, z, ca, sa) with C = 1 - ca
  const float gx = gR[0] * 2.f * x * C + (gR[1] + gR[3]) * y * C + (gR[2] + gR[6]) * z * C + (gR[7] - gR[5]) * s;
  const float gy = gR[4] * 2.f * y * C + (gR[1] + gR[3]) * x * C + (gR[5] + gR[7]) * z * C + (gR[2] - gR[6]) * s;
  const float gz = gR[8] * 2.f * z * C + (gR[2] + gR[6]) * x * C + (gR[5] + gR[7]) * y * C + (gR[3] - gR[1]) * s;
  const float gC = gR[0] * x * x + gR[4] * y * y + gR[8] * z * z + (gR[1] + gR[3]) * x * y + (gR[2] + gR[6]) * z * x +
                   (gR[5] + gR[7]) * y * z;
  const float gca = (gR[0] + gR[4] + gR[8]) - gC;
  const float gsa = (gR[3] - gR[1]) * z + (gR[2] - gR[6]) * y + (gR[7] - gR[5]) * x;
  const float g_angle_trig = -gca * r.sa + gsa * r.ca;
  // axis = v / (angle + eps)
  const float d = r.angle + 1e-7f;
  const float inv = div_safe_(1.0f, d);
  const float dot = gx * v[0] + gy * v[1] + gz * v[2];
  const float g_angle = g_angle_trig - dot * inv * inv;
  const float inva = r.angle > 0.f ? div_safe_(1.0f, r.angle) : 0.f;  // d|v|/dv = v/|v| (0 at 0, as torch.norm)
  float* ga = p.g_axisangle[f] ? p.g_axisangle[f] + b * 3 : nullptr;
  float* gtr = p.g_translation[f] ? p.g_translation[f] + b * 3 : nullptr;
  const float gv[3] = {gx * inv + g_angle * v[0] * inva, gy * inv + g_angle * v[1] * inva,
                       gz * inv + g_angle * v[2] * inva};
  for (int a = 0; a < 3; ++a) {
    if (ga) ga[a] = gv[a];
    if (gtr) gtr[a] = gt[a];
  }
}

}  // namespace mal

// mal_loss_multiscale_fwd / _bwd: the loss half of process_batch without --distil and with sclm > 0
// (manydepth/trainer.py:573-612 -> compute_losses, :1248-1475, for the teacher and for the student; generate_images_pred,
// :1078-1170, upsamples every scale's disparity to full resolution and warps the full-resolution sources with it) as one
// host call per direction.
//
// The operator-level route issues, per step at sclm = 3, 16 fused passes + 8 identity terms + 8 smoothness operators and
// ~250 ATen launches of glue between them (F.interpolate and its atomics-scattered backward, the per-scale scalar
// arithmetic, autograd's accumulation): 1.8 ms at B=12 192x640.  Here:
//
//   forward  1 first sweep (identity term, texel packing, poses, camera block)           shared by all scales
//            1 noise launch (Philox, one map per scale)   1 upsampling launch (all scales, both networks)
//            per scale: teacher pass (fwd + gradient), student pass (fwd + gradient + consistency), 2 smoothness sweeps
//            1 reduction: per-sample sums, pose gradients per scale, loss scalars, coefficients of the backward
//   backward 1 launch: per (network, scale) the adjoint of the upsampling as a GATHER in a fixed order (ATen scatters
//            with atomics) of the weighted gradient maps + the smoothness gradient; pose backward in its first block
#include "mal_march.h"
#include "mal_device.h"
#include "mal_pose.h"

namespace mal {
extern opt_t g_march_halo1;  // mal_step.hip

// mal_photo_march.hip / mal_step.hip
int smooth_march_sweep_batch(int n, const float* const* disp, const float* const* img, int B, const int* H, const int* W,
                             float* const* gn, double* const* partial, hipStream_t st, int* per_sample);
int photo_march_fused_more(const float* target, const float* cand0, const float* cand1, int idx0, const float* ident,
                           const float* noise, const float* prev_min, const uint8_t* prev_arg, int B, int H, int W,
                           float* min_reproj, uint8_t* argmin, float* weight_out, double* block_sums, float* g_cand0,
                           float* g_cand1, int* per_sample_out, hipStream_t st, const uint8_t* region, float* g_region0,
                           float* g_region1, unsigned* order, unsigned* order_count, const float* orig0, const float* orig1,
                           size_t orig_stride, int target_texels, int weight_given);
int smooth_march_batch_tasks(int H, int W);
// mal_step.hip: the library's side stream of the caller's stream (fork / join through events: capturable)
hipStream_t side_begin(hipStream_t st);
int side_end(hipStream_t st);
int side_wait(hipStream_t st, bool always);
int tiebreak_noise_launch(unsigned long long seed, unsigned long long step, const unsigned long long* counter, unsigned mult,
                          int n, int B, int H, int W, float* const* out, hipStream_t st);

constexpr int kMsS = MAL_MS_MAX_SCALES;
// option "ms_fold" (A/B): 0 = the noise maps and the upsampling as launches of their own behind the first sweep.  Worth 0.6 %
// only (0.5937 -> 0.5903 ms; the noise maps alone 0.5888, the upsampling alone 0.5914): the riders inherit the sweep's 152
// VGPRs per wave, i.e. three waves per SIMD for 160 000 tiny workgroups that as launches of their own run at eight
opt_t g_ms_fold{1};
constexpr int kMsLossSlots = 48;

struct MsWs {
  float* packed[3]; float* T[2]; float* gTs[2]; float* cam; unsigned* ticket;
  float* ident; float* cmask;
  float* noise[kMsS];
  float* up[2][kMsS];    // disparities of scale s >= 1 at full resolution (net 0 teacher, 1 student)
  float* G_r[2][kMsS];   // d (sum rp*w) / d upsampled disparity, unnormalised
  float* bnd[2][kMsS];   // boundary scratch rows of that pass (one-row halo, mal_march.h), folded into G_r by ms_fold_kernel
  float* G_c[kMsS];      // d consistency_s / d upsampled student disparity, already weighted by 1/((sclm+1) N)
  float* gn[2][kMsS];    // smoothness: d / d normalised disparity at the scale's own size
  double* bs[2][kMsS]; float* bgP[kMsS]; double* sm[2][kMsS];
  double* ps;            // [2S][B][8]
  double* stats;         // [2][2S][B]: mean, mean-coupling term
  float* gT;             // [S][2][B*16]
  float* coefs;          // [3S]: cR[net*S + s], cS[s]
  float* neg_inf;        // MAL_STEP_ENSEMBLE: a map of -inf -- handed to the student's epilogue as the "ensemble reprojection", it makes
                         // the ensemble win the distillation argmin at every pixel: that term IS upstream's ensemble loss then
  // temporal hint, per scale: what the teacher's forward pass in front of the producer leaves (min over the two warped candidates,
  // its winner), the four-way decision of the fused sweep (winner, automask weight, min) and that sweep's per-task differences
  float* rp_warp[kMsS]; unsigned char* arg_warp[kMsS]; float* rp4[kMsS]; unsigned char* arg_t[kMsS]; float* w_t[kMsS];
  double* bs_ph[kMsS];
  size_t bytes;
};

static MsWs carve_ms(void* base, int B, int H, int W, int sclm) {
  MsWs w = {};
  char* p = (char*)base;
  size_t o = 0;
  const int S = sclm + 1;
  const size_t HW = (size_t)H * W, map = B * HW * sizeof(float), nb = ws_blocks(B, H, W);
  auto take = [&](size_t bytes) { char* r = p + o; o += align256(bytes); return r; };
  for (int i = 0; i < 3; ++i) w.packed[i] = (float*)take(B * HW * kTexel * sizeof(float));
  for (int f = 0; f < 2; ++f) { w.T[f] = (float*)take(B * 16 * 4); w.gTs[f] = (float*)take(B * 16 * 4); }
  w.cam = (float*)take((size_t)B * kCamFloats * 4);
  w.ticket = (unsigned*)take(16);  // [1]: reset by the shared first launch too (the one-call step's task queue)
  w.ident = (float*)take(map); w.cmask = (float*)take(map);
  for (int s = 0; s < S; ++s) {
    w.noise[s] = (float*)take(map);
    for (int n = 0; n < 2; ++n) {
      w.up[n][s] = s ? (float*)take(map) : nullptr;
      w.G_r[n][s] = (float*)take(map);
      w.bnd[n][s] = (float*)take(march_bnd_floats(B, H, W) * sizeof(float));
      w.gn[n][s] = (float*)take(map >> (2 * s));
      w.bs[n][s] = (double*)take(nb * 8 * 8);
      w.sm[n][s] = (double*)take(nb * 4 * 8);
    }
    w.G_c[s] = (float*)take(map);
    w.bgP[s] = (float*)take(nb * 24 * 4);
  }
  w.ps = (double*)take((size_t)2 * S * B * 8 * 8);
  w.stats = (double*)take((size_t)2 * 2 * S * B * 8);
  w.gT = (float*)take((size_t)S * 2 * B * 16 * 4);
  w.coefs = (float*)take(3 * kMsS * 4);
  w.neg_inf = (float*)take(map);
  for (int s = 0; s < S; ++s) {
    w.rp_warp[s] = (float*)take(map); w.rp4[s] = (float*)take(map); w.w_t[s] = (float*)take(map);
    w.arg_warp[s] = (unsigned char*)take((size_t)B * HW); w.arg_t[s] = (unsigned char*)take((size_t)B * HW);
    w.bs_ph[s] = (double*)take(nb * 4 * 2 * 8);  // tasks of >= 2 rows, as the one-call step's
  }
  w.bytes = o;
  return w;
}

// ---------------------------------------------------------------- bilinear upsampling (align_corners=False: Tap, tap_of,
// bilinear_value, UpMaps, upsample_pixels in mal_march.h) + adjoint
// weight with which output index `dst` reads input index `i`
MAL_DEV float tap_weight(int dst, float scale, int n_in, int i) {
  const Tap t = tap_of(dst, scale, n_in);
  return (t.i0 == i ? t.l0 : 0.f) + (t.i1 == i ? t.l1 : 0.f);
}

// grid (x chunks, B*H rows, maps); one thread per V consecutive output pixels of a row (V = 4 when W % 4 == 0: one 16-byte
// store)
template <int V>
__global__ __launch_bounds__(256) void upsample_kernel(UpMaps m, int B, int H, int W) {
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * V;
  if (x0 >= W) return;
  upsample_pixels<V>(m, blockIdx.z, blockIdx.y, x0, H, W);
}
static void upsample_launch(const UpMaps& m, int n, int B, int H, int W, hipStream_t st) {
  bool vec = W % 4 == 0;
  for (int k = 0; k < n; ++k) vec = vec && ((uintptr_t)m.dst[k] % 16 == 0);
  if (vec) hipLaunchKernelGGL(upsample_kernel<4>, dim3((W / 4 + 255) / 256, B * H, n), dim3(256), 0, st, m, B, H, W);
  else hipLaunchKernelGGL(upsample_kernel<1>, dim3((W + 255) / 256, B * H, n), dim3(256), 0, st, m, B, H, W);
}

// Adjoint as a gather: input pixel (i, j) of a map upsampled by f = H / h (a power of two <= 8) is read by the output
// rows [f i - f/2, f i + 3f/2) and the same span of columns.  `f` adjacent lanes share one input pixel: lane k takes the
// columns f j - f/2 + k and + f of every row of the span, the f partial sums are added by xor-shuffles (fixed order).
// value gathered: ca * A + cb * Bm (Bm nullable).
template <int F>
MAL_DEV float adjoint_gather_f(const float* __restrict__ A, float ca, const float* __restrict__ Bm, float cb, int i, int j,
                               int k, int h, int w, int H, int W) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const int X0 = F * j - (F >> 1) + k, X1 = X0 + F;
  const bool in0 = X0 >= 0 && X0 < W, in1 = X1 >= 0 && X1 < W;
  const float wx0 = in0 ? tap_weight(X0, sx, w, j) : 0.f, wx1 = in1 ? tap_weight(X1, sx, w, j) : 0.f;
  const int c0 = in0 ? X0 : 0, c1 = in1 ? X1 : 0;
  const int Y0 = F * i - (F >> 1);
  // every load of the span issued before the first use: rows outside the map are clamped and carry weight 0
  float va[2 * F][2], vb[2 * F][2];
#pragma unroll
  for (int dy = 0; dy < 2 * F; ++dy) {
    const int Yc = min(max(Y0 + dy, 0), H - 1);
    va[dy][0] = A[Yc * W + c0]; va[dy][1] = A[Yc * W + c1];
    if (Bm) { vb[dy][0] = Bm[Yc * W + c0]; vb[dy][1] = Bm[Yc * W + c1]; }
  }
  float acc = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2 * F; ++dy) {
    const int Y = Y0 + dy;
    const float wy = (Y >= 0 && Y < H) ? tap_weight(Y, sy, h, i) : 0.f;
    float v0 = ca * va[dy][0], v1 = ca * va[dy][1];
    if (Bm) { v0 = fma_(cb, vb[dy][0], v0); v1 = fma_(cb, vb[dy][1], v1); }
    acc += wy * (wx0 * v0 + wx1 * v1);
  }
#pragma unroll
  for (int d = 1; d < F; d <<= 1) acc += __shfl_xor(acc, d);
  return acc;
}
MAL_DEV float adjoint_gather(const float* __restrict__ A, float ca, const float* __restrict__ Bm, float cb, int i, int j,
                             int k, int f, int h, int w, int H, int W) {
  switch (f) {  // uniform per launch / per blockIdx.y
    case 1: return adjoint_gather_f<1>(A, ca, Bm, cb, i, j, k, h, w, H, W);
    case 2: return adjoint_gather_f<2>(A, ca, Bm, cb, i, j, k, h, w, H, W);
    case 4: return adjoint_gather_f<4>(A, ca, Bm, cb, i, j, k, h, w, H, W);
    default: return adjoint_gather_f<8>(A, ca, Bm, cb, i, j, k, h, w, H, W);
  }
}

__global__ __launch_bounds__(256) void upsample_adjoint_kernel(const float* g_out, int B, int h, int w, int H, int W, float* g_x) {
  const int f = H / h;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t q = t / f;
  const int k = (int)(t - q * f);
  const bool live = q < (size_t)B * h * w;
  const size_t qq = live ? q : 0;
  const int b = (int)(qq / ((size_t)h * w)), pix = (int)(qq - (size_t)b * h * w), i = pix / w, j = pix - i * w;
  const float v = adjoint_gather(g_out + (size_t)b * H * W, 1.0f, nullptr, 0.f, i, j, k, f, h, w, H, W);
  if (live && k == 0) g_x[q] = v;
}

// ---------------------------------------------------------------- reduction + scalars
struct MsFinal {
  const double* bs[2][kMsS]; const double* sm[2][kMsS]; const float* bgP[kMsS]; int per_sample_sm[kMsS];
  // temporal hint: the teacher's sums are those of its forward pass (per_sample_t tasks) + the fused sweep's differences; its
  // pose partials do not exist yet (bgP all null: no pose blocks in this launch, ms_pose_kernel follows the backward's sweeps)
  const double* bs_ph[kMsS]; int per_sample_ph[kMsS]; int per_sample_t;
  const float* K; int per_sample, B, H, W, S;
  int ensemble;          // MAL_STEP_ENSEMBLE: slot 3 of the student's sums is the ensemble term
  double* ps; double* stats; float* gT; float* losses; float* coefs; float* loss_total; unsigned* ticket;
  unsigned long long* noise_counter;
};

// g_T[s][f][b] = K_b^T [gP ; 0] from the teacher's per-task pose partials of scale s (one workgroup of 256)
MAL_DEV void ms_pose_block(const float* bgP, int per_sample, const float* K, float* gT, int s, int b, int B, double* s_part,
                           double* s_gP) {
  const int tid = threadIdx.x;
  const int v = tid % 24, sub = tid / 24;
  double acc = 0.0;
  if (sub < 10) {
#pragma unroll 8
    for (int t = sub; t < per_sample; t += 10) acc += (double)bgP[((size_t)b * per_sample + t) * 24 + v];
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
    const float* Kb = K + b * 16;
    double a = 0.0;
    for (int i = 0; i < 3; ++i) a += (double)Kb[i * 4 + k] * s_gP[f * 12 + i * 4 + j];
    gT[((size_t)(s * 2 + f) * B + b) * 16 + e] = (float)a;
  }
}
struct MsPose { const float* bgP[kMsS]; const float* K; float* gT; int per_sample, B; };
// temporal hint: the teacher's gradient sweeps run in the backward call; S*B workgroups reduce their pose partials there
__global__ __launch_bounds__(256) void ms_pose_kernel(MsPose p) {
  __shared__ double s_part[256];
  __shared__ double s_gP[24];
  const int s = blockIdx.x / p.B, b = blockIdx.x - s * p.B;
  ms_pose_block(p.bgP[s], p.per_sample, p.K, p.gT, s, b, p.B, s_part, s_gP);
}

// 2S*B blocks: ps[pass][b][j], pass = net*S + s: j < 4 from the marching pass's partials, j >= 4 the smoothness sweep's;
// S*B blocks: g_T[s][f][b] = K_b^T [gP ; 0] from the teacher's per-task pose partials of scale s;
// the block that finishes last: loss scalars (trainer.py:1447-1475) and the coefficients of the backward.
__global__ __launch_bounds__(256) void ms_final_kernel(MsFinal p) {
  __shared__ double s_part[256];
  __shared__ double s_gP[24];
  __shared__ double sh_tot[2 * kMsS][8];
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, B = p.B, S = p.S;
  if ((int)blockIdx.x < 2 * S * B) {
    const int pass = blockIdx.x / B, b = blockIdx.x - pass * B, net = pass / S, s = pass - net * S;
    const int j = tid & 7, sub = tid >> 3;
    const int ps_n = net ? p.per_sample : p.per_sample_t;
    const int n_t = j < 4 ? ps_n : p.per_sample_sm[s];
    const size_t stride = j < 4 ? 8 : 4;
    const double* q = j < 4 ? p.bs[net][s] + (size_t)b * ps_n * 8 + j
                            : p.sm[net][s] + (size_t)b * p.per_sample_sm[s] * 4 + (j - 4);
    double acc = 0.0;
#pragma unroll 8
    for (int t = sub; t < n_t; t += 32) acc += q[(size_t)t * stride];
    if (net == 0 && p.bs_ph[s] && j < 2) {
      const double* bd = p.bs_ph[s] + (size_t)b * p.per_sample_ph[s] * 2 + j;
#pragma unroll 8
      for (int t = sub; t < p.per_sample_ph[s]; t += 32) acc += bd[(size_t)t * 2];
    }
    s_part[tid] = acc;
    __syncthreads();
    if (tid < 8) {
      double a = 0.0;
      for (int k = 0; k < 32; ++k) a += s_part[k * 8 + tid];
      p.ps[((size_t)pass * B + b) * 8 + tid] = a;
    }
  } else {
    const int r = blockIdx.x - 2 * S * B, s = r / B, b = r - s * B;
    ms_pose_block(p.bgP[s], p.per_sample, p.K, p.gT, s, b, B, s_part, s_gP);
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(p.ticket, 1u);
  __syncthreads();
  if (s_last != gridDim.x - 1) return;
  __threadfence();
  // per-sample statistics of the smoothness gradient (mean of the scale's own disparity map)
  for (int i = tid; i < 2 * S * B; i += 256) {
    const int s = (i / B) % S;
    const double hw = (double)((p.H >> s) * (p.W >> s));
    const double* q = p.ps + (size_t)i * 8;
    const double mean = q[7] / hw;
    const double m = (double)((float)mean + 1e-7f);
    p.stats[i] = mean;
    p.stats[2 * S * B + i] = q[6] / (hw * m * m);
  }
  // sums over samples in sample order; the smoothness slots weighted by the sample's 1/(mean + 1e-7).  One thread per
  // (pass, sample, slot) fetches and weighs its term (one round trip for all), 8 threads per pass then add B terms.
  constexpr int kStage = 2 * kMsS * 16 * 8;  // staged in LDS up to B = 16; beyond, the summing threads re-fetch
  __shared__ double s_term[kStage];
  const bool staged = 2 * S * B * 8 <= kStage;
  for (int i = tid; staged && i < 2 * S * B * 8; i += 256) {
    const int j = i & 7, s = ((i >> 3) / B) % S;
    const double* q = p.ps + (size_t)(i >> 3) * 8;
    double v = q[j];
    if (j == 4 || j == 5) v = v * (double)div_(1.0f, (float)(q[7] / (double)((p.H >> s) * (p.W >> s))) + 1e-7f);
    s_term[i] = v;
  }
  __syncthreads();
  if (tid < 2 * S * 8) {
    const int pass = tid >> 3, j = tid & 7, s = pass % S;
    const double hw = (double)((p.H >> s) * (p.W >> s));
    double a = 0.0;
    for (int b = 0; b < B; ++b) {
      double v;
      if (staged) v = s_term[((size_t)pass * B + b) * 8 + j];
      else {
        const double* q = p.ps + ((size_t)pass * B + b) * 8;
        v = q[j];
        if (j == 4 || j == 5) v = v * (double)div_(1.0f, (float)(q[7] / hw) + 1e-7f);
      }
      a += v;
    }
    sh_tot[pass][j] = a;
  }
  __syncthreads();
  // one thread per (network, scale): its four scalars and its coefficient (double divisions: not on one thread in a row)
  __shared__ float s_loss[2 * kMsS], s_rep[2 * kMsS], s_ens[2 * kMsS];
  if (tid < 2 * S) {
    const int net = tid / S, s = tid - net * S;
    const double N = (double)B * p.H * p.W;
    const double* t = sh_tot[tid];
    const int h = p.H >> s, w = p.W >> s;
    const double Nx = (double)B * h * (w - 1), Ny = (double)B * (h - 1) * w;
    const float reproj = (float)(t[0] / (t[1] + 1e-7));
    const float cons = net ? (float)(t[2] / N) : 0.f;
    const float ens = (net && p.ensemble) ? (float)(t[3] / N) : 0.f;
    const float smooth = (float)(t[4] / Nx + t[5] / Ny);
    const float loss = (net ? (reproj + cons) + ens : reproj) + (1e-3f * smooth) / (float)(1 << s);
    float* o = p.losses + (net * kMsS + s) * 4;
    o[0] = reproj; o[1] = cons; o[2] = smooth; o[3] = loss;
    s_loss[tid] = loss; s_rep[tid] = reproj; s_ens[tid] = ens;
    p.coefs[tid] = (float)(1.0 / ((double)S * (t[1] + 1e-7)));
    if (net == 0) p.coefs[2 * S + s] = (float)(1e-3 / ((double)(1 << s) * (double)S));
  }
  __syncthreads();
  if (tid != 0) return;
  if (p.noise_counter) *p.noise_counter += 1ull;
  float total[2] = {0.f, 0.f};
  for (int net = 0; net < 2; ++net)
    for (int s = 0; s < S; ++s) total[net] += s_loss[net * S + s];
  for (int net = 0; net < 2; ++net)
    for (int s = S; s < kMsS; ++s)
      for (int k = 0; k < 4; ++k) p.losses[(net * kMsS + s) * 4 + k] = 0.f;
  p.losses[32] = div_(total[0], (float)S);
  p.losses[33] = div_(total[1], (float)S);
  p.losses[34] = p.losses[33] + p.losses[32];
  p.losses[35] = 0.f;
  for (int s = 0; s < kMsS; ++s) {  // process_batch adds the teacher's entries to the student's (trainer.py:614-616)
    p.losses[36 + s] = s < S ? s_rep[S + s] + s_rep[s] : 0.f;
    p.losses[40 + s] = s < S ? s_loss[S + s] + s_loss[s] : 0.f;
  }
  static_assert(kMsLossSlots == 44 + kMsS, "losses[]: the ensemble terms are its last kMsS slots");
  for (int s = 0; s < kMsS; ++s) p.losses[44 + s] = s < S ? s_ens[S + s] : 0.f;
  if (p.loss_total) *p.loss_total = p.losses[34];
}

// One-row halo of the marching gradient passes (MarchParams::bnd): blockIdx.y = pass, blockIdx.x = (sample, segment, first /
// last row); the scratch row is added to the pass's gradient map in place (one add per element, fixed order).
struct MsFold { float* G[2 * kMsS]; const float* bnd[2 * kMsS]; int B, H, W, rows, segs; };
__global__ __launch_bounds__(256) void ms_fold_kernel(MsFold p) {
  const int which = blockIdx.x & 1, bs = blockIdx.x >> 1, seg = bs % p.segs, b = bs / p.segs;
  if ((which == 0 && seg == 0) || (which == 1 && seg == p.segs - 1)) return;  // the image's own border: nobody beyond it
  const int y = which == 0 ? seg * p.rows : min(seg * p.rows + p.rows, p.H) - 1;
  float* g = p.G[blockIdx.y] + ((size_t)b * p.H + y) * p.W;
  const float* r = p.bnd[blockIdx.y] + ((size_t)(b * p.segs + seg) * 2 + which) * p.W;
  for (int x = threadIdx.x; x < p.W; x += 256) g[x] += r[x];
}


// ---------------------------------------------------------------- backward
struct MsAssemble {
  const float* G_r[2][kMsS]; const float* G_c[kMsS]; const float* gn[2][kMsS]; float* g_disp[2][kMsS];
  const float* coefs; const double* stats; const float* g_total; const float* gT;
  float* gTs0; float* gTs1;
  int B, H, W, S;
  unsigned first[2 * kMsS + 1];
};

// blockIdx.y = net*S + s.  g_disp[net][s] = adjoint-upsample(cR * G_r + g * G_c) + cS * (gn / (mean+eps) - corr);
// block (0,0) also sums the pose gradients over the scales and runs the backward of transformation_from_parameters
__global__ __launch_bounds__(256) void ms_assemble_kernel(MsAssemble p, PoseParams pp, int pose_bwd, int vec0) {
  // workgroups of pass k: [first[k], first[k+1]) -- exactly as many as the pass has work for (an idle workgroup costs
  // ~0.4 ns of dispatch, and a y-dimension sized for the largest pass launched as many idle as busy ones)
  const int S = p.S, B = p.B;
  int pass = 0;
  while (pass + 1 < 2 * S && blockIdx.x >= p.first[pass + 1]) ++pass;
  const unsigned blk = blockIdx.x - p.first[pass];
  const int net = pass / S, s = pass - net * S;
  const float g = p.g_total ? *p.g_total : 1.0f;
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < B * 16; i += 256) {
      float a0 = 0.f, a1 = 0.f;
      for (int k = 0; k < S; ++k) {
        const float c = p.coefs[k] * g;
        a0 = fma_(c, p.gT[(size_t)(k * 2 + 0) * B * 16 + i], a0);
        a1 = fma_(c, p.gT[(size_t)(k * 2 + 1) * B * 16 + i], a1);
      }
      p.gTs0[i] = a0; p.gTs1[i] = a1;
    }
    __syncthreads();
    if (pose_bwd)
      for (int i = threadIdx.x; i < B * 2; i += 256) pose_bwd_one(pp, i / B, i % B);
  }
  float* out = p.g_disp[net][s];
  if (!out) return;
  const int f = 1 << s, h = p.H >> s, w = p.W >> s, W = p.W, H = p.H;
  const unsigned hw = (unsigned)(h * w), npix = (unsigned)B * hw;
  const float cR = p.coefs[pass] * g, cS = p.coefs[2 * S + s] * g;
  const float* G_c = net ? p.G_c[s] : nullptr;
  const float* gn = p.gn[net][s];
  // ONE element of work per thread and every load of it issued up front (the maps were written a step ago: this kernel
  // is HBM latency x bytes in flight): scale 0 four consecutive pixels (vec0: W % 4 == 0), the others member k of the f
  // lanes that share output pixel q
  const unsigned t = blk * 256u + threadIdx.x;
  if (s == 0) {
    const float* G_r = p.G_r[net][0];
    if (vec0) {
      const unsigned q = t * 4u;
      if (q >= npix) return;
      const unsigned b = q / hw;
      const float inv = div_(1.0f, (float)p.stats[pass * B + b] + 1e-7f), corr = (float)p.stats[2 * S * B + pass * B + b];
      const float4 r = *reinterpret_cast<const float4*>(G_r + q), n4 = *reinterpret_cast<const float4*>(gn + q);
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
      if (G_c) c = *reinterpret_cast<const float4*>(G_c + q);
      const float rr[4] = {r.x, r.y, r.z, r.w}, nn[4] = {n4.x, n4.y, n4.z, n4.w}, cc[4] = {c.x, c.y, c.z, c.w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = cR * rr[e];
        if (G_c) v = fma_(g, cc[e], v);
        o[e] = v + cS * (nn[e] * inv - corr);
      }
      *reinterpret_cast<float4*>(out + q) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
      if (t >= npix) return;
      const unsigned b = t / hw;
      const float inv = div_(1.0f, (float)p.stats[pass * B + b] + 1e-7f), corr = (float)p.stats[2 * S * B + pass * B + b];
      float v = cR * G_r[t];
      if (G_c) v = fma_(g, G_c[t], v);
      out[t] = v + cS * (gn[t] * inv - corr);
    }
    return;
  }
  const unsigned q = t >> s;
  const int k = (int)(t & (unsigned)(f - 1));
  const bool live = q < npix;
  const unsigned qq = live ? q : 0u;
  const unsigned b = qq / hw, pix = qq - b * hw;
  const int i = (int)(pix / (unsigned)w), j = (int)(pix - (unsigned)i * (unsigned)w);
  const size_t ob = (size_t)b * H * W;
  const float gnv = gn[qq];
  const float inv = div_(1.0f, (float)p.stats[pass * B + b] + 1e-7f), corr = (float)p.stats[2 * S * B + pass * B + b];
  const float v = adjoint_gather(p.G_r[net][s] + ob, cR, G_c ? G_c + ob : nullptr, g, i, j, k, f, h, w, H, W);
  if (live && k == 0) out[q] = v + cS * (gnv * inv - corr);
}

}  // namespace mal

using namespace mal;

static int ms_check(const mal_ms_args* a) {
  if (!a) return MAL_EINVAL;
  int rc = check_shape(a->B, a->H, a->W);
  if (rc) return rc;
  if (a->sclm < 0 || a->sclm >= kMsS) return MAL_EINVAL;
  if ((a->flags & MAL_STEP_NO_SSIM) && (a->flags & MAL_STEP_TEMPORAL)) return MAL_EINVAL;
  const int f = 1 << a->sclm;
  if (a->H % f || a->W % f || (a->H >> a->sclm) < 2 || (a->W >> a->sclm) < 2) return MAL_ESHAPE;
  if (!a->color0 || !a->color_m1 || !a->color_p1 || !a->K || !a->inv_K || !a->axisangle_m1 || !a->translation_m1 ||
      !a->axisangle_p1 || !a->translation_p1 || !a->consistency_mask || !a->augmentation_keep || !a->losses || !a->ws)
    return MAL_EINVAL;
  for (int s = 0; s <= a->sclm; ++s) {
    if (!a->disp_teacher[s] || !a->disp_student[s] || (s && !a->color0_s[s])) return MAL_EINVAL;
    if ((a->flags & MAL_STEP_NOISE_PHILOX) && a->noise[s]) return MAL_EINVAL;
  }
  if (a->ws_bytes < carve_ms(nullptr, a->B, a->H, a->W, a->sclm).bytes) return MAL_EWORKSPACE;
  return MAL_OK;
}

extern "C" size_t mal_ms_workspace_bytes(int B, int H, int W, int sclm) {
  if (B <= 0 || H <= 0 || W <= 0 || sclm < 0 || sclm >= kMsS) return 0;
  return carve_ms(nullptr, B, H, W, sclm).bytes;
}

extern "C" int mal_upsample_bilinear(const float* x, int B, int h, int w, int H, int W, float* out, void* stream) {
  if (!x || !out || B <= 0 || h <= 0 || w <= 0 || H < h || W < w) return MAL_EINVAL;
  UpMaps m = {};
  m.src[0] = x; m.dst[0] = out; m.h[0] = h; m.w[0] = w;
  upsample_launch(m, 1, B, H, W, (hipStream_t)stream);
  return launch_status();
}

extern "C" int mal_upsample_bilinear_adjoint(const float* g_out, int B, int h, int w, int H, int W, float* g_x, void* stream) {
  if (!g_out || !g_x || B <= 0 || h <= 0 || w <= 0) return MAL_EINVAL;
  const int f = h ? H / h : 0;
  if ((f != 1 && f != 2 && f != 4 && f != 8) || H != h * f || W != w * f) return MAL_ESHAPE;
  const size_t n = (size_t)B * h * w * f;
  hipLaunchKernelGGL(upsample_adjoint_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_out, B, h,
                     w, H, W, g_x);
  return launch_status();
}

static const float* ms_noise(const mal_ms_args* a, const MsWs& w, int s) {
  return (a->flags & MAL_STEP_NOISE_PHILOX) ? w.noise[s] : a->noise[s];
}
static MarchParams ms_teacher_params(const mal_ms_args* a, const MsWs& w, int s) {
  MarchParams p = march_params(a->B, a->H, a->W, a->min_depth, a->max_depth, 1e-7f, 0);
  p.disp = s ? w.up[0][s] : a->disp_teacher[0]; p.K = a->K; p.invK = a->inv_K; p.T[0] = w.T[0]; p.T[1] = w.T[1];
  p.src[0] = w.packed[1]; p.src[1] = w.packed[2]; p.target = w.packed[0];
  p.block_sums = w.bs[0][s]; p.block_gP = w.bgP[s];
  p.cam = w.cam; p.cam_ready = 1;
  p.no_ssim = (a->flags & MAL_STEP_NO_SSIM) ? 1 : 0;
  return p;
}
static void ms_fold_launch(const MsWs& w, int B, int H, int W, int S, bool teachers, bool students, hipStream_t st) {
  MsFold f = {};
  int segs = 0, rows = 0, n = 0;
  march_geometry(B, H, W, MAL_F_GRAD, nullptr, &segs, &rows);
  f.B = B; f.H = H; f.W = W; f.rows = rows; f.segs = segs;
  for (int s = 0; s < S; ++s)
    for (int net = 0; net < 2; ++net)
      if (net ? students : teachers) { f.G[n] = w.G_r[net][s]; f.bnd[n] = w.bnd[net][s]; ++n; }
  if (n) hipLaunchKernelGGL(ms_fold_kernel, dim3((unsigned)(B * segs * 2), (unsigned)n), dim3(256), 0, st, f);
}

// everything that does not wait for a temporal-hint producer
static int ms_front(const mal_ms_args* a, const MsWs& w, hipStream_t st, bool temporal, int* per_sample_sm /*[S]*/, int* forked) {
  *forked = 0;
  const int B = a->B, H = a->H, W = a->W, S = a->sclm + 1;
  int rc;
  // 1. first sweep: identity term (no noise: every scale adds its own), texel packing, poses, camera block
  {
    StepPoses sp = {};
    sp.pose.B = B; sp.pose.F = 2;
    sp.pose.axisangle[0] = a->axisangle_m1; sp.pose.axisangle[1] = a->axisangle_p1;
    sp.pose.translation[0] = a->translation_m1; sp.pose.translation[1] = a->translation_p1;
    sp.pose.invert[0] = 1; sp.pose.invert[1] = 0;
    sp.pose.T[0] = w.T[0]; sp.pose.T[1] = w.T[1];
    sp.K = a->K; sp.invK = a->inv_K; sp.cam = w.cam; sp.ticket = w.ticket;
    // 2. every scale's disparity at full resolution (trainer.py:1094-1096) and the per-scale tie-break noise maps: neither
    // reads what the sweep writes -- they ride on its launch as extra workgroups (four pixels per thread: W % 4 == 0 and
    // 16-byte aligned maps; launches of their own otherwise)
    UpMaps m = {};
    int k = 0;
    for (int s = 1; s < S; ++s)
      for (int n = 0; n < 2; ++n, ++k) {
        m.src[k] = n ? a->disp_student[s] : a->disp_teacher[s];
        m.dst[k] = w.up[n][s]; m.h[k] = H >> s; m.w[k] = W >> s;
      }
    bool fold = g_ms_fold && W % 4 == 0;
    for (int i = 0; i < k; ++i) fold = fold && ((uintptr_t)m.dst[i] % 16 == 0);
    const bool philox = (a->flags & MAL_STEP_NOISE_PHILOX) != 0;
    MsExtra x = {};
    if (fold) {
      x.seed = a->noise_seed; x.step = a->noise_step; x.counter = (const unsigned long long*)a->noise_counter; x.mult = (unsigned)S;
      x.n_noise = philox ? S : 0;
      for (int s = 0; s < S; ++s) x.noise[s] = w.noise[s];
      x.up = m; x.n_up = k; x.B = B; x.H = H; x.W = W;
      x.noise_blocks = (int)(((size_t)B * ((H + 3) / 4) * W + 63) / 64);
      x.up_blocks = (int)(((size_t)B * H * (W / 4) + 63) / 64);
    }
    rc = pack_identity_launch(a->color0, a->color_m1, a->color_p1, B, H, W, w.packed[1], w.packed[2], w.packed[0], w.ident, st,
                              &sp, nullptr, nullptr, nullptr, false, (a->flags & MAL_STEP_NO_SSIM) ? 1 : 0, nullptr,
                              fold ? &x : nullptr);
    if (rc) return rc;
    if (!fold) {
      if (philox) {
        rc = tiebreak_noise_launch(a->noise_seed, a->noise_step, (const unsigned long long*)a->noise_counter, (unsigned)S, S, B, H, W,
                                   w.noise, st);
        if (rc) return rc;
      }
      if (k > 0) {
        upsample_launch(m, k, B, H, W, st);
        rc = launch_status();
        if (rc) return rc;
      }
    }
  }
  const int packed = MAL_F_SRC_PACKED | MAL_F_TGT_PACKED;
  const float merge_cons = (float)(1.0 / ((double)S * (double)B * H * W));
  if (a->flags & MAL_STEP_ENSEMBLE) {
    if (hipMemsetD32Async((hipDeviceptr_t)w.neg_inf, (int)0xff800000u, (size_t)B * H * W, st) != hipSuccess) return MAL_ELAUNCH;
  }
  auto teacher = [&](int s) -> int {  // automask against the identity term + this scale's noise (trainer.py:1296-1311)
    MarchParams p = ms_teacher_params(a, w, s);
    p.ident = w.ident; p.noise = ms_noise(a, w, s);
    if (temporal) {
      // forward only, in front of the producer: the warped images out; min over the two warped candidates, its winner, the
      // automask weight and the sums stay for the fused sweep of mal_loss_multiscale_fwd (second copies: what it re-decides)
      p.min_reproj = w.rp_warp[s]; p.argmin_out = w.arg_warp[s];
      p.min_reproj2 = w.rp4[s]; p.argmin_out2 = w.arg_t[s]; p.weight_out = w.w_t[s];
      p.color_out[0] = a->warp_m1[s]; p.color_out[1] = a->warp_p1[s]; p.color_out_stride = a->warp_sample_stride;
      return march_launch(p, MAL_F_AUTOMASK | packed, st);
    }
    p.g_reproj = w.G_r[0][s]; p.bnd = g_march_halo1 ? w.bnd[0][s] : nullptr;
    p.dbg = a->dec_teacher[s];
    return march_launch(p, MAL_F_AUTOMASK | MAL_F_GRAD | MAL_F_POSE_GRAD | packed, st);
  };
  auto student = [&](int s, hipStream_t q) -> int {
    // mask = consistency (x matching, formed at scale 0 from the teacher's scale-0 depth, trainer.py:592-593)
    // x (1 - augmentation); consistency term against the teacher's depth of the same scale (:1330-1336)
    MarchParams p = march_params(B, H, W, a->min_depth, a->max_depth, 1e-7f, 0);
    p.disp = s ? w.up[1][s] : a->disp_student[0]; p.K = a->K; p.invK = a->inv_K; p.T[0] = w.T[0]; p.T[1] = w.T[1];
    p.src[0] = w.packed[1]; p.src[1] = w.packed[2]; p.target = w.packed[0];
    if (!(a->flags & MAL_STEP_NO_AUG)) { p.sample_scale = a->augmentation_keep; p.sample_scale_is_mask = (a->flags & MAL_STEP_AUG_MASK) ? 1 : 0; }
    p.mono_disp = s ? w.up[0][s] : a->disp_teacher[0];
    if (a->flags & MAL_STEP_NO_MOTION_MASK) {
      // the weight is 1 (x augmentation): no mask operand at all
    } else if (s == 0) {
      p.ext_mask = a->consistency_mask; p.lowest_cost = a->lowest_cost;
      if (a->lowest_cost) p.cmask_out = a->consistency_mask_out ? a->consistency_mask_out : w.cmask;
    } else {
      p.ext_mask = a->lowest_cost ? (a->consistency_mask_out ? a->consistency_mask_out : w.cmask) : a->consistency_mask;
    }
    p.mono_reproj = w.ident;  // the distillation selection is not part of this loss: any map serves, its term has weight 0
    p.g_reproj = w.G_r[1][s]; p.g_cons = w.G_c[s]; p.g_distil = nullptr; p.bnd = g_march_halo1 ? w.bnd[1][s] : nullptr;
    p.merge_cons = merge_cons; p.merge_distil = 0.f;
    if (a->flags & MAL_STEP_ENSEMBLE) {
      // --ensemble (trainer.py:1346-1351): |(mono + multi)/2 - multi| * mask IS the distillation term with the ensemble winning
      // everywhere (index 1: target (mono + multi)/2, mono detached, weight = the mask), same 1/N as the consistency term
      p.ens_reproj = w.neg_inf; p.merge_distil = merge_cons;
    }
    p.block_sums = w.bs[1][s]; p.block_gP = w.bgP[s];
    p.cam = w.cam; p.cam_ready = 1;
    p.no_ssim = (a->flags & MAL_STEP_NO_SSIM) ? 1 : 0;
    p.dbg = a->dec_student[s];
    return march_launch(p, MAL_F_GRAD | MAL_F_EPILOGUE | packed, q);
  };
  // Without the hint the two networks' passes of a scale stay next to each other.  With it: the teachers' forward passes first;
  // everything else that does not wait for a producer -- the students' passes, their boundary rows, the smoothness sweeps --
  // goes onto the library's side stream, beside the producers' small kernels that the host enqueues on `st` between this call
  // and mal_loss_multiscale_fwd (which joins)
  hipStream_t st2 = st;
  if (!temporal) {
    for (int s = 0; s < S; ++s) {
      rc = teacher(s);
      if (!rc) rc = student(s, st);
      if (rc) return rc;
    }
  } else {
    for (int s = 0; s < S; ++s) {
      rc = teacher(s);
      if (rc) return rc;
    }
    hipStream_t side = side_begin(st);
    if (side) { st2 = side; *forked = 1; }
    for (int s = 0; s < S && !rc; ++s) rc = student(s, st2);
    if (rc) { if (*forked) (void)side_end(st); return rc; }
  }
  if (g_march_halo1) {  // one-row halo of the gradient passes: each boundary row's missing window row, in ONE launch
    ms_fold_launch(w, B, H, W, S, !temporal, true, st2);
    rc = launch_status();
    if (rc) { if (*forked) (void)side_end(st); return rc; }
  }
  {  // smoothness of both disparity maps at every scale's own size against the target at that size (:1469-1471): ONE launch
    const float *sd[2 * kMsS], *si[2 * kMsS];
    float* sg[2 * kMsS];
    double* sp[2 * kMsS];
    int sh[2 * kMsS], sw[2 * kMsS], per[2 * kMsS];
    for (int s = 0; s < S; ++s)
      for (int n = 0; n < 2; ++n) {
        const int k = 2 * s + n;
        sd[k] = n ? a->disp_student[s] : a->disp_teacher[s];
        si[k] = (s == 0 && !a->color0_s[0]) ? a->color0 : a->color0_s[s];
        sg[k] = w.gn[n][s]; sp[k] = w.sm[n][s]; sh[k] = H >> s; sw[k] = W >> s;
      }
    rc = smooth_march_sweep_batch(2 * S, sd, si, B, sh, sw, sg, sp, st2, per);
    if (rc) { if (*forked) (void)side_end(st); return rc; }
    for (int s = 0; s < S; ++s) per_sample_sm[s] = per[2 * s];
  }
  if (*forked) return side_end(st);
  return MAL_OK;
}

// tasks per sample of the smoothness sweep of scale s (the sweep ran in an earlier call: mal_loss_multiscale_warp)
static int ms_smooth_tasks(int H, int W, int s) { return smooth_march_batch_tasks(H >> s, W >> s); }

// the fused sweeps of the temporal hint (one per scale) and the reduction
static int ms_back(const mal_ms_args* a, const MsWs& w, hipStream_t st, bool temporal, const int* per_sample_sm) {
  const int B = a->B, H = a->H, W = a->W, S = a->sclm + 1;
  MsFinal fin = {};
  int strips = 0, segs = 0;
  march_geometry(B, H, W, MAL_F_GRAD, &strips, &segs, nullptr);
  fin.per_sample = strips * segs; fin.per_sample_t = fin.per_sample;
  if (temporal) {
    march_geometry(B, H, W, 0, &strips, &segs, nullptr);
    fin.per_sample_t = strips * segs;
    // every scale's sweep in ONE launch when all of them have a region map (same decomposition): each is a few short tasks
    bool batch = g_ms_fold && S > 1;
    for (int s = 0; s < S; ++s) batch = batch && a->syn_region[s] != nullptr;
    FusedMoreArgs fa[kMsS] = {};
    for (int s = 0; s < S; ++s) {
      if (!a->syn_m1[s] || !a->syn_p1[s] || !a->g_syn_m1[s] || !a->g_syn_p1[s]) { (void)side_wait(st, false); return MAL_EINVAL; }
      const bool sparse = (a->syn_sparse >> s) & 1;
      if (sparse && (!a->syn_region[s] || !a->warp_m1[s] || !a->warp_p1[s])) { (void)side_wait(st, false); return MAL_EINVAL; }
      if (batch) {
        FusedMoreArgs& x = fa[s];
        x.target = w.packed[0]; x.cand0 = a->syn_m1[s]; x.cand1 = a->syn_p1[s]; x.idx0 = 2; x.ident = w.ident; x.noise = ms_noise(a, w, s);
        x.prev_min = w.rp_warp[s]; x.prev_arg = w.arg_warp[s]; x.min_reproj = w.rp4[s]; x.argmin = w.arg_t[s]; x.weight_out = w.w_t[s];
        x.block_sums = w.bs_ph[s]; x.g_cand0 = a->g_syn_m1[s]; x.g_cand1 = a->g_syn_p1[s]; x.region = a->syn_region[s];
        x.g_region0 = a->g_syn_region_m1[s]; x.g_region1 = a->g_syn_region_p1[s];
        x.orig0 = sparse ? a->warp_m1[s] : nullptr; x.orig1 = sparse ? a->warp_p1[s] : nullptr;
        x.orig_stride = (size_t)a->warp_sample_stride; x.weight_given = 0;
        fin.bs_ph[s] = w.bs_ph[s];
        continue;
      }
      int rc = photo_march_fused_more(w.packed[0], a->syn_m1[s], a->syn_p1[s], 2, w.ident, ms_noise(a, w, s), w.rp_warp[s],
                                      w.arg_warp[s], B, H, W, w.rp4[s], w.arg_t[s], w.w_t[s], w.bs_ph[s], a->g_syn_m1[s],
                                      a->g_syn_p1[s], &fin.per_sample_ph[s], st, a->syn_region[s], a->g_syn_region_m1[s],
                                      a->g_syn_region_p1[s], nullptr, nullptr, sparse ? a->warp_m1[s] : nullptr,
                                      sparse ? a->warp_p1[s] : nullptr, (size_t)a->warp_sample_stride, 1, 0);
      if (rc) { (void)side_wait(st, false); return rc; }
      fin.bs_ph[s] = w.bs_ph[s];
    }
    if (batch) {
      int per = 0;
      const int rc = photo_march_fused_more_n(S, fa, B, H, W, &per, st);
      if (rc) { (void)side_wait(st, false); return rc; }
      for (int s = 0; s < S; ++s) fin.per_sample_ph[s] = per;
    }
    // the students' passes and the smoothness sweeps ran on the side stream beside the producers (mal_loss_multiscale_warp)
    int rc = side_wait(st, false);
    if (rc) return rc;
  }
  for (int s = 0; s < S; ++s) {
    for (int n = 0; n < 2; ++n) { fin.bs[n][s] = w.bs[n][s]; fin.sm[n][s] = w.sm[n][s]; }
    fin.bgP[s] = temporal ? nullptr : w.bgP[s];
    fin.per_sample_sm[s] = per_sample_sm[s];
  }
  fin.K = a->K; fin.B = B; fin.H = H; fin.W = W; fin.S = S; fin.ensemble = (a->flags & MAL_STEP_ENSEMBLE) ? 1 : 0;
  fin.ps = w.ps; fin.stats = w.stats; fin.gT = w.gT; fin.losses = a->losses; fin.coefs = w.coefs;
  fin.loss_total = a->loss_total; fin.ticket = w.ticket;
  fin.noise_counter = (a->flags & MAL_STEP_NOISE_PHILOX) ? (unsigned long long*)a->noise_counter : nullptr;
  hipLaunchKernelGGL(ms_final_kernel, dim3((temporal ? 2 : 3) * S * B), dim3(256), 0, st, fin);
  return launch_status();
}

extern "C" int mal_loss_multiscale_warp(const mal_ms_args* a) {
  int rc = ms_check(a);
  if (rc) return rc;
  if (!(a->flags & MAL_STEP_TEMPORAL)) return MAL_EINVAL;
  for (int s = 0; s <= a->sclm; ++s)
    if (!a->warp_m1[s] || !a->warp_p1[s]) return MAL_EINVAL;
  MsWs w = carve_ms(a->ws, a->B, a->H, a->W, a->sclm);
  int per_sm[kMsS], forked = 0;
  rc = side_wait((hipStream_t)a->stream, false);  // a previous step that was abandoned after its fork
  if (rc) return rc;
  return ms_front(a, w, (hipStream_t)a->stream, true, per_sm, &forked);
}

// A MAL_STEP_TEMPORAL step that called mal_loss_multiscale_warp but will not call _fwd (a producer raised): joins the side stream
extern "C" int mal_loss_multiscale_abort(const mal_ms_args* a) {
  if (!a) return MAL_EINVAL;
  return side_wait((hipStream_t)a->stream, false);
}

extern "C" int mal_loss_multiscale_fwd(const mal_ms_args* a) {
  int rc = ms_check(a);
  if (rc) return rc;
  MsWs w = carve_ms(a->ws, a->B, a->H, a->W, a->sclm);
  hipStream_t st = (hipStream_t)a->stream;
  const bool temporal = (a->flags & MAL_STEP_TEMPORAL) != 0;
  int per_sm[kMsS] = {};
  if (!temporal) {
    int forked = 0;
    rc = ms_front(a, w, st, false, per_sm, &forked);
    if (rc) return rc;
  } else {
    for (int s = 0; s <= a->sclm; ++s) per_sm[s] = ms_smooth_tasks(a->H, a->W, s);
  }
  return ms_back(a, w, st, temporal, per_sm);
}

extern "C" int mal_loss_multiscale_bwd(const mal_ms_args* a) {
  int rc = ms_check(a);
  if (rc) return rc;
  const int B = a->B, H = a->H, W = a->W, S = a->sclm + 1;
  MsWs w = carve_ms(a->ws, B, H, W, a->sclm);
  if (a->flags & MAL_STEP_TEMPORAL) {
    // the teacher's gradient sweep per scale, with the four-way decisions of _fwd and what reaches the warped images through
    // syn added before the chain rule through the warp; then the boundary rows and the pose partials it left
    hipStream_t st = (hipStream_t)a->stream;
    MsPose mp = {};
    for (int s = 0; s < S; ++s) {
      if (!a->g_warp_m1[s] || !a->g_warp_p1[s]) return MAL_EINVAL;
      MarchParams p = ms_teacher_params(a, w, s);
      p.ident = w.ident;  // unused by the TEMPORAL instantiation (the launch checks the flag combination only)
      p.g_reproj = w.G_r[0][s]; p.bnd = g_march_halo1 ? w.bnd[0][s] : nullptr;
      p.forced_w = w.w_t[s]; p.forced_arg = w.arg_t[s]; p.g_color[0] = a->g_warp_m1[s]; p.g_color[1] = a->g_warp_p1[s];
      p.dbg = a->dec_teacher[s];
      rc = march_launch(p, MAL_F_AUTOMASK | MAL_F_GRAD | MAL_F_POSE_GRAD | MAL_F_SRC_PACKED | MAL_F_TGT_PACKED, st);
      if (rc) return rc;
      mp.bgP[s] = w.bgP[s]; mp.per_sample = p.strips * p.segs;
    }
    if (g_march_halo1) {
      ms_fold_launch(w, B, H, W, S, true, false, st);
      rc = launch_status();
      if (rc) return rc;
    }
    mp.K = a->K; mp.gT = w.gT; mp.B = B;
    hipLaunchKernelGGL(ms_pose_kernel, dim3((unsigned)(S * B)), dim3(256), 0, st, mp);
    rc = launch_status();
    if (rc) return rc;
  }
  MsAssemble p = {};
  for (int s = 0; s < S; ++s) {
    for (int n = 0; n < 2; ++n) { p.G_r[n][s] = w.G_r[n][s]; p.gn[n][s] = w.gn[n][s]; }
    p.G_c[s] = w.G_c[s];
    p.g_disp[0][s] = a->g_disp_teacher[s]; p.g_disp[1][s] = a->g_disp_student[s];
  }
  p.coefs = w.coefs; p.stats = w.stats; p.g_total = a->g_total; p.gT = w.gT; p.gTs0 = w.gTs[0]; p.gTs1 = w.gTs[1];
  p.B = B; p.H = H; p.W = W; p.S = S;
  PoseParams pp = {};
  pp.B = B; pp.F = 2;
  pp.axisangle[0] = a->axisangle_m1; pp.axisangle[1] = a->axisangle_p1;
  pp.translation[0] = a->translation_m1; pp.translation[1] = a->translation_p1;
  pp.invert[0] = 1; pp.invert[1] = 0;
  pp.gT[0] = w.gTs[0]; pp.gT[1] = w.gTs[1];
  pp.g_axisangle[0] = a->g_axisangle_m1; pp.g_axisangle[1] = a->g_axisangle_p1;
  pp.g_translation[0] = a->g_translation_m1; pp.g_translation[1] = a->g_translation_p1;
  const int pose_bwd = (a->g_axisangle_m1 || a->g_translation_m1 || a->g_axisangle_p1 || a->g_translation_p1) ? 1 : 0;
  // one unit of work per thread: B*H*W / 4 quads at scale 0 (B*H*W when W % 4 != 0), B*H*W / 2**s group members above;
  // a pass whose output is not wanted gets no workgroups (pass 0 keeps one: it carries the pose backward)
  bool vec0 = W % 4 == 0;
  for (int n = 0; n < 2; ++n) vec0 = vec0 && ((uintptr_t)p.g_disp[n][0] % 16 == 0);
  const size_t px = (size_t)B * H * W;
  for (int k = 0; k < 2 * S; ++k) {
    const int s = k % S;
    const size_t units = s ? px >> s : (vec0 ? px / 4 : px);
    unsigned blocks = p.g_disp[k / S][s] ? (unsigned)((units + 255) / 256) : 0u;
    if (k == 0 && blocks == 0) blocks = 1;
    p.first[k + 1] = p.first[k] + blocks;
  }
  hipLaunchKernelGGL(ms_assemble_kernel, dim3(p.first[2 * S]), dim3(256), 0, (hipStream_t)a->stream, p, pp, pose_bwd,
                     vec0 ? 1 : 0);
  return launch_status();
}

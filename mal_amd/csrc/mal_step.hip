// mal_loss_step_fwd / _bwd: the whole loss half of process_batch (manydepth/trainer.py:573-642,
// --distil) as ONE host call that enqueues a fixed list of kernels on the caller's stream.
//
// The reference issues ~500 ATen launches for this (3 warp passes, 10 SSIM evaluations, masks,
// reductions, pose composition, and the same again in autograd's backward); the operator-level
// API of this library still needs ~60 launches plus Python glue between them.  Here:
//
//   forward  1 identity term min_f r(src_f, target) (loss_utils.py:92-101) + packing of the images into 12-byte
//              texels; B extra workgroups: poses of both frames (layers.py:26-100) + camera block P = K T
//            1 teacher pass  (warp+SSIM+L1+min+automask+smoothness, fwd+bwd to disp and poses)   :573-581
//            1 ensemble pass ((disp_t+disp_s)/2 formed in the kernel, no grad)                   :594-600
//            1 student pass  (matching mask, consistency*(1-augmentation) mask, mono depth from the teacher's
//                             disparity, consistency + distillation epilogue, smoothness)       :592-612
//            1 reduction of the passes' partials + pose gradients + loss scalars (fixed order, no atomics)
//   backward 1 gradient assembly (both disparity maps) with the pose backward in its first block
//
// All intermediate maps live in the caller's workspace; loss scalars stay on the device.
#include "mal_march.h"
#include "mal_device.h"
#include "mal_pose.h"
#include "mal_pairs.h"
#include <mutex>

namespace mal {

// mal_photo_march.hip: the materialised-candidate kernels (the two synthesised images of the temporal hint)
int photo_march_fused_more(const float* target, const float* cand0, const float* cand1, int idx0, const float* ident,
                           const float* noise, const float* prev_min, const uint8_t* prev_arg, int B, int H, int W,
                           float* min_reproj, uint8_t* argmin, float* weight_out, double* block_sums, float* g_cand0,
                           float* g_cand1, int* per_sample_out, hipStream_t st, const uint8_t* region, float* g_region0,
                           float* g_region1, unsigned* order, unsigned* order_count, const float* orig0, const float* orig1,
                           size_t orig_stride, int target_texels, int weight_given);

constexpr int kLossSlots = 16;
constexpr int kEpiBlocks = 64;  // epilogue workgroups per sample in the final launch (their partials are summed in block order)

struct StepWs {
  float* packed[3];   // target, src-1, src+1 as (B,H,W,kTexel) texels
  float* T[2]; float* gT[2]; float* gTs[2];
  float* ident; float* mono_reproj; float* ens_reproj;
  float* G_r_t; float* G_r_s; float* G_c; float* gn_t; float* gn_s;
  float* G_e;         // --learn_ens: merge_distil * d distil / d ens_disp (student pass -> assembly)
  float* multi_reproj;  // --temporal step: the student's per-pixel min (its pass runs beside the producer, the epilogue after it)
  double* bs_d;       // ... and the epilogue launch's per-block partials [B][kEpiBlocks][2] = sum |d cons|, sum |d distil|
  double* bs_t; double* bs_s; double* bs_e; float* bgP;  // per-task partials of the three passes
  float* bgP_e;       // the ensemble pass's own (unused) pose-partial sink: it may run beside other launches
  double* bs_p;       // per-task smoothness partials of the first launch: [task][map][4]
  // temporal hint: winner of the four-way min (0/1 warped, 2/3 syn) and automask weight of the teacher, the
  // materialised-candidate kernel's per-task partials [task][2], what the step remembers between its calls
  unsigned char* arg_t; float* w_t; double* bs_ph;
  float* rp_warp; unsigned char* arg_warp;  // ... and min_f r(warp_f) / its winner as the pass in front of the producer leaves them
  // --main_temporal: the same for the student's pass (w_s: its weight = consistency x matching x (1 - augmentation), constant
  // under the four-way min; bs_s then holds the sums of the forward pass in front of the producer, bs_sh the sweep's differences)
  unsigned char* arg_s; float* w_s; double* bs_sh; float* rp_warp_s; unsigned char* arg_warp_s;
  double* ps;         // per-sample sums of the teacher's, then the student's partials: [2][B][8]
  unsigned* ticket;   // completion counter of step_final_kernel; ticket[1..2]: counts of the fused sweep's task order
  unsigned* order;    // dispatch order of the fused sweep's tasks (temporal hint, region map given)
  double* sm_stats;   // [4B]: mean_t[b], mean_s[b], corr_t[b], corr_s[b] of the mean-normalised smoothness
  float* coefs;       // 16 device scalars for the backward
  float* cam;         // [B][40] camera block of the marching kernels
  float* bnd_t; float* bnd_s;  // boundary scratch rows of the teacher's / student's gradient pass (one-row halo, mal_march.h)
  size_t bytes;
};

static StepWs carve_step(void* base, int B, int H, int W) {
  StepWs w;
  char* p = (char*)base;
  size_t o = 0;
  const size_t HW = (size_t)H * W, map = align256(B * HW * sizeof(float)), nb = ws_blocks(B, H, W);
  auto take = [&](size_t bytes) { char* r = p + o; o += align256(bytes); return r; };
  for (int i = 0; i < 3; ++i) w.packed[i] = (float*)take(B * HW * kTexel * sizeof(float));
  for (int f = 0; f < 2; ++f) { w.T[f] = (float*)take(B * 16 * 4); w.gT[f] = (float*)take(B * 16 * 4); w.gTs[f] = (float*)take(B * 16 * 4); }
  float** maps[] = {&w.ident, &w.mono_reproj, &w.ens_reproj, &w.G_r_t, &w.G_r_s, &w.G_c,
                    &w.gn_t, &w.gn_s, &w.G_e, &w.multi_reproj};
  for (auto m : maps) *m = (float*)take(map);
  w.bs_d = (double*)take((size_t)B * kEpiBlocks * 2 * 8);
  w.bs_t = (double*)take(nb * 8 * 8); w.bs_s = (double*)take(nb * 8 * 8); w.bs_e = (double*)take(nb * 8 * 8);
  w.bgP = (float*)take(nb * 24 * 4);
  w.bgP_e = (float*)take(nb * 24 * 4);
  w.bs_p = (double*)take(nb * 8 * 8);
  w.arg_t = (unsigned char*)take((size_t)B * HW);
  w.w_t = (float*)take(map);
  w.bs_ph = (double*)take(nb * 4 * 2 * 8);  // tasks of >= 2 rows (the fused sweep shortens them when a region map is given)
  w.rp_warp = (float*)take(map);
  w.arg_warp = (unsigned char*)take((size_t)B * HW);
  w.arg_s = (unsigned char*)take((size_t)B * HW);
  w.w_s = (float*)take(map);
  w.bs_sh = (double*)take(nb * 4 * 2 * 8);
  w.rp_warp_s = (float*)take(map);
  w.arg_warp_s = (unsigned char*)take((size_t)B * HW);
  w.ps = (double*)take((size_t)2 * B * 8 * 8);
  w.ticket = (unsigned*)take(16);
  w.order = (unsigned*)take(nb * 4 * sizeof(unsigned));  // tasks of >= 2 rows, as bs_ph
  w.sm_stats = (double*)take((size_t)4 * B * 8);
  w.coefs = (float*)take(16 * 4);
  w.cam = (float*)take((size_t)B * 40 * 4);
  w.bnd_t = (float*)take(march_bnd_floats(B, H, W) * 4);
  w.bnd_s = (float*)take(march_bnd_floats(B, H, W) * 4);
  w.bytes = o;
  return w;
}

// The consistency / distillation terms of the student (loss_utils.py:193-254) as a launch of their own -- the --temporal step:
// they are pointwise in the two depths and in the three per-pixel minima (the distillation index is not differentiable, the
// gradient reaches multi_depth directly, not through the warp), so the student's marching pass need not wait for the
// teacher's four-way min and runs beside the temporal hint's producer; this launch follows the join.  The arithmetic is the
// epilogue of march_body, statement for statement (weight = consistency mask x matching mask x (1 - augmentation), as the
// pass forms it; depth_of, the argmin's first-minimum rule, sign conventions); sums leave as per-block partials.  (Riding as
// leading workgroups of step_final_kernel -- one launch less -- was built in round 4 and measured SLOWER, 0.330 against 0.3255 ms:
// 768 more workgroups each pay an agent-scope release and a ticket atomic where a kernel boundary publishes for free; removed.)
struct EpiParams {
  const float *disp_s, *disp_t, *mono_reproj, *ens_reproj, *multi_reproj, *ext_mask, *lowest_cost, *sample_scale, *ens_disp;
  int scale_is_mask, dual;
  float min_disp, range, merge_cons, merge_distil;
  float *G_c, *G_e; double* partials; unsigned* dbg; int B, HW;
};
MAL_DEV void step_epilogue_block(const EpiParams& p, int bid) {
  __shared__ double s_red[2][4];
  const int b = bid / kEpiBlocks, blk = bid - b * kEpiBlocks, tid = threadIdx.x;
  const int per = (p.HW + kEpiBlocks - 1) / kEpiBlocks, lo = blk * per, hi = min(lo + per, p.HW);
  const float sscale = p.sample_scale ? (p.scale_is_mask ? 1.0f - p.sample_scale[b] : p.sample_scale[b]) : 1.0f;
  const bool has_er = p.ens_reproj != nullptr, learnt = p.ens_disp != nullptr;
  float acc_cons = 0.f, acc_dist = 0.f;
  // one pixel: -> merged gradient, the extra map's value (learnt ensemble / dual distillation), the distillation index
  auto one = [&](float ext, float dt, float cost, float rp, float ds, float mr, float er, float ensd, float& g_c, float& g_e,
                 unsigned& idx_out) __attribute__((always_inline)) {
    // the weight of the student's masked reprojection term, as its pass forms it (march_body, stage S)
    float em = ext;
    const float mono = depth_of(dt, p.min_disp, p.range);
    {
      const float matching = div_safe_(1.0f, cost);
      const bool ok = (div_safe_(matching - mono, mono) < 1.0f) && (div_safe_(mono - matching, matching) < 1.0f);
      em = ok ? em : em * 0.0f;
    }
    float w = 1.0f;
    w *= em;
    w *= sscale;
    // march_body::epilogue
    const float dm = depth_of(ds, p.min_disp, p.range);
    const float ddepth = -(dm * dm) * p.range;
    const float dmono = mono;
    const float m = w, cm = 1.0f - m, mm = 1.0f - cm;
    const float dc = dm - dmono;
    acc_cons += fabsf(dc) * cm;
    int idx = 0;
    float best = mr;
    if (has_er) {
      if (er < best) { best = er; idx = 1; }
    }
    if (rp < best) idx = 2;
    float dens = 0.f, ens = (dmono + dm) / 2.0f;
    if (learnt) { dens = depth_of(ensd, p.min_disp, p.range); ens = dens; }
    const float target = idx == 0 ? dmono : (idx == 2 ? dm : ens);
    const float dd = target - dm;
    acc_dist += fabsf(dd) * mm;
    const float gc = sgnf(dc) * cm * ddepth;
    const float gd = sgnf(dd) * (idx == 0 ? -1.0f : (idx == 2 ? 0.0f : (learnt ? -1.0f : -0.5f))) * mm * ddepth;
    g_e = 0.f;
    if (learnt) g_e = idx == 1 ? p.merge_distil * (sgnf(dd) * mm * (-(dens * dens) * p.range)) : 0.0f;
    if (p.dual) g_e = idx == 0 ? p.merge_distil * (sgnf(dd) * mm * (-(dmono * dmono) * p.range)) : 0.0f;
    g_c = fma_(p.merge_cons, gc, p.merge_distil * gd);
    idx_out = (unsigned)idx;
  };
  const size_t base = (size_t)b * p.HW;
  // the 16-byte path needs every map it touches 16-byte aligned (torch's allocations are; a caller's view at an odd offset is not)
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  const bool aligned = al16(p.ext_mask) && al16(p.disp_t) && al16(p.lowest_cost) && al16(p.multi_reproj) && al16(p.disp_s) &&
                       al16(p.mono_reproj) && al16(p.ens_reproj) && al16(p.ens_disp) && al16(p.G_c) && al16(p.G_e);
  if (((per | p.HW) & 3) == 0 && aligned) {
    // four consecutive pixels per thread, 16-byte accesses (every map's sample and every block's range start 16-byte aligned)
    typedef float v4 __attribute__((ext_vector_type(4)));
    for (int i0 = lo + tid * 4; i0 < hi; i0 += 1024) {
      const size_t i = base + i0;
      auto ld = [&](const float* m) { return *reinterpret_cast<const v4*>(m + i); };
      const v4 ext = ld(p.ext_mask), dt = ld(p.disp_t), cost = ld(p.lowest_cost), rp = ld(p.multi_reproj), ds = ld(p.disp_s),
               mr = ld(p.mono_reproj), er = has_er ? ld(p.ens_reproj) : (v4){0.f, 0.f, 0.f, 0.f},
               ensd = learnt ? ld(p.ens_disp) : (v4){0.f, 0.f, 0.f, 0.f};
      v4 gc, ge;
      unsigned idx[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { float a, e; one(ext[k], dt[k], cost[k], rp[k], ds[k], mr[k], er[k], ensd[k], a, e, idx[k]); gc[k] = a; ge[k] = e; }
      *reinterpret_cast<v4*>(p.G_c + i) = gc;
      if (p.G_e && (learnt || p.dual)) *reinterpret_cast<v4*>(p.G_e + i) = ge;
      if (p.dbg)
#pragma unroll
        for (int k = 0; k < 4; ++k) p.dbg[(size_t)MAL_DEC_DISTIL * ((size_t)p.B * p.HW) + i + k] = idx[k];
    }
  } else {
    for (int i0 = lo + tid; i0 < hi; i0 += 256) {
      const size_t i = base + i0;
      float a, e;
      unsigned idx;
      one(p.ext_mask[i], p.disp_t[i], p.lowest_cost[i], p.multi_reproj[i], p.disp_s[i], p.mono_reproj[i],
          has_er ? p.ens_reproj[i] : 0.f, learnt ? p.ens_disp[i] : 0.f, a, e, idx);
      p.G_c[i] = a;
      if (p.G_e && (learnt || p.dual)) p.G_e[i] = e;
      if (p.dbg) p.dbg[(size_t)MAL_DEC_DISTIL * ((size_t)p.B * p.HW) + i] = idx;
    }
  }
  const double r0 = wave_sum_d((double)acc_cons), r1 = wave_sum_d((double)acc_dist);
  if ((tid & 63) == 0) { s_red[0][tid >> 6] = r0; s_red[1][tid >> 6] = r1; }
  __syncthreads();
  if (tid < 2) p.partials[((size_t)b * kEpiBlocks + blk) * 2 + tid] = (s_red[tid][0] + s_red[tid][1]) + (s_red[tid][2] + s_red[tid][3]);
}

__global__ __launch_bounds__(256) void step_epilogue_kernel(EpiParams p) { step_epilogue_block(p, (int)blockIdx.x); }

// ---------------------------------------------------------------- small kernels
// Everything after the three passes, fixed summation order, no floating-point atomics.  3B blocks:
//   blocks [0, 2B):  ps[pass][b][j] = sum over the sample's tasks (contiguous) of the per-task partials: j < 4 from the
//                    marching pass (block_sums[task][j]), j >= 4 the smoothness partials of the first launch
//   blocks [2B, 3B): g_T[f][b] = K_b^T [gP_fb ; 0]  from the teacher's per-task pose partials
// and the block that finishes last (a ticket counter, reset by the step's first launch) turns the per-sample sums into the smoothness of the mean-normalised disparities (layers.py:210-223,
// loss_utils.py:119-121), the loss scalars (loss_utils.py:112-127,198-279; trainer.py:625-629) and the
// coefficients of the backward.
__global__ __launch_bounds__(256) void step_final_kernel(const double* bs_t, const double* bs_s, const float* bgP,
                                                         const double* bs_p, int per_sample_p,
                                                         const double* bs_ph, int per_sample_ph, const double* bs_sh,
                                                         int per_sample_sh, const double* bs_d,
                                                         const float* K, int per_sample, int per_sample_t, int B, int H, int W,
                                                         float w_main, float w_distil, double* ps, float* gT0, float* gT1,
                                                         double* stats, float* losses, float* coefs, float* loss_total,
                                                         unsigned* ticket, unsigned long long* noise_counter) {
  __shared__ double s_part[256];
  __shared__ double s_gP[24];
  __shared__ double sh_tot[2][8];
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, HW = H * W;
  const int bid = (int)blockIdx.x;
  if (bid < 2 * B) {
    const int pass = bid / B, b = bid - pass * B;
    const int j = tid & 7, sub = tid >> 3;  // 32 strided partial sums per quantity
    // temporal hint: the teacher's sum(rp*w), sum(w) come from the materialised-candidate kernel ([task][2])
    // temporal hint: the teacher's sums are those of the pass in front of the producer (bs_t, its own decomposition:
    // per_sample_t tasks) plus the fused sweep's per-task differences ([task][2])
    const bool ph = pass == 0 && bs_ph != nullptr;
    const int ps_pass = pass ? per_sample : per_sample_t;
    const double* bs = j < 4 ? (pass ? bs_s : bs_t) + (size_t)b * ps_pass * 8 + j
                             : bs_p + (size_t)b * per_sample_p * 8 + pass * 4 + (j - 4);
    const int n_t = j < 4 ? ps_pass : per_sample_p;
    double acc = 0.0;
    if (bs_d && pass == 1 && (j == 2 || j == 3)) {
      // the consistency / distillation sums of the student come from step_epilogue_kernel's per-block partials (block order)
      const double* bd = bs_d + (size_t)b * kEpiBlocks * 2 + (j - 2);
#pragma unroll 8
      for (int t = sub; t < kEpiBlocks; t += 32) acc += bd[(size_t)t * 2];
    } else {
#pragma unroll 8
    for (int t = sub; t < n_t; t += 32) acc += bs[(size_t)t * 8];  // independent loads: issue them together
    }
    if (ph && j < 2) {
      const double* bd = bs_ph + (size_t)b * per_sample_ph * 2 + j;
#pragma unroll 8
      for (int t = sub; t < per_sample_ph; t += 32) acc += bd[(size_t)t * 2];
    }
    if (pass == 1 && bs_sh != nullptr && j < 2) {  // --main_temporal: the student's sums likewise
      const double* bd = bs_sh + (size_t)b * per_sample_sh * 2 + j;
#pragma unroll 8
      for (int t = sub; t < per_sample_sh; t += 32) acc += bd[(size_t)t * 2];
    }
    s_part[tid] = acc;
    __syncthreads();
    if (tid < 8) {
      double a = 0.0;
      for (int k = 0; k < 32; ++k) a += s_part[k * 8 + tid];
      ps[((size_t)pass * B + b) * 8 + tid] = a;
    }
  } else if (bgP != nullptr) {
    const int b = bid - 2 * B;
    // 24 sums of per_sample partials: 10 threads per value (240 of 256), then a 10-term sum (fixed order)
    const int v = tid % 24, sub = tid / 24;
    double acc = 0.0;
    if (sub < 10) {
#pragma unroll 8
      for (int t = sub; t < per_sample_t; t += 10) acc += (double)bgP[((size_t)b * per_sample_t + t) * 24 + v];  // the teacher's gradient pass
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
      (f ? gT1 : gT0)[b * 16 + e] = (float)a;
    }
  }
  // ---- the last block to get here does the scalar epilogue.  Publishing relies on this order: (1) the workgroup barrier --
  // __syncthreads() is a workgroup-scope release/acquire, so every thread's stores above happen-before lane 0's fence (barrier
  // cumulativity); (2) ONE agent-scope release fence by lane 0 (an L2 write-back: cheap once per workgroup, 40 us when all 256
  // threads of 768 workgroups issued a __threadfence() each); (3) the ticket atomic; (4) an agent-scope acquire in the one
  // workgroup that continues.
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    s_last = atomicAdd(ticket, 1u);
  }
  __syncthreads();
  if (s_last != gridDim.x - 1) return;
  __threadfence();  // (the one workgroup that continues: acquire what the others published)
  // sums over samples, fixed order; slots 4, 5 (smoothness) are weighted by the sample's 1/(mean+1e-7).  One
  // thread per (pass, sample, slot) fetches and weighs its term (a single round trip for all of them), 16 threads
  // then add the B terms in sample order.
  constexpr int kMaxStepB = 64;  // beyond this the terms are re-fetched by the summing threads instead of staged in LDS
  __shared__ double s_term[2 * kMaxStepB * 8];
  const bool staged = B <= kMaxStepB;
  for (int i = tid; staged && i < 2 * B * 8; i += 256) {
    const int j = i & 7;
    const double* q = ps + (size_t)(i >> 3) * 8;
    double v = q[j];
    if (j == 4 || j == 5) {
      const float m = (float)(q[7] / (double)HW) + 1e-7f;
      v = v * (double)div_(1.0f, m);
    }
    s_term[i] = v;
  }
  // per-sample statistics of the smoothness gradient: mean and the mean-coupling term dot/(HW (mean+eps)^2)
  for (int s = tid; s < 2 * B; s += 256) {
    const double* q = ps + (size_t)s * 8;
    const double mean = q[7] / (double)HW;
    const double m = (double)((float)mean + 1e-7f);
    stats[s] = mean;
    stats[2 * B + s] = q[6] / ((double)HW * m * m);
  }
  __syncthreads();
  if (tid < 16) {
    const int pass = tid >> 3, j = tid & 7;
    double a = 0.0;
    for (int b = 0; b < B; ++b) {
      double v;
      if (staged) v = s_term[((size_t)pass * B + b) * 8 + j];
      else {
        const double* q = ps + ((size_t)pass * B + b) * 8;
        v = q[j];
        if (j == 4 || j == 5) v = v * (double)div_(1.0f, (float)(q[7] / (double)HW) + 1e-7f);
      }
      a += v;
    }
    sh_tot[pass][j] = a;
  }
  __syncthreads();
  if (tid != 0) return;
  if (noise_counter) *noise_counter += 1ull;  // every kernel of this step that reads it has finished (stream order)
  const double* st = sh_tot[0];
  const double* ss = sh_tot[1];
  const double N = (double)B * HW, Nx = (double)B * H * (W - 1), Ny = (double)B * (H - 1) * W;
  const double reproj_t = st[0] / (st[1] + 1e-7), reproj_s = ss[0] / (ss[1] + 1e-7);
  const double cons = ss[2] / N, distil = ss[3] / N;
  const double smooth_t = st[4] / Nx + st[5] / Ny, smooth_s = ss[4] / Nx + ss[5] / Ny;
  const float loss_t = (float)reproj_t + 1e-3f * (float)smooth_t;
  const float loss_m = ((float)reproj_s + (float)cons) + 1e-3f * (float)smooth_s;
  losses[0] = (float)reproj_t; losses[1] = (float)smooth_t; losses[2] = loss_t;
  losses[3] = (float)reproj_s; losses[4] = (float)cons; losses[5] = (float)smooth_s; losses[6] = (float)distil;
  losses[7] = loss_m;
  losses[8] = w_main * (loss_m + loss_t) + w_distil * (float)distil;
  if (loss_total) *loss_total = losses[8];
  losses[9] = (float)reproj_s + (float)reproj_t;            // "reproj_loss/0" after the mono losses are added in
  losses[10] = (loss_m + (float)distil) + loss_t;           // "loss/0" / "loss" without loss balancing
  losses[11] = loss_m + loss_t;                             // loss_list[0] with loss balancing
  for (int i = 12; i < kLossSlots; ++i) losses[i] = 0.f;
  coefs[0] = (float)((double)w_main / (st[1] + 1e-7));      // teacher reprojection map
  coefs[1] = (float)((double)w_main / (ss[1] + 1e-7));      // student reprojection map
  coefs[2] = (float)((double)w_main / N);                   // consistency map
  coefs[3] = (float)((double)w_distil / N);                 // distillation map
  coefs[4] = w_main * 1e-3f;                                // smoothness
}

// d total / d disp for both maps; block 0 also scales the pose gradients and runs the backward of
// transformation_from_parameters (pp.gT = the scaled gradients, pp.g_axisangle / g_translation nullable).  With the
// temporal hint (bgP != nullptr) the first B workgroups reduce their sample's pose partials first.
__global__ __launch_bounds__(256) void step_assemble_kernel(const float* G_r_t, const float* G_r_s, const float* G_cd,
                                                            const float* gn_t, const float* gn_s,
                                                            const float* coefs, const double* stats, const float* g_total,
                                                            int B, int HW, const float* gT0, const float* gT1,
                                                            float* gTs0, float* gTs1, float* g_disp_t, float* g_disp_s,
                                                            PoseParams pp, int pose_bwd, const float* bgP, const float* K,
                                                            int per_sample, int W, const float* bnd_t, const float* bnd_s,
                                                            int rows, int segs, float* fix_t, const float* G_e, float* g_ens,
                                                            const float* G_dual, int parts) {
  // parts (option "tail_overlap": the student's half is launched while the teacher's sweep still runs, the rest behind it):
  // bit 0 = the teacher's map, bit 1 = the student's map (and the ensemble head's), bit 2 = the pose gradients; 7 = everything
  const float g = g_total ? *g_total : 1.0f;
  const float cRt = coefs[0] * g, cRs = coefs[1] * g, cS = coefs[4] * g;  // coefs[2], [3] are already inside G_cd
  if (!(parts & 1)) { g_disp_t = nullptr; fix_t = nullptr; }
  if (!(parts & 2)) { g_disp_s = nullptr; g_ens = nullptr; }
  if (!(parts & 4)) {
    // (no pose work in this launch)
  } else if (bgP) {
    // temporal hint: the teacher's sweep ran in this backward call, its pose partials are still per task: workgroup b
    // reduces sample b's (step_final_kernel's pose branch), scales and runs that sample's pose backward
    if ((int)blockIdx.x < B) {
      __shared__ double s_part[256];
      __shared__ double s_gP[24];
      const int tid = threadIdx.x, b = blockIdx.x;
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
        (f ? gTs1 : gTs0)[b * 16 + e] = (float)a * cRt;
      }
      __syncthreads();
      if (pose_bwd && tid < 2) pose_bwd_one(pp, tid, b);
    }
  } else if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < B * 16; i += 256) { gTs0[i] = gT0[i] * cRt; gTs1[i] = gT1[i] * cRt; }
    __syncthreads();
    if (pose_bwd)
      for (int i = threadIdx.x; i < B * 2; i += 256) pose_bwd_one(pp, i / B, i % B);
  }
  // one image row per workgroup round: the row index is wave-uniform, so is the test for a segment boundary whose
  // gradient is completed by the neighbouring task's scratch row (one-row halo of the marching passes)
  const int H = HW / W;
  const bool vec4 = (W & 3) == 0;  // rows are then 16-byte aligned in every map
  for (int row = blockIdx.x; row < B * H; row += gridDim.x) {
    const int b = row / H, y = row - b * H;
    const float inv_t = div_(1.0f, (float)stats[b] + 1e-7f), inv_s = div_(1.0f, (float)stats[B + b] + 1e-7f);
    const float corr_t = (float)stats[2 * B + b], corr_s = (float)stats[3 * B + b];
    const float* brow_t = bnd_t ? march_boundary_row(bnd_t, b, y, H, W, rows, segs) : nullptr;
    const float* brow_s = bnd_s ? march_boundary_row(bnd_s, b, y, H, W, rows, segs) : nullptr;
    const size_t r0 = (size_t)row * W;
    auto px = [&](size_t i, int x) {
      if (g_disp_t) {
        const float G = brow_t ? G_r_t[i] + brow_t[x] : G_r_t[i];
        float v = fma_(cRt, G, cS * (gn_t[i] * inv_t - corr_t));
        if (G_dual) v += g * G_dual[i];  // --dual_distil: what the distillation term sends to the teacher's disparity
        g_disp_t[i] = v;
      } else if (fix_t && (brow_t || G_dual)) {  // the teacher's sweep finished its own rows (temporal hint): add the neighbour's term
        float v = brow_t ? cRt * brow_t[x] : 0.f;
        if (G_dual) v += g * G_dual[i];
        if (v != 0.f) fix_t[i] = fix_t[i] + v;
      }
      if (g_disp_s) {
        const float G = brow_s ? G_r_s[i] + brow_s[x] : G_r_s[i];
        g_disp_s[i] = fma_(cRs, G, fma_(g, G_cd[i], cS * (gn_s[i] * inv_s - corr_s)));
      }
      if (g_ens) g_ens[i] = g * G_e[i];  // --learn_ens: the loss weight is inside G_e already
    };
    if (vec4 && !brow_t && !brow_s && !g_ens && !G_dual) {
      // four pixels per thread, 16-byte accesses (no boundary row here: the arithmetic per element is px()'s)
      for (int x = threadIdx.x * 4; x < W; x += blockDim.x * 4) {
        const size_t i = r0 + x;
        if (g_disp_t) {
          const float4 G = *reinterpret_cast<const float4*>(G_r_t + i), n = *reinterpret_cast<const float4*>(gn_t + i);
          float4 o;
          o.x = fma_(cRt, G.x, cS * (n.x * inv_t - corr_t)); o.y = fma_(cRt, G.y, cS * (n.y * inv_t - corr_t));
          o.z = fma_(cRt, G.z, cS * (n.z * inv_t - corr_t)); o.w = fma_(cRt, G.w, cS * (n.w * inv_t - corr_t));
          *reinterpret_cast<float4*>(g_disp_t + i) = o;
        }
        if (g_disp_s) {
          const float4 G = *reinterpret_cast<const float4*>(G_r_s + i), n = *reinterpret_cast<const float4*>(gn_s + i);
          const float4 c = *reinterpret_cast<const float4*>(G_cd + i);
          float4 o;
          o.x = fma_(cRs, G.x, fma_(g, c.x, cS * (n.x * inv_s - corr_s))); o.y = fma_(cRs, G.y, fma_(g, c.y, cS * (n.y * inv_s - corr_s)));
          o.z = fma_(cRs, G.z, fma_(g, c.z, cS * (n.z * inv_s - corr_s))); o.w = fma_(cRs, G.w, fma_(g, c.w, cS * (n.w * inv_s - corr_s)));
          *reinterpret_cast<float4*>(g_disp_s + i) = o;
        }
      }
    } else {
      for (int x = threadIdx.x; x < W; x += blockDim.x) px(r0 + x, x);
    }
  }
}

}  // namespace mal

using namespace mal;

namespace mal {
// blockIdx.y = map k of `maps`: step number (counter ? *counter : step) * mult + k.  One thread per pixel GROUP (the four
// rows (y & ~3) .. + 3 of a column share one Philox block): consecutive threads = consecutive columns
struct NoiseMaps { float* out[4]; };
__global__ void tiebreak_noise_kernel(unsigned long long seed, unsigned long long step, const unsigned long long* counter,
                                      unsigned mult, int B, int H, int W, NoiseMaps maps) {
  const int H4 = (H + 3) >> 2;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)B * H4 * W) return;
  const int x = (int)(i % W), row = (int)(i / W), y4 = row % H4, b = row / H4, y0 = y4 * 4;
  const unsigned long long st = (counter ? *counter : step) * mult + blockIdx.y;
  float n[4];
  tie_noise4(seed, st, (unsigned)(b * H * W + y0 * W + x), n);
  float* o = maps.out[blockIdx.y] + (size_t)b * H * W + (size_t)y0 * W + x;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (y0 + j < H) o[(size_t)j * W] = n[j];
}
// n <= 4 noise maps in one launch (the multi-scale step draws one per scale: step number = step * mult + k)
int tiebreak_noise_launch(unsigned long long seed, unsigned long long step, const unsigned long long* counter, unsigned mult,
                          int n, int B, int H, int W, float* const* out, hipStream_t st) {
  NoiseMaps maps = {};
  for (int k = 0; k < n; ++k) maps.out[k] = out[k];
  const size_t px = (size_t)B * ((H + 3) / 4) * W;
  hipLaunchKernelGGL(tiebreak_noise_kernel, dim3((unsigned)((px + 255) / 256), n), dim3(256), 0, st, seed, step, counter, mult,
                     B, H, W, maps);
  return launch_status();
}
}  // namespace mal

extern "C" int mal_tiebreak_noise(uint64_t seed, uint64_t step, int B, int H, int W, float* out, void* stream) {
  int rc = check_shape(B, H, W);
  if (rc) return rc;
  if (!out) return MAL_EINVAL;
  return tiebreak_noise_launch(seed, step, nullptr, 1u, 1, B, H, W, &out, (hipStream_t)stream);
}

extern "C" size_t mal_step_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return carve_step(nullptr, B, H, W).bytes;
}

// MAL_STEP_TEXEL_INPUTS: the caller's three images ARE the (B,H,W,3) texel images (torch.channels_last): nothing is repacked
static void use_texel_inputs(const mal_step_args* a, StepWs& w) {
  if (a->flags & MAL_STEP_TEXEL_INPUTS) {
    w.packed[0] = const_cast<float*>(a->color0); w.packed[1] = const_cast<float*>(a->color_m1); w.packed[2] = const_cast<float*>(a->color_p1);
  }
}

static int step_check(const mal_step_args* a) {
  if (!a) return MAL_EINVAL;
  int rc = check_shape(a->B, a->H, a->W);
  if (rc) return rc;
  if (!a->color0 || !a->color_m1 || !a->color_p1 || !a->K || !a->inv_K || !a->disp_teacher || !a->disp_student ||
      !a->axisangle_m1 || !a->translation_m1 || !a->axisangle_p1 || !a->translation_p1 || !a->consistency_mask ||
      !a->augmentation_keep || !a->lowest_cost || !a->losses || !a->ws)
    return MAL_EINVAL;
  if (a->ws_bytes < carve_step(nullptr, a->B, a->H, a->W).bytes) return MAL_EWORKSPACE;
  if (a->ens_disp && (a->flags & MAL_STEP_NO_ENS)) return MAL_EINVAL;  // the learnt ensemble IS the third candidate
  if ((a->flags & MAL_STEP_DUAL_DISTIL) && !(a->flags & MAL_STEP_NO_ENS)) return MAL_EINVAL;  // upstream reads it on the two-way branch only
  if (a->g_ens_disp && !a->ens_disp) return MAL_EINVAL;
  // zero-copy texels: the caller's channels_last images are gathered as 12-byte texels -- a -DMAL_TEXEL_FLOATS=4 A/B build
  // strides its gathers by 16 bytes and must take the packed path (advisor, round 4)
  if ((a->flags & MAL_STEP_TEXEL_INPUTS) && kTexel != 3) return MAL_EINVAL;
  return MAL_OK;
}

// 1. identity term + texel packing of the three images (one texel load per pixel in the passes) + both smoothness
//    terms + the tie-break noise; B extra workgroups of the same launch: poses (frame -1 is inverted,
//    networks/repdepth.py:159-160) + camera block
static int first_sweep(const mal_step_args* a, const StepWs& w, hipStream_t st, int* per_sample_p) {
  const int B = a->B, H = a->H, W = a->W;
  StepPoses sp = {};
  sp.pose.B = B; sp.pose.F = 2;
  sp.pose.axisangle[0] = a->axisangle_m1; sp.pose.axisangle[1] = a->axisangle_p1;
  sp.pose.translation[0] = a->translation_m1; sp.pose.translation[1] = a->translation_p1;
  sp.pose.invert[0] = 1; sp.pose.invert[1] = 0;
  sp.pose.T[0] = w.T[0]; sp.pose.T[1] = w.T[1];
  sp.K = a->K; sp.invK = a->inv_K; sp.cam = w.cam; sp.ticket = w.ticket;
  TieNoise tn = {};
  if (a->flags & MAL_STEP_NOISE_PHILOX) {
    if (a->noise) return MAL_EINVAL;
    tn.on = 1; tn.seed = a->noise_seed; tn.step = a->noise_step; tn.counter = (const unsigned long long*)a->noise_counter;
    tn.noise_out = a->noise_out;
  }
  // both smoothness terms (teacher's and student's disparity) ride on this sweep: it holds the target rows anyway
  SmoothParams sm = {};
  sm.n = 2; sm.disp[0] = a->disp_teacher; sm.disp[1] = a->disp_student; sm.gn[0] = w.gn_t; sm.gn[1] = w.gn_s;
  sm.partials = w.bs_p; sm.dec[0] = a->dec_teacher; sm.dec[1] = a->dec_student;
  const bool tex = (a->flags & MAL_STEP_TEXEL_INPUTS) != 0;
  return pack_identity_launch(a->color0, a->color_m1, a->color_p1, B, H, W, tex ? nullptr : w.packed[1], tex ? nullptr : w.packed[2],
                              tex ? nullptr : w.packed[0], w.ident, st, &sp, &tn, &sm, per_sample_p, tex);
}

namespace mal { opt_t g_march_halo1{1}; }   // option "march_halo1": one-row halo of the step's gradient passes (0: two rows, A/B)

static MarchParams teacher_params(const mal_step_args* a, const StepWs& w, float* mono_reproj) {
  MarchParams p = march_params(a->B, a->H, a->W, a->min_depth, a->max_depth, 1e-7f, 0);
  p.disp = a->disp_teacher; p.K = a->K; p.invK = a->inv_K; p.T[0] = w.T[0]; p.T[1] = w.T[1];
  p.src[0] = w.packed[1]; p.src[1] = w.packed[2]; p.target = w.packed[0];
  p.min_reproj = mono_reproj; p.block_gP = w.bgP;
  p.cam = w.cam; p.cam_ready = 1;  // the first launch filled the camera block
  return p;
}

// the teacher's pass of the step without the temporal hint: forward and gradient in one sweep (the north-star kernel)
static int launch_teacher(const mal_step_args* a, const StepWs& w, float* mono_reproj, hipStream_t st) {
  MarchParams p = teacher_params(a, w, mono_reproj);
  p.ident = w.ident; p.noise = a->noise; p.g_reproj = w.G_r_t;
  p.block_sums = w.bs_t;
  p.bnd = g_march_halo1 ? w.bnd_t : nullptr;
  p.dbg = a->dec_teacher;
  return march_launch(p, MAL_F_AUTOMASK | MAL_F_GRAD | MAL_F_POSE_GRAD | MAL_F_SRC_PACKED | MAL_F_TGT_PACKED, st);
}

// the ensemble pass (no gradient): the averaged disparity is formed inside the kernel (trainer.py:594-600)
static int launch_ensemble(const mal_step_args* a, const StepWs& w, float* ens_reproj, hipStream_t st) {
  MarchParams p = march_params(a->B, a->H, a->W, a->min_depth, a->max_depth, 1e-7f, 0);
  p.disp = a->disp_teacher; p.disp2 = a->disp_student; p.K = a->K; p.invK = a->inv_K; p.T[0] = w.T[0]; p.T[1] = w.T[1];
  if (a->ens_disp) { p.disp = a->ens_disp; p.disp2 = nullptr; }  // --learn_ens: the head's disparity (trainer.py:596-597)
  p.src[0] = w.packed[1]; p.src[1] = w.packed[2]; p.target = w.packed[0];
  p.min_reproj = ens_reproj; p.block_sums = w.bs_e; p.block_gP = w.bgP_e;
  p.cam = w.cam; p.cam_ready = 1;
  return march_launch(p, MAL_F_SRC_PACKED | MAL_F_TGT_PACKED, st);
}

// the student's pass (trainer.py:592-612): with `epi` the consistency / distillation epilogue rides in it (the step without the
// temporal hint); without, it leaves its per-pixel min in multi_reproj and the final launch forms those terms later
static int launch_student(const mal_step_args* a, const StepWs& w, float* mono_reproj, float* ens_reproj, float* multi_reproj,
                          bool epi, hipStream_t st, int* per_sample) {
  const int B = a->B, H = a->H, W = a->W;
  MarchParams p = march_params(B, H, W, a->min_depth, a->max_depth, 1e-7f, 0);
  p.disp = a->disp_student; p.K = a->K; p.invK = a->inv_K; p.T[0] = w.T[0]; p.T[1] = w.T[1];
  p.src[0] = w.packed[1]; p.src[1] = w.packed[2]; p.target = w.packed[0];
  p.ext_mask = a->consistency_mask; p.sample_scale = a->augmentation_keep;
  p.sample_scale_is_mask = (a->flags & MAL_STEP_AUG_MASK) ? 1 : 0;
  p.mono_disp = a->disp_teacher; p.lowest_cost = a->lowest_cost; p.cmask_out = a->consistency_mask_out;
  p.min_reproj = multi_reproj; p.g_reproj = w.G_r_s;
  if (epi) {
    p.mono_reproj = mono_reproj; p.ens_reproj = ens_reproj;
    p.g_cons = w.G_c; p.g_distil = nullptr;  // one merged map: weights as coefs[2], coefs[3] of step_final_kernel
    p.ens_disp = a->ens_disp; p.g_ens = a->ens_disp ? w.G_e : nullptr;
    if (a->flags & MAL_STEP_DUAL_DISTIL) { p.dual_distil = 1; p.g_ens = w.G_e; }  // G_e = merge_distil * d distil / d disp_teacher
    p.merge_cons = (float)((double)a->w_main / ((double)B * H * W)); p.merge_distil = (float)((double)a->w_distil / ((double)B * H * W));
  }
  p.block_sums = w.bs_s; p.block_gP = w.bgP_e;  // (no pose gradient in this pass: the sink is never written)
  p.bnd = g_march_halo1 ? w.bnd_s : nullptr;
  p.cam = w.cam; p.cam_ready = 1;
  p.dbg = a->dec_student;
  int rc = march_launch(p, MAL_F_GRAD | (epi ? MAL_F_EPILOGUE : 0) | MAL_F_SRC_PACKED | MAL_F_TGT_PACKED, st);
  if (per_sample) *per_sample = p.strips * p.segs;
  return rc;
}
// --main_temporal: the student's forward pass in front of its producer -- warped images out; the min over the two warped
// candidates, its winner and the pass's weight (consistency x matching x (1 - augmentation): the four-way min does not change it)
// stay in the workspace, with the sums, for mal_loss_step_fwd
static int launch_student_warp(const mal_step_args* a, const StepWs& w, hipStream_t st) {
  MarchParams p = march_params(a->B, a->H, a->W, a->min_depth, a->max_depth, 1e-7f, 0);
  p.disp = a->disp_student; p.K = a->K; p.invK = a->inv_K; p.T[0] = w.T[0]; p.T[1] = w.T[1];
  p.src[0] = w.packed[1]; p.src[1] = w.packed[2]; p.target = w.packed[0];
  p.ext_mask = a->consistency_mask; p.sample_scale = a->augmentation_keep;
  p.sample_scale_is_mask = (a->flags & MAL_STEP_AUG_MASK) ? 1 : 0;
  p.mono_disp = a->disp_teacher; p.lowest_cost = a->lowest_cost; p.cmask_out = a->consistency_mask_out;
  p.min_reproj = w.rp_warp_s; p.min_reproj2 = a->multi_reproj ? a->multi_reproj : w.multi_reproj;
  p.argmin_out = w.arg_warp_s; p.argmin_out2 = w.arg_s; p.weight_out = w.w_s;
  p.color_out[0] = a->warp_s_m1; p.color_out[1] = a->warp_s_p1; p.color_out_stride = a->warp_sample_stride;
  p.block_sums = w.bs_s; p.block_gP = w.bgP_e;
  p.cam = w.cam; p.cam_ready = 1;
  return march_launch(p, MAL_F_SRC_PACKED | MAL_F_TGT_PACKED, st);
}
// ... and its gradient sweep in mal_loss_step_bwd: the four-way decisions of _fwd, what arrives through syn added
static int launch_student_temporal(const mal_step_args* a, const StepWs& w, hipStream_t st) {
  MarchParams p = march_params(a->B, a->H, a->W, a->min_depth, a->max_depth, 1e-7f, 0);
  p.disp = a->disp_student; p.K = a->K; p.invK = a->inv_K; p.T[0] = w.T[0]; p.T[1] = w.T[1];
  p.src[0] = w.packed[1]; p.src[1] = w.packed[2]; p.target = w.packed[0];
  p.g_reproj = w.G_r_s; p.block_sums = w.bs_e; p.block_gP = w.bgP_e;  // (the sums were taken by _fwd: a sink here)
  p.forced_w = w.w_s; p.forced_arg = w.arg_s; p.g_color[0] = a->g_warp_s_m1; p.g_color[1] = a->g_warp_s_p1;
  p.bnd = g_march_halo1 ? w.bnd_s : nullptr;
  p.cam = w.cam; p.cam_ready = 1;
  p.dbg = a->dec_student;
  return march_launch(p, MAL_F_GRAD | MAL_F_SRC_PACKED | MAL_F_TGT_PACKED, st);
}
static EpiParams epilogue_params(const mal_step_args* a, const StepWs& w, const float* mono_reproj, const float* ens_reproj,
                                 const float* multi_reproj) {
  const int B = a->B, H = a->H, W = a->W;
  MarchParams mp = march_params(B, H, W, a->min_depth, a->max_depth, 1e-7f, 0);
  EpiParams e = {};
  e.disp_s = a->disp_student; e.disp_t = a->disp_teacher; e.mono_reproj = mono_reproj; e.ens_reproj = ens_reproj;
  e.multi_reproj = multi_reproj; e.ext_mask = a->consistency_mask; e.lowest_cost = a->lowest_cost;
  e.sample_scale = a->augmentation_keep; e.scale_is_mask = (a->flags & MAL_STEP_AUG_MASK) ? 1 : 0;
  e.ens_disp = a->ens_disp; e.dual = (a->flags & MAL_STEP_DUAL_DISTIL) ? 1 : 0;
  e.min_disp = mp.min_disp; e.range = mp.range;
  e.merge_cons = (float)((double)a->w_main / ((double)B * H * W)); e.merge_distil = (float)((double)a->w_distil / ((double)B * H * W));
  e.G_c = w.G_c; e.G_e = (a->ens_disp || e.dual) ? w.G_e : nullptr; e.partials = w.bs_d; e.dbg = a->dec_student; e.B = B; e.HW = H * W;
  return e;
}

// Temporal hint: between the warp pass and the fused sweep the device runs the producer's small, latency-bound kernels
// (and whatever the segmenter does) -- the ensemble pass depends on none of it.  It is forked onto a side stream AFTER the
// warp pass (beside it, two ALU-bound passes only slow each other down: measured) and joined before the student pass,
// which reads its map.  Fork / join through events: capturable into the caller's HIP graph.
namespace mal { opt_t g_step_overlap{1}; }  // option "step_overlap"
// option "temporal_spec" (--temporal step): 1 = the pass in front of the producer is the teacher's GRADIENT pass (it
// exports the warped images as well) and the sweep after the producer's backward only corrects the tasks near the region
// map; 0 = forward-only pass in front, full gradient sweep behind (rounds 2-3, kept for same-box A/B)
namespace mal { opt_t g_temporal_spec{0}; }  // settable in -DMAL_EXPERIMENTS builds only (mal_set_option refuses it otherwise)
// One side stream (with its fork / join events) per (device, caller stream): two steps in flight on different streams of a
// device -- each with its own workspace -- then cannot consume or overwrite each other's pending join (round 3 kept one
// `pending` flag per device: step B's _warp cleared A's, and B's _fwd could return without waiting for its own ensemble
// pass).  `pending`: a fork on this caller stream whose join has not been enqueued yet.  The table is small and fixed; a
// caller stream that finds it full shares slot 0 of its device, and every _fwd of a forked step waits for the slot's join
// event whatever `pending` says, so sharing costs ordering, never correctness.  Lookups are serialised by a mutex; the
// launches themselves are the caller's (one thread per stream, as for every HIP stream).
// option "side_priority": 1 = the side stream is created with the device's LOWEST priority.  Measured: every kernel of the
// replayed step slows down -- 0.566 ms per step against 0.323 (profiles/r04_step_timelines.txt) --, so the default stays 0.
namespace mal { opt_t g_side_priority{0}; }
// sweep / tail (option "tail_overlap"): `sweep` is recorded on the caller's stream behind the fused sweep of a --temporal step;
// the producer's backward and the teacher's gradient sweep then run on the side stream behind it -- beside the epilogue and the
// reduction, which the caller's stream runs meanwhile -- and `tail` joins them back in front of the assembly
// (Measured, profiles/r05_tail_overlap.txt: a stream of its own for the chain, an anchor kernel behind the sweep that the chain
// then continues, the assembly on the chain's stream -- the graph runtime places the nodes on hardware queues by its own rules
// and none of these beat the plain form: the chain on the side stream, the assembly back on the caller's.  Nor does the rest
// of the assembly on the chain's stream: it then follows the sweep 5 us sooner, and the graph ends 19 us later.)
struct SideStream { hipStream_t caller; hipStream_t s; hipEvent_t fork, join, mid, sweep, tail; bool ok, init, pending, sweep_valid, tail_pending; int dev; };
static SideStream* side_stream(hipStream_t caller) {
  constexpr int kSlots = 64;
  static SideStream all[kSlots] = {};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lock(mu);
  SideStream* slot = nullptr;
  SideStream* first_of_dev = nullptr;
  for (int i = 0; i < kSlots && !slot; ++i) {
    if (all[i].init && all[i].dev == dev) {
      if (!first_of_dev) first_of_dev = &all[i];
      if (all[i].caller == caller) slot = &all[i];
    }
  }
  // never create a stream or an event while the caller's stream is being captured: share the device's first slot then
  // (it exists after the eager warm-up steps that precede every capture)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (!slot && first_of_dev && hipStreamIsCapturing(caller, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    slot = first_of_dev;
  (void)hipGetLastError();
  for (int i = 0; i < kSlots && !slot; ++i)
    if (!all[i].init) {
      slot = &all[i];
      slot->init = true; slot->dev = dev; slot->caller = caller; slot->pending = false;
      // lowest priority: what is forked here (ensemble + student marching passes) should yield wave slots to the producer
      // chain and the fused sweep on the caller's stream, which are the critical path (option "side_priority" 0: default priority)
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      slot->ok = (g_side_priority ? hipStreamCreateWithPriority(&slot->s, hipStreamNonBlocking, least)
                                  : hipStreamCreateWithFlags(&slot->s, hipStreamNonBlocking)) == hipSuccess &&
                 hipEventCreateWithFlags(&slot->fork, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&slot->join, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&slot->mid, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&slot->sweep, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&slot->tail, hipEventDisableTiming) == hipSuccess;
      slot->sweep_valid = slot->tail_pending = false;
      (void)hipGetLastError();
    }
  if (!slot) slot = first_of_dev;  // table full: share this device's first slot
  return (slot && slot->ok) ? slot : nullptr;
}
// the caller's stream waits for the forked ensemble pass; every path that leaves a step after the fork goes through here.
// `always`: the step KNOWS it forked (mal_loss_step_fwd of a forked step): wait for the slot's last recorded join even if
// another step that shares the slot has consumed the flag meanwhile.
// a tail that was begun (mal_loss_step_tail_begin) and never finished by mal_loss_step_bwd: whatever runs on the side stream
// is joined back, so that nothing of an abandoned backward is left in flight (or outside a stream capture)
static int join_tail(SideStream* ss, hipStream_t st) {
  if (!ss || !ss->tail_pending) return MAL_OK;
  ss->tail_pending = false;
  if (hipEventRecord(ss->tail, ss->s) != hipSuccess || hipStreamWaitEvent(st, ss->tail, 0) != hipSuccess) return MAL_ELAUNCH;
  return MAL_OK;
}
static int join_side(hipStream_t st, bool always = false) {
  SideStream* ss = side_stream(st);
  if (ss) { const int rc = join_tail(ss, st); if (rc) return rc; }
  if (!ss || (!ss->pending && !always)) return MAL_OK;
  ss->pending = false;
  return hipStreamWaitEvent(st, ss->join, 0) == hipSuccess ? MAL_OK : MAL_ELAUNCH;
}
namespace mal {
// the same side stream for the other whole-step lists (mal_step_ms.hip): side_begin forks it off `st` (nullptr: no side stream
// here -- the caller keeps everything on `st`), side_end records the join and marks it pending, side_wait joins it back
hipStream_t side_begin(hipStream_t st) {
  if (!g_step_overlap) return nullptr;
  SideStream* ss = side_stream(st);
  if (!ss) return nullptr;
  if (hipEventRecord(ss->fork, st) != hipSuccess || hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return ss->s;
}
int side_end(hipStream_t st) {
  SideStream* ss = side_stream(st);
  if (!ss || hipEventRecord(ss->join, ss->s) != hipSuccess) return MAL_ELAUNCH;
  ss->pending = true;
  return MAL_OK;
}
int side_wait(hipStream_t st, bool always) { return join_side(st, always); }
}
// what is forked beside the producer: the ensemble pass (unless --no_ens) and, with option "student_overlap" (default), the
// student's marching pass without its epilogue
namespace mal { opt_t g_student_overlap{1}; opt_t g_side_order{0}; }
// option "tail_overlap": 1 = a --temporal step's backward chain (producer's backward -> teacher's gradient sweep) runs on the
// library's side stream behind the FUSED SWEEP only, beside the epilogue and the reduction of the forward (in a captured graph
// the two chains are independent; eager: the same order as before, the side stream simply has nothing to wait for)
namespace mal { opt_t g_tail_overlap{1}; }
// option "sweeps_batched" (A/B): --temporal --main_temporal: the teacher's and the student's fused sweep as one launch
namespace mal { opt_t g_sweeps_batched{1}; }
static bool tail_applies(const mal_step_args* a) {
  return g_tail_overlap && g_step_overlap == 1 && !g_temporal_spec && (a->flags & (MAL_STEP_TEMPORAL | MAL_STEP_MAIN_TEMPORAL)) &&
         side_stream((hipStream_t)a->stream) != nullptr;
}
// Call between a --temporal step's forward and the producer's backward: *stream receives the stream the producer's backward
// should be enqueued on -- the library's side stream, made to wait for the fused sweep (the producer of d loss / d syn), when
// the option applies; args->stream otherwise.  mal_loss_step_bwd then enqueues the teacher's gradient sweep behind it on that
// stream and joins it back in front of the assembly.  mal_loss_step_tail_cancel: the caller did not use the stream after all.
extern "C" int mal_loss_step_tail_begin(const mal_step_args* a, void** stream) {
  if (!a || !stream) return MAL_EINVAL;
  *stream = a->stream;
  if (!tail_applies(a)) return MAL_OK;
  SideStream* ss = side_stream((hipStream_t)a->stream);
  if (!ss || !ss->sweep_valid || ss->tail_pending) return MAL_OK;
  ss->sweep_valid = false;  // one forward, one chain: a second backward of the same step (or an option switched on in between) runs in order
  if (hipStreamWaitEvent(ss->s, ss->sweep, 0) != hipSuccess) { (void)hipGetLastError(); return MAL_OK; }
  ss->tail_pending = true;
  *stream = ss->s;
  return MAL_OK;
}
extern "C" int mal_loss_step_tail_cancel(const mal_step_args* a) {
  if (!a) return MAL_EINVAL;
  SideStream* ss = side_stream((hipStream_t)a->stream);
  return join_tail(ss, (hipStream_t)a->stream);
}  // side_order 1 (student first) measured slower: 0.3265 vs 0.3200 ms -- the producer's small kernels starve beside a pass that holds every wave slot and all of the LDS
static bool side_forked(const mal_step_args* a) {
  return g_step_overlap && (a->flags & (MAL_STEP_TEMPORAL | MAL_STEP_MAIN_TEMPORAL)) && side_stream((hipStream_t)a->stream) != nullptr;
}
// (--main_temporal: the student's pass has its own producer to wait for; only the ensemble pass is forked then)
static bool student_forked(const mal_step_args* a) {
  return side_forked(a) && g_student_overlap && g_step_overlap == 1 && !(a->flags & MAL_STEP_MAIN_TEMPORAL);
}
static bool ensemble_forked(const mal_step_args* a) {
  return side_forked(a) && !(a->flags & MAL_STEP_NO_ENS);
}
// --temporal --main_temporal: the student's forward pass in front of ITS producer runs on the side stream, beside the teacher's
// producer chain (small, latency-bound kernels); mal_loss_step_student_ready orders the student's producer behind it
static bool student_warp_forked(const mal_step_args* a) {
  return side_forked(a) && g_step_overlap == 1 && (a->flags & MAL_STEP_TEMPORAL) && (a->flags & MAL_STEP_MAIN_TEMPORAL);
}

static int fork_ensemble(const mal_step_args* a, const StepWs& w, hipStream_t st) {
  SideStream* ss = side_stream(st);
  if (!ss) return MAL_ELAUNCH;
  if (hipEventRecord(ss->fork, st) != hipSuccess || hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess) return MAL_ELAUNCH;
  int rc = MAL_OK;
  if (student_warp_forked(a)) {
    rc = launch_student_warp(a, w, ss->s);
    if (hipEventRecord(ss->mid, ss->s) != hipSuccess && !rc) rc = MAL_ELAUNCH;
  }
  // option "side_order" 1: the student's pass first, the (lighter, forward-only) ensemble pass behind it -- the one that ends up
  // beside the fused sweep; 0: ensemble first
  if (g_side_order && student_forked(a))
    rc = launch_student(a, w, nullptr, nullptr, a->multi_reproj ? a->multi_reproj : w.multi_reproj, false, ss->s, nullptr);
  if (!rc && ensemble_forked(a)) rc = launch_ensemble(a, w, a->ens_reproj ? a->ens_reproj : w.ens_reproj, ss->s);
  if (!rc && !g_side_order && student_forked(a))
    rc = launch_student(a, w, nullptr, nullptr, a->multi_reproj ? a->multi_reproj : w.multi_reproj, false, ss->s, nullptr);
  if (hipEventRecord(ss->join, ss->s) != hipSuccess) return rc ? rc : MAL_ELAUNCH;
  ss->pending = true;  // whatever was enqueued on the side stream is joined before the step's buffers are reused
  return rc;
}

// A step that called mal_loss_step_warp but will not call mal_loss_step_fwd (its producer raised): join the forked
// ensemble pass, so that the next step's first launch cannot overwrite buffers the pass still reads and a stream
// capture does not end with unjoined work.
extern "C" int mal_loss_step_abort(const mal_step_args* a) {
  if (!a) return MAL_EINVAL;
  return join_side((hipStream_t)a->stream);
}

// MAL_STEP_MAIN_TEMPORAL: call between mal_loss_step_warp and the STUDENT's producer (after the teacher's, if any): the student's
// warped images (warp_s_*) may have been written on the library's side stream; args->stream waits for them here.  A no-op when
// they were written on args->stream itself.
extern "C" int mal_loss_step_student_ready(const mal_step_args* a) {
  if (!a) return MAL_EINVAL;
  if (!(a->flags & MAL_STEP_MAIN_TEMPORAL) || !student_warp_forked(a)) return MAL_OK;
  SideStream* ss = side_stream((hipStream_t)a->stream);
  if (!ss) return MAL_ELAUNCH;
  return hipStreamWaitEvent((hipStream_t)a->stream, ss->mid, 0) == hipSuccess ? MAL_OK : MAL_ELAUNCH;
}

// MAL_STEP_TEMPORAL, first call: the first sweep and the teacher's warped images (forward only: the per-pixel min over
// the two warped candidates and its winner stay in the workspace for mal_loss_step_fwd)
extern "C" int mal_loss_step_warp(const mal_step_args* a) {
  int rc = step_check(a);
  if (rc) return rc;
  const bool temporal = a->flags & MAL_STEP_TEMPORAL, main_t = a->flags & MAL_STEP_MAIN_TEMPORAL;
  if (!temporal && !main_t) return MAL_EINVAL;
  if (temporal && (!a->warp_m1 || !a->warp_p1)) return MAL_EINVAL;
  if (main_t && (!a->warp_s_m1 || !a->warp_s_p1)) return MAL_EINVAL;
  if ((a->warp2_m1 == nullptr) != (a->warp2_p1 == nullptr)) return MAL_EINVAL;
  StepWs w = carve_step(a->ws, a->B, a->H, a->W);
  use_texel_inputs(a, w);
  hipStream_t st = (hipStream_t)a->stream;
  rc = join_side(st);  // a previous step that was abandoned after its fork (no _fwd, no _abort)
  if (rc) return rc;
  int per_sample_p = 1;
  rc = first_sweep(a, w, st, &per_sample_p);
  if (rc) return rc;
  if (temporal) {
  MarchParams p = teacher_params(a, w, w.rp_warp);
  p.block_sums = w.bs_t;
  // ... with the automask: where no synthesised candidate can win, this IS the teacher's forward (min, winner, weight,
  // sums); the fused sweep of mal_loss_step_fwd overwrites the second copies where it re-decides
  p.ident = w.ident; p.noise = a->noise;
  int flags = MAL_F_AUTOMASK | MAL_F_SRC_PACKED | MAL_F_TGT_PACKED;
  if (g_temporal_spec) {
    // ... and its gradient: away from the region map nothing the producer does can change a decision or add a cotangent,
    // so the map, boundary rows and pose partials this pass leaves are final there (mal_loss_step_bwd redoes the rest)
    flags |= MAL_F_GRAD | MAL_F_POSE_GRAD;
    p.g_reproj = w.G_r_t;
    p.bnd = g_march_halo1 ? w.bnd_t : nullptr;
    p.dbg = a->dec_teacher;
  }
  p.min_reproj2 = a->mono_reproj ? a->mono_reproj : w.mono_reproj; p.argmin_out2 = w.arg_t; p.weight_out = w.w_t;
  p.color_out[0] = a->warp_m1; p.color_out[1] = a->warp_p1; p.argmin_out = w.arg_warp;
  p.color_out_stride = a->warp_sample_stride;
  p.color_out2[0] = a->warp2_m1; p.color_out2[1] = a->warp2_p1;
  rc = march_launch(p, flags, st);
  if (rc) return rc;
  }
  if (main_t && !student_warp_forked(a)) {
    rc = launch_student_warp(a, w, st);
    if (rc) return rc;
  }
  if ((ensemble_forked(a) || student_forked(a) || student_warp_forked(a)) && g_step_overlap == 1) {
    rc = fork_ensemble(a, w, st);
    if (rc) return rc;
  }
  return MAL_OK;
}

extern "C" int mal_loss_step_fwd(const mal_step_args* a) {
  int rc = step_check(a);
  if (rc) return rc;
  const int B = a->B, H = a->H, W = a->W;
  StepWs w = carve_step(a->ws, B, H, W);
  use_texel_inputs(a, w);
  hipStream_t st = (hipStream_t)a->stream;
  const bool no_ens = a->flags & MAL_STEP_NO_ENS, temporal = a->flags & MAL_STEP_TEMPORAL,
             main_t = a->flags & MAL_STEP_MAIN_TEMPORAL, hinted = temporal || main_t;
  float* mono_reproj = a->mono_reproj ? a->mono_reproj : w.mono_reproj;
  float* ens_reproj = no_ens ? nullptr : (a->ens_reproj ? a->ens_reproj : w.ens_reproj);
  float* multi_reproj = a->multi_reproj;  // only written when the caller wants the map

  int per_sample_p = 1, per_sample_ph = 0;
  int per_sample = 1, per_sample_t = 0;
  bool both_sweeps = false;
  {  // tasks per sample of the teacher's sums: its gradient pass, or (temporal hint) the forward pass in front of the producer
    int strips = 0, segs = 0;
    march_geometry(B, H, W, temporal && !g_temporal_spec ? 0 : MAL_F_GRAD, &strips, &segs, nullptr);
    per_sample_t = strips * segs;
  }
  if (hinted && ensemble_forked(a) && g_step_overlap == 2) {  // option "step_overlap" 2: the ensemble pass beside the sweeps of this call
    rc = fork_ensemble(a, w, st);
    if (rc) return rc;
  }
  if (!temporal) {
    if (main_t) per_sample_p = pack_identity_tasks_per_sample(H, W);  // (mal_loss_step_warp ran the first sweep)
    else rc = first_sweep(a, w, st, &per_sample_p);
    if (rc) return rc;
    // teacher pass: forward and gradient in one sweep
    rc = launch_teacher(a, w, mono_reproj, st);
    if (rc) { if (main_t) (void)join_side(st); return rc; }
  } else {
    // mal_loss_step_warp ran the first sweep and left min_f r(warp_f) / its winner in rp_warp / arg_warp: the two
    // synthesised images join the running min (first minimum wins, as torch.min over [warp-1, warp+1, syn-1, syn+1],
    // loss_utils.py:79-90,103), the automask and the teacher's sums are formed over all four, and the gradient w.r.t.
    // the synthesised images leaves unnormalised
    if (!a->syn_m1 || !a->syn_p1 || !a->g_syn_m1 || !a->g_syn_p1) { (void)join_side(st); return MAL_EINVAL; }
    // MAL_STEP_SYN_SPARSE: syn_* hold the synthesised images at the region pixels only; elsewhere they are the warped
    // images mal_loss_step_warp wrote (warp_*), which the sweep then reads instead
    const bool sparse = (a->flags & MAL_STEP_SYN_SPARSE) != 0;
    if (sparse && (!a->syn_region || !a->warp_m1 || !a->warp_p1)) { (void)join_side(st); return MAL_EINVAL; }
    per_sample_p = pack_identity_tasks_per_sample(H, W);
    // with --main_temporal too, both passes' sweeps (region-gated: a few short tasks each) go out as ONE launch, below
    both_sweeps = main_t && g_sweeps_batched && a->syn_region && a->syn_s_region;
    if (!both_sweeps)
    rc = photo_march_fused_more(w.packed[0], a->syn_m1, a->syn_p1, 2, w.ident, a->noise, w.rp_warp, w.arg_warp, B, H, W,
                                mono_reproj, w.arg_t, w.w_t, w.bs_ph, a->g_syn_m1, a->g_syn_p1, &per_sample_ph, st, a->syn_region,
                                a->g_syn_region_m1, a->g_syn_region_p1, w.order, w.ticket + 1,
                                sparse ? a->warp_m1 : nullptr, sparse ? a->warp_p1 : nullptr, (size_t)a->warp_sample_stride,
                                1 /* the target as texels: one 12-byte load instead of three planes */, 0);
    if (rc) { (void)join_side(st); return rc; }
  }
  int per_sample_sh = 0;
  if (main_t && both_sweeps) {
    if (!a->syn_s_m1 || !a->syn_s_p1 || !a->g_syn_s_m1 || !a->g_syn_s_p1) { (void)join_side(st); return MAL_EINVAL; }
    const bool sp_t = (a->flags & MAL_STEP_SYN_SPARSE) != 0, sp_s = (a->flags & MAL_STEP_SYN_S_SPARSE) != 0;
    if (sp_s && (!a->warp_s_m1 || !a->warp_s_p1)) { (void)join_side(st); return MAL_EINVAL; }
    FusedMoreArgs fa[2] = {};
    fa[0].target = w.packed[0]; fa[0].cand0 = a->syn_m1; fa[0].cand1 = a->syn_p1; fa[0].idx0 = 2; fa[0].ident = w.ident; fa[0].noise = a->noise;
    fa[0].prev_min = w.rp_warp; fa[0].prev_arg = w.arg_warp; fa[0].min_reproj = mono_reproj; fa[0].argmin = w.arg_t; fa[0].weight_out = w.w_t;
    fa[0].block_sums = w.bs_ph; fa[0].g_cand0 = a->g_syn_m1; fa[0].g_cand1 = a->g_syn_p1; fa[0].region = a->syn_region;
    fa[0].g_region0 = a->g_syn_region_m1; fa[0].g_region1 = a->g_syn_region_p1;
    fa[0].orig0 = sp_t ? a->warp_m1 : nullptr; fa[0].orig1 = sp_t ? a->warp_p1 : nullptr; fa[0].orig_stride = (size_t)a->warp_sample_stride;
    fa[1].target = w.packed[0]; fa[1].cand0 = a->syn_s_m1; fa[1].cand1 = a->syn_s_p1; fa[1].idx0 = 2; fa[1].ident = w.w_s; fa[1].noise = nullptr;
    fa[1].prev_min = w.rp_warp_s; fa[1].prev_arg = w.arg_warp_s; fa[1].min_reproj = a->multi_reproj ? a->multi_reproj : w.multi_reproj;
    fa[1].argmin = w.arg_s; fa[1].weight_out = w.w_s; fa[1].block_sums = w.bs_sh; fa[1].g_cand0 = a->g_syn_s_m1; fa[1].g_cand1 = a->g_syn_s_p1;
    fa[1].region = a->syn_s_region; fa[1].g_region0 = a->g_syn_s_region_m1; fa[1].g_region1 = a->g_syn_s_region_p1;
    fa[1].orig0 = sp_s ? a->warp_s_m1 : nullptr; fa[1].orig1 = sp_s ? a->warp_s_p1 : nullptr; fa[1].orig_stride = (size_t)a->warp_sample_stride;
    fa[1].weight_given = 1;  // `ident` is the weight
    int per = 0;
    rc = photo_march_fused_more_n(2, fa, B, H, W, &per, st);
    if (rc) { (void)join_side(st); return rc; }
    per_sample_ph = per_sample_sh = per;
  } else if (main_t) {
    // --main_temporal: the student's pair joins ITS running min the same way; the weight is the pass's mask (w_s), whatever wins
    if (!a->syn_s_m1 || !a->syn_s_p1 || !a->g_syn_s_m1 || !a->g_syn_s_p1) { (void)join_side(st); return MAL_EINVAL; }
    const bool sparse = (a->flags & MAL_STEP_SYN_S_SPARSE) != 0;
    if (sparse && (!a->syn_s_region || !a->warp_s_m1 || !a->warp_s_p1)) { (void)join_side(st); return MAL_EINVAL; }
    rc = photo_march_fused_more(w.packed[0], a->syn_s_m1, a->syn_s_p1, 2, w.w_s, nullptr, w.rp_warp_s, w.arg_warp_s, B, H, W,
                                a->multi_reproj ? a->multi_reproj : w.multi_reproj, w.arg_s, w.w_s, w.bs_sh, a->g_syn_s_m1,
                                a->g_syn_s_p1, &per_sample_sh, st, a->syn_s_region, a->g_syn_s_region_m1, a->g_syn_s_region_p1,
                                nullptr, nullptr, sparse ? a->warp_s_m1 : nullptr, sparse ? a->warp_s_p1 : nullptr,
                                (size_t)a->warp_sample_stride, 1, 1 /* `ident` is the weight */);
    if (rc) { (void)join_side(st); return rc; }
  }
  if (hinted) {  // what the backward chain waits for (mal_loss_step_tail_begin): the last fused sweep
    SideStream* ss = g_step_overlap ? side_stream(st) : nullptr;
    if (ss) ss->sweep_valid = tail_applies(a) && hipEventRecord(ss->sweep, st) == hipSuccess;
    (void)hipGetLastError();
  }
  // ensemble pass (no gradient); with the temporal hint it was forked beside the producer by mal_loss_step_warp -- and so
  // was the student's marching pass (without its epilogue)
  const bool ens_forked = hinted && ensemble_forked(a), stu_forked = temporal && student_forked(a);
  const bool stu_deferred = stu_forked || main_t;  // the student's marching pass has run: its epilogue is a launch of its own
  EpiParams epi = {};
  if (ens_forked || stu_forked || (hinted && student_warp_forked(a))) {
    rc = join_side(st, true);
    if (rc) return rc;
  }
  if (!no_ens && !ens_forked) {
    rc = launch_ensemble(a, w, ens_reproj, st);
    if (rc) return rc;
  }
  if (stu_deferred) {
    // the student's pass ran beside the producer: its consistency / distillation terms now, from the three per-pixel minima
    int strips = 0, segs = 0;
    march_geometry(B, H, W, main_t ? 0 : MAL_F_GRAD, &strips, &segs, nullptr);  // the pass that left the student's sums
    per_sample = strips * segs;
    epi = epilogue_params(a, w, mono_reproj, ens_reproj, a->multi_reproj ? a->multi_reproj : w.multi_reproj);
    hipLaunchKernelGGL(step_epilogue_kernel, dim3((unsigned)(B * kEpiBlocks)), dim3(256), 0, st, epi);
    rc = launch_status();
    if (rc) return rc;
  } else {
    // student pass with the consistency / distillation epilogue
    rc = launch_student(a, w, mono_reproj, ens_reproj, multi_reproj, true, st, &per_sample);
    if (rc) return rc;
  }
  // per-sample sums of both gradient passes, pose gradients (temporal: in _bwd, after the teacher's sweep), scalars
  hipLaunchKernelGGL(step_final_kernel, dim3(temporal ? 2 * B : 3 * B), dim3(256), 0, st, w.bs_t, w.bs_s, temporal ? nullptr : w.bgP,
                     w.bs_p, per_sample_p, temporal ? w.bs_ph : nullptr, per_sample_ph, main_t ? w.bs_sh : nullptr, per_sample_sh,
                     stu_deferred ? w.bs_d : nullptr, a->K,
                     per_sample, per_sample_t, B, H, W,
                     a->w_main, a->w_distil, w.ps, w.gT[0], w.gT[1], w.sm_stats, a->losses, w.coefs, a->loss_total,
                     w.ticket, (a->flags & MAL_STEP_NOISE_PHILOX) ? (unsigned long long*)a->noise_counter : nullptr);
  return launch_status();
}

// Measurement hook (bench.py): `launches` back-to-back launches of the teacher's pass exactly as mal_loss_step_fwd enqueues
// it for a step WITHOUT the temporal hint (same parameter block, same instantiation, same decomposition), and nothing else.
// The workspace must hold a finished mal_loss_step_fwd of the same arguments (texels, identity map, camera block); the
// launches overwrite what that call's teacher pass left with the same values.  Captured into a graph and replayed, this
// times the north-star kernel alone with two events outside the graph.
extern "C" int mal_loss_step_teacher_replay(const mal_step_args* a, int launches) {
  int rc = step_check(a);
  if (rc) return rc;
  if ((a->flags & (MAL_STEP_TEMPORAL | MAL_STEP_MAIN_TEMPORAL)) || launches <= 0 || launches > 4096) return MAL_EINVAL;
  StepWs w = carve_step(a->ws, a->B, a->H, a->W);
  use_texel_inputs(a, w);
  float* mono_reproj = a->mono_reproj ? a->mono_reproj : w.mono_reproj;
  for (int i = 0; i < launches; ++i) {
    rc = launch_teacher(a, w, mono_reproj, (hipStream_t)a->stream);
    if (rc) return rc;
  }
  return MAL_OK;
}

extern "C" int mal_loss_step_bwd(const mal_step_args* a) {
  int rc = step_check(a);
  if (rc) return rc;
  const int B = a->B, H = a->H, W = a->W, HW = H * W;
  StepWs w = carve_step(a->ws, B, H, W);
  use_texel_inputs(a, w);
  hipStream_t st = (hipStream_t)a->stream;
  int per_sample_t = 0;
  bool teacher_done = false;
  SideStream* tail = nullptr;  // option "tail_overlap": the sweeps below run on the side stream, behind the producers' backwards
  if (tail_applies(a)) {
    SideStream* ss = side_stream(st);
    if (ss && ss->tail_pending) tail = ss;
  }
  if (a->flags & MAL_STEP_TEMPORAL) {
    // the teacher's gradient sweep, with the decisions of the four-way min taken from _fwd and the gradient that
    // reaches the warped images through syn added before the chain rule through the warp
    if (!a->g_warp_m1 || !a->g_warp_p1) { if (tail) (void)join_tail(tail, st); return MAL_EINVAL; }
    MarchParams p = teacher_params(a, w, nullptr);
    p.g_reproj = w.G_r_t; p.block_sums = w.bs_t;
    p.ident = w.ident;  // unused by the TEMPORAL instantiation (the launch checks the flag combination only)
    p.forced_w = w.w_t; p.forced_arg = w.arg_t; p.g_color[0] = a->g_warp_m1; p.g_color[1] = a->g_warp_p1;
    if (g_temporal_spec) {
      // mal_loss_step_warp left the gradient as it is without the synthesised candidates: only the tasks near the region
      // map are redone (all of them without a map); the assembly finishes the teacher's gradient as in the plain step.
      // The sums of the redone tasks are not the step's (those are the pass's in front + the fused sweep's differences)
      p.region = a->syn_region; p.block_sums = w.bs_e;
    } else if (tail) {
      // beside the reduction: the loss scalars may not exist yet -- the sweep leaves the unnormalised map (the assembly finishes
      // it, as in the step without the hint) and its sums go to a scratch of their own (the reduction reads bs_t meanwhile)
      p.block_sums = w.bs_e;
    } else {
      // the loss scalars exist already: the sweep writes d total / d disp_teacher itself (the assembly does the student's)
      p.fin_gn = w.gn_t; p.fin_coefs = w.coefs; p.fin_stats = w.sm_stats; p.fin_g_total = a->g_total; p.fin_out = a->g_disp_teacher;
      teacher_done = a->g_disp_teacher != nullptr;
    }
    p.bnd = g_march_halo1 ? w.bnd_t : nullptr;
    p.dbg = a->dec_teacher;
    rc = march_launch(p, MAL_F_AUTOMASK | MAL_F_GRAD | MAL_F_POSE_GRAD | MAL_F_SRC_PACKED | MAL_F_TGT_PACKED, tail ? tail->s : st);
    if (tail && rc) (void)join_tail(tail, st);  // (nothing may be left outside the caller's stream)
    if (rc) return rc;
    per_sample_t = p.strips * p.segs;
  }
  if (a->flags & MAL_STEP_MAIN_TEMPORAL) {
    if (!a->g_warp_s_m1 || !a->g_warp_s_p1) { if (tail) (void)join_tail(tail, st); return MAL_EINVAL; }
    rc = launch_student_temporal(a, w, tail ? tail->s : st);
    if (tail && rc) (void)join_tail(tail, st);
    if (rc) return rc;
  }
  size_t g = (size_t)B * H;  // a workgroup per image row
  if (g > 4096) g = 4096;
  if (g < (size_t)B) g = (size_t)B;
  int segs = 0, rows = 0;
  march_geometry(B, H, W, MAL_F_GRAD, nullptr, &segs, &rows);  // the decomposition the gradient passes of this step used
  PoseParams pp = {};
  pp.B = B; pp.F = 2;
  pp.axisangle[0] = a->axisangle_m1; pp.axisangle[1] = a->axisangle_p1;
  pp.translation[0] = a->translation_m1; pp.translation[1] = a->translation_p1;
  pp.invert[0] = 1; pp.invert[1] = 0;
  pp.gT[0] = w.gTs[0]; pp.gT[1] = w.gTs[1];
  pp.g_axisangle[0] = a->g_axisangle_m1; pp.g_axisangle[1] = a->g_axisangle_p1;
  pp.g_translation[0] = a->g_translation_m1; pp.g_translation[1] = a->g_translation_p1;
  const int pose_bwd = (a->g_axisangle_m1 || a->g_translation_m1 || a->g_axisangle_p1 || a->g_translation_p1) ? 1 : 0;
  auto assemble = [&](int parts, hipStream_t on) {
    hipLaunchKernelGGL(step_assemble_kernel, dim3((unsigned)g), dim3(256), 0, on, w.G_r_t, w.G_r_s, w.G_c, w.gn_t,
                       w.gn_s, w.coefs, w.sm_stats, a->g_total, B, HW, w.gT[0], w.gT[1], w.gTs[0], w.gTs[1],
                       teacher_done ? nullptr : a->g_disp_teacher, a->g_disp_student, pp, pose_bwd, per_sample_t ? w.bgP : nullptr,
                       a->K, per_sample_t, W, g_march_halo1 ? w.bnd_t : nullptr, g_march_halo1 ? w.bnd_s : nullptr, rows, segs,
                       teacher_done ? a->g_disp_teacher : nullptr, a->ens_disp ? w.G_e : nullptr,
                       a->ens_disp ? a->g_ens_disp : nullptr, (a->flags & MAL_STEP_DUAL_DISTIL) ? w.G_e : nullptr, parts);
    return launch_status();
  };
  if (tail) {
    // the student's half needs nothing of the side stream's chain (unless its own sweep runs there: --main_temporal; and
    // --dual_distil adds a term of the epilogue to the TEACHER's map only): it goes out now, beside the teacher's sweep
    const bool split = !(a->flags & MAL_STEP_MAIN_TEMPORAL) && a->g_disp_student != nullptr;
    if (split) {
      rc = assemble(2, st);
      if (rc) { (void)join_tail(tail, st); return rc; }
    }
    rc = join_tail(tail, st);  // the sweep's results (and the producer's backward in front of it) come back to the caller's stream
    if (rc) return rc;
    return assemble(split ? 5 : 7, st);
  }
  return assemble(7, st);
}

// The min-reprojection / automask reduction over MATERIALISED candidate images and its backward, in the
// marching form of mal_march.hip (one wavefront per 64-column strip walking rows, DPP horizontal sums,
// running vertical sums, packed fp32 over colour pairs) -- the consumer side of the temporal hint
// (manydepth/loss_utils.py:84-101,152-160: the candidates are the warped sources plus ("syn", f, s)) and of
// every `fuse=False` route.  Candidates are processed two at a time (the pair structure of the fused pass):
//
//   forward   one launch per pair; a later pair continues the running min / argmin of the earlier ones
//             (first minimum wins, as torch.min), the last one forms the weight (automask x ext_mask),
//             the outputs and the per-task partial sums
//   backward  one launch per pair: d sum(rp*w) / d candidate for the pair's two images, from the argmin and
//             weight maps of the forward.  Only the winner of a pixel has SSIM/L1 partials, but a pixel
//             collects them from its 3x3 neighbours, whose winners differ: 18 partial planes per pair.
//
// Same numerics as march_kernel (SSIM on window sums scaled by 81^2; reassociation <= 1e-5 against the
// one-pixel-per-thread kernels of mal_photo.hip, which keep ATen's summation order and stay available:
// mal_set_option("photo_impl", 0); they also serve MAL_F_NO_SSIM / MAL_F_AVG).
#include "mal_common.h"
#include "mal_device.h"
#include "mal_march.h"  // FusedMoreArgs
#include "mal_pairs.h"

namespace mal {

struct PhotoMarchParams {
  const float* target; const float* cand[2];   // planar (B,3,H,W)
  int target_texels;                            // the target is (B,H,W,3) texels instead (request9)
  int weight_given;                             // FUSED: `ident` holds the pixel's WEIGHT (the student's mask, --main_temporal) instead of
                                                // the automask's identity term: the min changes, the weight does not
  int idx[2];                                   // the candidates' indices in the caller's list; idx[1] < 0: single candidate
  const float* ident; const float* noise; const float* ext_mask;
  const float* prev_min; const uint8_t* prev_arg;   // running min / argmin of earlier pairs (nullable)
  int last, automask;
  float* min_reproj; uint8_t* argmin; float* weight_out; double* block_sums;  // [task][2]
  // backward
  const uint8_t* argmin_in; const float* weight_in; const float* scale; const double* sums;
  float* g_cand[2];                              // (B,3,H,W) each, [1] nullable
  int B, H, W, strips, segs, rows, ntasks, per_xcd;
  // FUSED, nullable: (B,H,W) bytes, bit 0 = the pixel lies in the region the producer of this pair touched (elsewhere the
  // pair EQUALS the candidates behind prev_min / prev_arg).  A candidate whose 3x3 window holds no such pixel ties with
  // its original and the first minimum wins (torch.min): it is not evaluated there; a task without such a pixel within
  // two pixels of its tile copies the running min through and writes zero gradients.
  const uint8_t* region;
  // FUSED with a region map, nullable: a second copy of the gradient, written ONLY at region pixels -- the snapshot the
  // producer's in-place backward gathers from while it overwrites g_cand at those very pixels
  float* g_region[2];
  // FUSED with a region map: the order the tasks are dispatched in.  A classification launch (one wavefront per task: does
  // the producer's region come near its tile?) writes the tasks that do the full work to the front of `order` and the
  // others to its back (order_count[0..1] = how many of each, reset by the step's first launch); workgroup i then takes
  // task order[i], so the few expensive tasks are all resident from the start instead of each of three rounds of
  // workgroups waiting for its slowest member.  Which position a task gets depends on timing; what it computes does not.
  unsigned* order; unsigned* order_count;
  // FUSED with a region map, nullable (both or none): the candidates hold the pair ONLY at region pixels (bit 0 of the
  // byte); everywhere else the pixel is read from orig[f] -- the images the pair was made from, planar (3,H,W) per sample,
  // orig_stride floats between samples.  (The whole-step API: the pass in front of the producer then writes the warped
  // images once instead of twice, and the producer touches its regions only.)
  const float* orig[2]; size_t orig_stride;
};

struct Px9 { float t[3], a[3], c[3]; };

// the nine planes of pixel (row rr reflected, column gxr) of sample b
MAL_DEV void request9(const PhotoMarchParams& p, const float* tb, const float* ab, const float* cb, int HW, int row,
                      int gxr, Px9& q) {
  const unsigned bo = (unsigned)(row * p.W + gxr) * 4u;
  if (p.target_texels) {  // wave-uniform: the target as (B,H,W,3) texels (the step's packed copy / a channels_last input)
    const texel_t t = ldt(tb, bo * (unsigned)kTexel);
    q.t[0] = t.x; q.t[1] = t.y; q.t[2] = t.z;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) { q.a[ch] = ldf(ab + (size_t)ch * HW, bo); q.c[ch] = ldf(cb + (size_t)ch * HW, bo); }
    return;
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    q.t[ch] = ldf(tb + (size_t)ch * HW, bo); q.a[ch] = ldf(ab + (size_t)ch * HW, bo); q.c[ch] = ldf(cb + (size_t)ch * HW, bo);
  }
}

__global__ __launch_bounds__(64, 4) void photo_march_fwd_kernel(PhotoMarchParams p) {
  constexpr int HALO = 1, CW = 62;
  const int id = blockIdx.x;
  const int task = (id & 7) * p.per_xcd + (id >> 3);
  if (task >= p.ntasks) return;
  const int per_b = p.strips * p.segs;
  const int b = task / per_b, tt = task - b * per_b;
  const int seg = tt / p.strips, strip = tt - seg * p.strips;
  const int H = p.H, W = p.W, HW = H * W, lane = threadIdx.x;
  const int y_lo = seg * p.rows, y_hi = min(y_lo + p.rows, H);
  const int gx = strip * CW - HALO + lane;
  const bool in_x = gx >= 0 && gx < W;
  const int gxr = min(max(reflect1(gx, W), 0), W - 1);
  const bool out_x = in_x && lane >= HALO && lane < 64 - HALO;
  const float* tb = p.target + (size_t)b * 3 * HW;
  const float* ab = p.cand[0] + (size_t)b * 3 * HW;
  const float* cb = p.cand[1] + (size_t)b * 3 * HW;
  const size_t map_b = (size_t)b * HW;
  const bool single = p.idx[1] < 0;
  f2 hsA[9], hsB[9], hyA[2], hyB[2];
  float hzA[2], hzB[2];
#pragma unroll
  for (int i = 0; i < 9; ++i) { hsA[i] = bc(0.f); hsB[i] = bc(0.f); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { hyA[i] = bc(0.f); hyB[i] = bc(0.f); hzA[i] = 0.f; hzB[i] = 0.f; }
  f2 x1[3] = {bc(0.f), bc(0.f), bc(0.f)}, y1rg = bc(0.f);
  float y1b = 0.f, acc_rw = 0.f, acc_w = 0.f;
  auto row_of = [&](int rr) { return min(max(reflect1(rr, H), 0), H - 1); };
  const int r_first = max(y_lo - 1, -1);
  Px9 nxt;
  request9(p, tb, ab, cb, HW, row_of(r_first), gxr, nxt);
  for (int r = r_first; r <= y_hi; ++r) {
    const Px9 cur = nxt;
    request9(p, tb, ab, cb, HW, row_of(r + 1), gxr, nxt);
    const int c = r - 1;
    const bool c_own = c >= y_lo && c < y_hi;
    // the maps of the centre row, requested before the arithmetic of this row
    const unsigned go = (unsigned)(min(max(c, 0), H - 1) * W + gxr) * 4u;
    float pm = 0.f, idn = 0.f, nz = 0.f, em = 1.f;
    int pa = 0;
    if (c_own) {
      if (p.prev_min) { pm = ldf(p.prev_min + map_b, go); pa = p.prev_arg[map_b + (go >> 2)]; }
      if (p.last) {
        if (p.automask) { idn = ldf(p.ident + map_b, go); if (p.noise) nz = ldf(p.noise + map_b, go); }
        if (p.ext_mask) em = ldf(p.ext_mask + map_b, go);
      }
    }
    const f2 x0[3] = {(f2){cur.a[0], cur.a[1]}, (f2){cur.c[0], cur.c[1]}, (f2){cur.a[2], cur.c[2]}};
    const f2 y0rg = (f2){cur.t[0], cur.t[1]};
    const float y0b = cur.t[2];
    f2 h[9], hy[2];
    float hz[2];
    hy[0] = hsum3(y0rg); hy[1] = hsum3(y0rg * y0rg);
    hz[0] = hsum3(y0b);  hz[1] = hsum3(y0b * y0b);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f2 x = x0[k], y = k < 2 ? y0rg : bc(y0b);
      h[k * 3 + 0] = hsum3(x);
      h[k * 3 + 1] = hsum3(x * x);
      h[k * 3 + 2] = hsum3(x * y);
    }
    if (c_own) {  // wave-uniform
      const f2 syq = (hyA[0] + hy[0]) + hyB[0], syyq = (hyA[1] + hy[1]) + hyB[1];  // outer rows first (mal_march.hip)
      const float syz = (hzA[0] + hz[0]) + hzB[0], syyz = (hzA[1] + hz[1]) + hzB[1];
      const f2 vyq = fma2(-syq, syq, bc(9.0f) * syyq), d1yq = fma2(syq, syq, bc(kC1s));
      const float vyz = fma_(-syz, syz, 9.0f * syyz), d1yz = fma_(syz, syz, kC1s);
      f2 vc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f2 sx = (hsA[k * 3] + h[k * 3]) + hsB[k * 3], sxx = (hsA[k * 3 + 1] + h[k * 3 + 1]) + hsB[k * 3 + 1],
                 sxy = (hsA[k * 3 + 2] + h[k * 3 + 2]) + hsB[k * 3 + 2];
        const f2 v = ssim_sums2<false>(sx, k < 2 ? syq : bc(syz), sxx, k < 2 ? vyq : bc(vyz), k < 2 ? d1yq : bc(d1yz), sxy,
                                       nullptr, nullptr, nullptr);
        vc[k] = (f2){clamp01(v.x), clamp01(v.y)};
      }
      const f2 ssum = (f2){(vc[0].x + vc[0].y) + vc[2].x, (vc[1].x + vc[1].y) + vc[2].y};
      const f2 l0 = y1rg - x1[0], l1 = y1rg - x1[1], l2 = bc(y1b) - x1[2];
      const f2 lsum = (f2){(fabsf(l0.x) + fabsf(l0.y)) + fabsf(l2.x), (fabsf(l1.x) + fabsf(l1.y)) + fabsf(l2.y)};
      const f2 rr = bc(0.85f) * div3_2(ssum) + bc(0.15f) * div3_2(lsum);
      // running min, first minimum wins (torch.min)
      float rp = rr.x;
      int win = p.idx[0];
      if (p.prev_min) { rp = pm; win = pa; if (rr.x < rp) { rp = rr.x; win = p.idx[0]; } }
      if (!single && rr.y < rp) { rp = rr.y; win = p.idx[1]; }
      if (out_x) {
        if (p.min_reproj) stf(p.min_reproj + map_b, go, rp);
        if (p.argmin) p.argmin[map_b + (go >> 2)] = (uint8_t)win;
        if (p.last) {
          float w = 1.0f;
          if (p.automask) w = (rp <= idn + nz * 0.00001f) ? 1.0f : 0.0f;
          w *= em;
          if (p.weight_out) stf(p.weight_out + map_b, go, w);
          acc_rw += rp * w;
          acc_w += w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) { hsA[i] = hsB[i]; hsB[i] = h[i]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { hyA[i] = hyB[i]; hyB[i] = hy[i]; hzA[i] = hzB[i]; hzB[i] = hz[i]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) x1[k] = x0[k];
    y1rg = y0rg; y1b = y0b;
  }
  if (p.last) {
    const double r0 = wave_sum_d((double)acc_rw), r1 = wave_sum_d((double)acc_w);
    if (lane == 0) { p.block_sums[(size_t)task * 2] = r0; p.block_sums[(size_t)task * 2 + 1] = r1; }
  }
}

// FUSED (the temporal hint inside the whole-step list): the forward of the pair -- r of both candidates, the running min
// continued from prev_min / prev_arg, the automask -- is formed in the same sweep instead of being read back from the
// argmin / weight maps of an earlier launch.  Halo rows and lanes re-decide their pixels (same arithmetic as the owner:
// the window sums are direction-symmetric and these kernels do not flip), so the outputs must not alias prev_min /
// prev_arg.  The outputs (min_reproj / argmin / weight_out) ARRIVE holding the decision over the earlier candidates
// (the pass in front of the producer wrote them, with its sums): this sweep overwrites them only at the pixels it
// re-decides and leaves per-task DIFFERENCES of sum(rp*w), sum(w) -- a task the producer's region does not reach
// writes zero gradients, zero differences and nothing else.
// slot: the parameter block's index inside the kernel's argument (the batched launch: one block per scale)
template <bool FUSED>
MAL_DEV void photo_march_bwd_task(const PhotoMarchParams& p, const int task, const int slot = 0) {
  constexpr int HALO = 2, CW = 60;
  const int per_b = p.strips * p.segs;
  const int b = task / per_b, tt = task - b * per_b;
  const int seg = tt / p.strips, strip = tt - seg * p.strips;
  const int H = p.H, W = p.W, HW = H * W, lane = threadIdx.x;
  const int y_lo = seg * p.rows, y_hi = min(y_lo + p.rows, H);
  const int gx = strip * CW - HALO + lane;
  const bool in_x = gx >= 0 && gx < W;
  const int gxr = min(max(reflect1(gx, W), 0), W - 1);
  const bool out_x = in_x && lane >= HALO && lane < 64 - HALO;
  const float sL = gx == 0 ? 2.0f : 1.0f, sR = gx == W - 1 ? 2.0f : 1.0f;
  // Every access is (parameter-block pointer) + (32-bit per-lane byte offset that already holds the sample's and the plane's
  // base): with a pointer per sample and plane (rounds 2-4) the sweep formed 51 of its 82 addresses as 64-bit vector adds --
  // the scalar registers were exhausted -- in the kernel whose end gates the join of the --temporal step.  A (B,3,H,W) image
  // stays below 2^32 bytes (check_shape bounds a (B,1,H,W) map below 2^31; photo_march_fused_more checks orig_stride).
  const unsigned hw4 = (unsigned)HW * 4u;
  const unsigned img_b = (unsigned)b * 3u * hw4;   // sample b of a (B,3,H,W) image / (B,H,W,3) texel image, bytes
  const unsigned map_b4 = (unsigned)b * hw4;       // sample b of a (B,1,H,W) map, bytes
  const unsigned map_b = (unsigned)b * (unsigned)HW;  // ... of a (B,H,W) byte map
  const bool sparse = FUSED && p.region && p.orig[0];
  const unsigned org_b = sparse ? (unsigned)b * (unsigned)p.orig_stride * 4u : 0u;
  typedef __attribute__((address_space(4))) const PhotoMarchParams CPhoto;
  CPhoto* const kp0 = (CPhoto*)__builtin_amdgcn_kernarg_segment_ptr() + slot;  // (the kernel's only argument: block `slot` of it)
  float sc = p.scale ? *p.scale : 1.0f;
  if (p.sums) sc = (float)((double)sc / (p.sums[1] + 1e-7));
  f2 hsA[9], hsB[9], hyA[2], hyB[2], hcA[9], hcB[9];
  float hzA[2], hzB[2];
#pragma unroll
  for (int i = 0; i < 9; ++i) { hsA[i] = bc(0.f); hsB[i] = bc(0.f); hcA[i] = bc(0.f); hcB[i] = bc(0.f); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { hyA[i] = bc(0.f); hyB[i] = bc(0.f); hzA[i] = 0.f; hzB[i] = 0.f; }
  // raw values of rows r-1 and r-2 (L1 term of the centre row; own pixel of the gradient row)
  f2 x1[3] = {bc(0.f), bc(0.f), bc(0.f)}, x2[3] = {bc(0.f), bc(0.f), bc(0.f)}, y1rg = bc(0.f), y2rg = bc(0.f);
  float y1b = 0.f, y2b = 0.f;
  float w1 = 0.f;  // weight and winner of row c-1 = r-2
  int win1 = 255;
  float acc_rw = 0.f, acc_w = 0.f;  // FUSED: sum rp*w, sum w over the pixels this task owns
  auto row_of = [&](int rr) { return min(max(reflect1(rr, H), 0), H - 1); };
  const uint8_t* rgn0 = (FUSED && p.region) ? p.region + map_b : nullptr;
  if (FUSED && rgn0) {
    // does the producer's region come within two pixels of this tile?  (owned pixel <- decisions of its 3x3 neighbours <-
    // their 3x3 windows)
    unsigned any = 0u;  // every load issued before the first use: no short-circuit
#pragma unroll 8
    for (int rr = max(y_lo - 2, 0); rr <= min(y_hi + 1, H - 1); ++rr) any |= rgn0[rr * W + gxr];
    if (!__any((any & 1u) && in_x)) {
      if (out_x)
        for (int c = y_lo; c < y_hi; ++c) {
          const unsigned o0 = (unsigned)(c * W + gxr) * 4u + img_b, o1 = o0 + hw4, o2 = o1 + hw4;
          stf(p.g_cand[0], o0, 0.f); stf(p.g_cand[0], o1, 0.f); stf(p.g_cand[0], o2, 0.f);
          if (p.g_cand[1]) { stf(p.g_cand[1], o0, 0.f); stf(p.g_cand[1], o1, 0.f); stf(p.g_cand[1], o2, 0.f); }
        }
      if (lane == 0) { p.block_sums[(size_t)task * 2] = 0.0; p.block_sums[(size_t)task * 2 + 1] = 0.0; }
      return;
    }
  }
  float nfA = 0.f, nfB = 0.f;  // region within one column, of rows r-2 and r-1
  const int r_first = max(y_lo - HALO, -1), r_last = y_hi - 1 + HALO;
  Px9 nxt;
  auto request9o = [&](CPhoto& P, int row, Px9& q) {  // the nine planes of pixel (row, gxr) of sample b
    const unsigned bo = (unsigned)(row * W + gxr) * 4u;
    const unsigned o0 = bo + img_b, o1 = o0 + hw4, o2 = o1 + hw4;
    // (FUSED <=> the target is handed over as (B,H,W,3) texels -- the step's packed copy / a channels_last input --, checked by
    // the launcher: as a RUN-TIME branch it split the block, the offsets' zero-extension was hoisted across it, and instruction
    // selection no longer saw (scalar base + zext(32-bit offset)): the six candidate loads became 64-bit vector adds)
    if (FUSED) {
      const texel_t t = ldt(P.target, bo * (unsigned)kTexel + (unsigned)b * (unsigned)HW * (unsigned)(kTexel * 4));
      q.t[0] = t.x; q.t[1] = t.y; q.t[2] = t.z;
    } else {
      q.t[0] = ldf(P.target, o0); q.t[1] = ldf(P.target, o1); q.t[2] = ldf(P.target, o2);
    }
    q.a[0] = ldf(P.cand[0], o0); q.a[1] = ldf(P.cand[0], o1); q.a[2] = ldf(P.cand[0], o2);
    q.c[0] = ldf(P.cand[1], o0); q.c[1] = ldf(P.cand[1], o1); q.c[2] = ldf(P.cand[1], o2);
  };
  request9o(*kp0, row_of(r_first), nxt);
  float on_a[3] = {0.f, 0.f, 0.f}, on_c[3] = {0.f, 0.f, 0.f};  // sparse pair: the same pixel of the images it was made from
  auto request_orig = [&](CPhoto& P, int row) {
    const unsigned o0 = (unsigned)(row * W + gxr) * 4u + org_b, o1 = o0 + hw4, o2 = o1 + hw4;
    on_a[0] = ldf(P.orig[0], o0); on_a[1] = ldf(P.orig[0], o1); on_a[2] = ldf(P.orig[0], o2);
    on_c[0] = ldf(P.orig[1], o0); on_c[1] = ldf(P.orig[1], o1); on_c[2] = ldf(P.orig[1], o2);
  };
  if (sparse) request_orig(*kp0, row_of(r_first));
  unsigned fr_nxt = (FUSED && rgn0) ? rgn0[row_of(r_first) * W + gxr] : 1u;  // the region byte travels one row ahead, like the planes
  unsigned fr_m0 = 0u, fr_m1 = 0u;  // region bytes of rows r-1 and r-2
  float pm_nxt = 0.f, idn_nxt = 0.f;
  int win_nxt = 255;
  if (FUSED) {  // centre row of the first iteration: r_first - 1
    const unsigned g0 = (unsigned)(min(max(r_first - 1, 0), H - 1) * W + gxr);
    pm_nxt = ldf(p.prev_min, g0 * 4u + map_b4);
    win_nxt = *(p.prev_arg + (map_b + g0));
    idn_nxt = ldf(p.ident, g0 * 4u + map_b4);
    if (p.noise) idn_nxt += ldf(p.noise, g0 * 4u + map_b4) * 0.00001f;
  }
  for (int r = r_first; r <= r_last; ++r) {
    // the parameter block's pointers are re-read from the kernarg segment where they are used (scalar loads through a
    // pointer made opaque per iteration, as in march_body): held across the loop they exhaust the scalar registers and the
    // compiler falls back to 64-bit VECTOR addresses (two VALU instructions and two VGPRs per access)
    CPhoto* kp = kp0;
    asm volatile("" : "+s"(kp));
    CPhoto& P = *kp;
    float* const ga = P.g_cand[0];
    float* const gb = P.g_cand[1];
    const uint8_t* const rgn = (FUSED && P.region) ? P.region + map_b : nullptr;
    Px9 cur = nxt;
    const unsigned fr_q = fr_m1;  // region byte of the gradient row q = r-2
    fr_m1 = fr_m0; fr_m0 = fr_nxt;
    const unsigned fr_cur = fr_nxt;
    if (sparse && !(fr_cur & 1u)) {  // outside the region the pair IS the images it was made from
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) { cur.a[ch] = on_a[ch]; cur.c[ch] = on_c[ch]; }
    }
    request9o(P, row_of(r + 1), nxt);
    if (sparse) request_orig(P, row_of(r + 1));
    if (FUSED && P.region) fr_nxt = *(P.region + fresh(map_b + (unsigned)(row_of(r + 1) * W + gxr)));
    const int c = r - 1;
    const bool c_valid = c >= 0 && c < H && c >= y_lo - 1 && c <= y_hi;
    float w0 = 0.f;
    int win0 = 255;
    float pm = 0.f, idn = 0.f;  // FUSED: running min of the earlier candidates, identity term (+ noise) of the centre row
    if (FUSED) {
      // the centre row's maps were requested an iteration ago (a lone wave on a SIMD -- most tasks leave early when a
      // region map is given -- would otherwise stall a full memory round trip per row here)
      pm = pm_nxt; win0 = win_nxt; idn = idn_nxt;
      const int cn = min(max(r, 0), H - 1);  // next iteration's centre row (clamped: unused when not valid)
      const unsigned gn = (unsigned)(cn * W + gxr);
      pm_nxt = ldf(P.prev_min, gn * 4u + map_b4);
      win_nxt = *(P.prev_arg + (map_b + gn));
      idn_nxt = ldf(P.ident, gn * 4u + map_b4);
      if (P.noise) idn_nxt += ldf(P.noise, fresh(gn * 4u + map_b4)) * 0.00001f;
      if (!c_valid) { pm = 0.f; win0 = 255; idn = 0.f; }
    } else if (c_valid) {
      const unsigned go = (unsigned)(c * W + gxr);
      w0 = in_x ? ldf(P.weight_in, go * 4u + map_b4) : 0.f;  // not a pixel: contributes nothing
      win0 = *(P.argmin_in + (map_b + go));
    }
    const f2 x0[3] = {(f2){cur.a[0], cur.a[1]}, (f2){cur.c[0], cur.c[1]}, (f2){cur.a[2], cur.c[2]}};
    const f2 y0rg = (f2){cur.t[0], cur.t[1]};
    const float y0b = cur.t[2];
    f2 h[9], hy[2];
    float hz[2];
    hy[0] = hsum3(y0rg); hy[1] = hsum3(y0rg * y0rg);
    hz[0] = hsum3(y0b);  hz[1] = hsum3(y0b * y0b);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f2 x = x0[k], y = k < 2 ? y0rg : bc(y0b);
      h[k * 3 + 0] = hsum3(x);
      h[k * 3 + 1] = hsum3(x * x);
      h[k * 3 + 2] = hsum3(x * y);
    }
    // does the producer's region reach the 3x3 window of the centre row's pixel?  (rows r-2, r-1, r; without a region map: yes)
    bool near_c = true;
    if (FUSED && rgn) {
      const float fr = (fr_cur & 1u) ? 1.0f : 0.0f;
      const float nf = (dpp_shr1(fr) + fr) + dpp_shl1(fr);
      near_c = (nfA + nfB) + nf > 0.f;
      nfA = nfB; nfB = nf;
    }
    // ---- SSIM partials of the centre row c for the candidate that won there (if it is one of this pair)
    f2 coef[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) coef[i] = bc(0.f);
    if (c_valid) {  // wave-uniform
      const f2 syq = (hyA[0] + hy[0]) + hyB[0], syyq = (hyA[1] + hy[1]) + hyB[1];  // outer rows first (mal_march.hip)
      const float syz = (hzA[0] + hz[0]) + hzB[0], syyz = (hzA[1] + hz[1]) + hzB[1];
      const f2 vyq = fma2(-syq, syq, bc(9.0f) * syyq), d1yq = fma2(syq, syq, bc(kC1s));
      const float vyz = fma_(-syz, syz, 9.0f * syyz), d1yz = fma_(syz, syz, kC1s);
      f2 vraw[3], pa[3], pb[3], pc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f2 sx = (hsA[k * 3] + h[k * 3]) + hsB[k * 3], sxx = (hsA[k * 3 + 1] + h[k * 3 + 1]) + hsB[k * 3 + 1],
                 sxy = (hsA[k * 3 + 2] + h[k * 3 + 2]) + hsB[k * 3 + 2];
        vraw[k] = ssim_sums2<true>(sx, k < 2 ? syq : bc(syz), sxx, k < 2 ? vyq : bc(vyz), k < 2 ? d1yq : bc(d1yz), sxy,
                                   &pa[k], &pb[k], &pc[k]);
      }
      if (FUSED) {  // the forward of this pair at the centre row, exactly as photo_march_fwd_kernel forms it
        const f2 vc0 = (f2){clamp01(vraw[0].x), clamp01(vraw[0].y)}, vc1 = (f2){clamp01(vraw[1].x), clamp01(vraw[1].y)},
                 vc2 = (f2){clamp01(vraw[2].x), clamp01(vraw[2].y)};
        const f2 ssum = (f2){(vc0.x + vc0.y) + vc2.x, (vc1.x + vc1.y) + vc2.y};
        const f2 l0 = y1rg - x1[0], l1 = y1rg - x1[1], l2 = bc(y1b) - x1[2];
        const f2 lsum = (f2){(fabsf(l0.x) + fabsf(l0.y)) + fabsf(l2.x), (fabsf(l1.x) + fabsf(l1.y)) + fabsf(l2.y)};
        const f2 rr = bc(0.85f) * div3_2(ssum) + bc(0.15f) * div3_2(lsum);
        float rp = pm;  // running min, first minimum wins (torch.min)
        if (near_c) {
          if (rr.x < rp) { rp = rr.x; win0 = P.idx[0]; }
          if (rr.y < rp) { rp = rr.y; win0 = P.idx[1]; }
        }
        const float w = P.weight_given ? idn : ((rp <= idn) ? 1.0f : 0.0f);
        w0 = in_x ? w : 0.f;
        if (out_x && c >= y_lo && c < y_hi) {
          const unsigned go = (unsigned)(c * W + gxr);
          stf(P.min_reproj, go * 4u + map_b4, rp);
          *(P.argmin + (map_b + go)) = (uint8_t)win0;
          if (!P.weight_given) stf(P.weight_out, fresh(go * 4u + map_b4), w);
          const float w_old = P.weight_given ? idn : ((pm <= idn) ? 1.0f : 0.0f);  // what the pass over the earlier candidates decided and summed
          acc_rw += rp * w - pm * w_old;
          acc_w += w - w_old;
        }
      }
      const float kk = -w0 * (0.85f / 3.0f) * 0.5f;
      const float kk0 = win0 == P.idx[0] ? kk : 0.f, kk1 = win0 == P.idx[1] ? kk : 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        // torch.clamp passes gradient on [0,1] inclusive (clamped == raw)
        const f2 v = vraw[k];
        const float ka = k == 1 ? kk1 : kk0, kb = k == 0 ? kk0 : kk1;
        const f2 g = (f2){clamp01(v.x) == v.x ? ka : 0.f, clamp01(v.y) == v.y ? kb : 0.f};
        coef[k * 3 + 0] = g * pa[k]; coef[k * 3 + 1] = g * pb[k]; coef[k * 3 + 2] = g * pc[k];
      }
    }
    // ---- horizontal sums of the partial planes of row c
    f2 hc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const f2 vv = coef[i], l = vv * bc(sL), rr = vv * bc(sR);
      hc[i] = (f2){(dpp_shr1(l.x) + vv.x) + dpp_shl1(rr.x), (dpp_shr1(l.y) + vv.y) + dpp_shl1(rr.y)};
    }
    // ---- gradient row q = r-2
    const int q = r - 2;
    if (q >= y_lo && q < y_hi) {  // wave-uniform
      const float wyd = (q == H - 2) ? 2.0f : 1.0f;
      const float lw = w1 * (0.15f / 3.0f);
      const float lw0 = win1 == P.idx[0] ? lw : 0.f, lw1 = win1 == P.idx[1] ? lw : 0.f;
      const f2 lwk[3] = {bc(lw0), bc(lw1), (f2){lw0, lw1}};
      f2 g[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f2 SA = fma2(bc(wyd), hc[k * 3], hcA[k * 3]);
        const f2 SB = fma2(bc(wyd), hc[k * 3 + 1], hcA[k * 3 + 1]);
        const f2 SC = fma2(bc(wyd), hc[k * 3 + 2], hcA[k * 3 + 2]);
        const f2 xq = x2[k], yq = k < 2 ? y2rg : bc(y2b);
        const f2 df = xq - yq;
        const f2 sg = (f2){sgnf(df.x), sgnf(df.y)};
        g[k] = fma2(lwk[k], sg, fma2(SC, yq, fma2(SB, xq, SA))) * bc(sc);
      }
      if (out_x) {
        const unsigned o0 = (unsigned)(q * W + gxr) * 4u + img_b, o1 = o0 + hw4, o2 = o1 + hw4;
        stf(ga, o0, g[0].x); stf(ga, o1, g[0].y); stf(ga, o2, g[2].x);
        if (gb) { const unsigned q0 = fresh(o0), q1 = fresh(o1), q2 = fresh(o2); stf(gb, q0, g[1].x); stf(gb, q1, g[1].y); stf(gb, q2, g[2].y); }
        if (FUSED && P.g_region[0] && (fr_q & 1u)) {
          float* const ra = P.g_region[0];
          float* const rb = P.g_region[1];
          const unsigned q0 = fresh(o0), q1 = fresh(o1), q2 = fresh(o2);
          stf(ra, q0, g[0].x); stf(ra, q1, g[0].y); stf(ra, q2, g[2].x);
          stf(rb, q0, g[1].x); stf(rb, q1, g[1].y); stf(rb, q2, g[2].y);
        }
      }
    }
    // ---- rolls
    {
      const float wyu = (c == 0) ? 2.0f : 1.0f;
#pragma unroll
      for (int i = 0; i < 9; ++i) { hcA[i] = hcB[i] + hc[i]; hcB[i] = bc(wyu) * hc[i]; }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) { hsA[i] = hsB[i]; hsB[i] = h[i]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { hyA[i] = hyB[i]; hyB[i] = hy[i]; hzA[i] = hzB[i]; hzB[i] = hz[i]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { x2[k] = x1[k]; x1[k] = x0[k]; }
    y2rg = y1rg; y2b = y1b; y1rg = y0rg; y1b = y0b;
    w1 = w0; win1 = win0;
  }
  if (FUSED) {
    const double r0 = wave_sum_d((double)acc_rw), r1 = wave_sum_d((double)acc_w);
    if (lane == 0) { p.block_sums[(size_t)task * 2] = r0; p.block_sums[(size_t)task * 2 + 1] = r1; }
  }
}

template <bool FUSED>
__global__ __launch_bounds__(64, 2) void photo_march_bwd_kernel(PhotoMarchParams p) {
  const int id = blockIdx.x;
#ifdef MAL_EXPERIMENTS
  if (FUSED && p.order) {  // grid = ntasks exactly
    photo_march_bwd_task<FUSED>(p, (int)p.order[id]);
    return;
  }
#endif
  const int task = (id & 7) * p.per_xcd + (id >> 3);
  if (task >= p.ntasks) return;
  photo_march_bwd_task<FUSED>(p, task);
}

// The fused sweeps of several scales (the four-scale --temporal step: same shape, independent operands) as ONE launch,
// blockIdx.y = scale: each is region-gated -- a few short tasks on an otherwise idle chip -- and four of them one behind the
// other took 135 us where one takes 30-40.
struct PhotoMarchBatch { PhotoMarchParams s[MAL_MS_MAX_SCALES]; };
__global__ __launch_bounds__(64, 2) void photo_march_bwd_batch_kernel(PhotoMarchBatch q) {
  const int slot = blockIdx.y;
  const PhotoMarchParams& p = q.s[slot];
  const int id = blockIdx.x;
  const int task = (id & 7) * p.per_xcd + (id >> 3);
  if (task >= p.ntasks) return;
  photo_march_bwd_task<true>(p, task, slot);
}

#ifdef MAL_EXPERIMENTS  // option "syn_queue" (measured slower, LABBOOK.md 6): not in the default build
// one wavefront per task of the fused sweep's decomposition: the test photo_march_bwd_task<true> opens with
__global__ __launch_bounds__(64) void photo_march_classify_kernel(PhotoMarchParams p) {
  constexpr int HALO = 2, CW = 60;
  const int task = blockIdx.x;
  if (task >= p.ntasks) return;
  const int per_b = p.strips * p.segs;
  const int b = task / per_b, tt = task - b * per_b;
  const int seg = tt / p.strips, strip = tt - seg * p.strips;
  const int H = p.H, W = p.W, lane = threadIdx.x;
  const int y_lo = seg * p.rows, y_hi = min(y_lo + p.rows, H);
  const int gx = strip * CW - HALO + lane;
  const bool in_x = gx >= 0 && gx < W;
  const int gxr = min(max(reflect1(gx, W), 0), W - 1);
  const uint8_t* rgn = p.region + (size_t)b * H * W;
  unsigned any = 0u;
  for (int rr = max(y_lo - 2, 0); rr <= min(y_hi + 1, H - 1); ++rr) any |= rgn[rr * W + gxr];
  const bool active = __any((any & 1u) && in_x);
  if (lane == 0) {
    const unsigned pos = atomicAdd(p.order_count + (active ? 0 : 1), 1u);
    p.order[active ? pos : (unsigned)p.ntasks - 1u - pos] = (unsigned)task;
  }
}
#endif  // MAL_EXPERIMENTS

// ---- get_smooth_loss (manydepth/layers.py:210-223) on the mean-normalised disparity (loss_utils.py:119-121) in one
// sweep: lane = column, the right edge comes from the neighbouring lane (DPP), the up edge from the previous
// row in registers.  The per-sample 1/(mean+1e-7) is a positive constant, so it is factored out of |d n|: the
// sweep accumulates sum|dx d| w, sum|dy d| w, sum gn*d and sum d per task and writes the gradient w.r.t. the
// normalised map; a second small kernel forms the means, the loss and the mean-coupling term.
struct SmoothMarchParams {
  const float* disp; const float* img;  // (B,1,H,W), (B,3,H,W)
  float* gn;                            // (B,1,H,W) nullable
  double* partial;                      // [task][4]
  int B, H, W, strips, segs, rows, ntasks, per_xcd;
};

MAL_DEV void smooth_march_task(const SmoothMarchParams& p, int task) {
  constexpr int HALO = 1, CW = 62;
  const int per_b = p.strips * p.segs;
  const int b = task / per_b, tt = task - b * per_b;
  const int seg = tt / p.strips, strip = tt - seg * p.strips;
  const int H = p.H, W = p.W, HW = H * W, lane = threadIdx.x;
  const int y_lo = seg * p.rows, y_hi = min(y_lo + p.rows, H);
  const int gx = strip * CW - HALO + lane;
  const bool in_x = gx >= 0 && gx < W;
  const int gxr = min(max(gx, 0), W - 1);
  const bool out_x = in_x && lane >= HALO && lane < 64 - HALO;
  const float* d = p.disp + (size_t)b * HW;
  const float* im = p.img + (size_t)b * 3 * HW;
  const float nx = 1.0f / ((float)p.B * (float)H * (float)(W - 1)), ny = 1.0f / ((float)p.B * (float)(H - 1) * (float)W);
  float d1 = 0.f, c1[3] = {0.f, 0.f, 0.f}, g1 = 0.f;  // previous row: disparity, colour, pending gradient
  float acc_x = 0.f, acc_y = 0.f, acc_g = 0.f, acc_d = 0.f;
  // rows y_lo-1 (context for the up edge of y_lo) .. y_hi (finishes row y_hi-1 through its down edge; row H is
  // virtual: no edges)
  const int r0 = max(y_lo - 1, 0);
  for (int r = r0; r <= y_hi; ++r) {
    const bool row_ok = r < H;
    const unsigned bo = (unsigned)(min(r, H - 1) * W + gxr) * 4u;
    const float d0 = ldf(d, bo);
    const float c0[3] = {ldf(im, bo), ldf(im + HW, bo), ldf(im + 2 * (size_t)HW, bo)};
    // right edge (r,x)-(r,x+1)
    const float eR = (fabsf(c0[0] - dpp_shl1(c0[0])) + fabsf(c0[1] - dpp_shl1(c0[1]))) + fabsf(c0[2] - dpp_shl1(c0[2]));
    const float wx = (row_ok && in_x && gx + 1 < W) ? __expf(-(eR * (1.0f / 3.0f))) : 0.f;
    const float dfx = d0 - dpp_shl1(d0);
    const float sx = sgnf(dfx) * wx * nx;
    // up edge (r-1,x)-(r,x)
    const float eU = (fabsf(c1[0] - c0[0]) + fabsf(c1[1] - c0[1])) + fabsf(c1[2] - c0[2]);
    const bool have_up = r > r0;  // the first row of the sweep has no previous row in registers (and row 0 has none)
    const float wyu = (row_ok && have_up && in_x) ? __expf(-(eU * (1.0f / 3.0f))) : 0.f;
    const float dfy = d1 - d0;
    const float sy = sgnf(dfy) * wyu * ny;
    const int q = r - 1;  // row finished now
    if (q >= y_lo && q < y_hi && out_x) {
      const float g = g1 + sy;
      if (p.gn) stf(p.gn + (size_t)b * HW, (unsigned)(q * W + gxr) * 4u, g);
      acc_y += fabsf(dfy) * wyu;
      acc_g += g * d1;
    }
    if (r >= y_lo && r < y_hi && out_x) { acc_x += fabsf(dfx) * wx; acc_d += d0; }
    g1 = (sx - dpp_shr1(sx)) - sy;
    d1 = d0; c1[0] = c0[0]; c1[1] = c0[1]; c1[2] = c0[2];
  }
  const double s0 = wave_sum_d((double)acc_x), s1 = wave_sum_d((double)acc_y), s2 = wave_sum_d((double)acc_g),
               s3 = wave_sum_d((double)acc_d);
  if (lane == 0) { double* o = p.partial + (size_t)task * 4; o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; }
}

__global__ __launch_bounds__(64, 4) void smooth_march_kernel(SmoothMarchParams p) {
  const int id = blockIdx.x;
  const int task = (id & 7) * p.per_xcd + (id >> 3);
  if (task >= p.ntasks) return;
  smooth_march_task(p, task);
}

// several maps (of different sizes) in one launch: the multi-scale step sweeps 2 maps per scale, and the small ones are
// all launch latency on their own.  first[k] = first task of map k, first[n] = total.
constexpr int kSmoothBatch = 8;
struct SmoothMarchBatch { SmoothMarchParams m[kSmoothBatch]; int first[kSmoothBatch + 1]; int n, per_xcd; };
__global__ __launch_bounds__(64, 4) void smooth_march_batch_kernel(SmoothMarchBatch q) {
  const int id = blockIdx.x;
  const int task = (id & 7) * q.per_xcd + (id >> 3);
  if (task >= q.first[q.n]) return;
  int k = 0;
  while (k + 1 < q.n && task >= q.first[k + 1]) ++k;
  smooth_march_task(q.m[k], task - q.first[k]);
}

// per-sample sums over the tasks (contiguous), then loss, means, coupling terms: one workgroup, one wave per sample
// (16 at a time), the loss summed over the samples in sample order
__global__ __launch_bounds__(1024) void smooth_march_finish_kernel(const double* partial, int per_sample, int B, int H, int W,
                                                                   int normalise, double* stats, double* sums,
                                                                   double* loss_out) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, HW = H * W;
  for (int b = wv; b < B; b += 16) {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t = lane; t < per_sample; t += 64)
      for (int j = 0; j < 4; ++j) a[j] += partial[((size_t)b * per_sample + t) * 4 + j];
    for (int j = 0; j < 4; ++j) a[j] = wave_sum_d(a[j]);
    if (lane == 0) {
      const double mean = a[3] / (double)HW;
      const float m = normalise ? (float)mean + 1e-7f : 1.0f;
      const double inv = normalise ? (double)div_(1.0f, m) : 1.0;
      stats[b] = mean;
      stats[B + b] = normalise ? a[2] / ((double)HW * (double)m * (double)m) : 0.0;
      sums[2 * b] = a[0] * inv; sums[2 * b + 1] = a[1] * inv;
    }
  }
  __syncthreads();  // the per-sample sums, written to global memory by this workgroup, are visible to it
  if (threadIdx.x != 0) return;
  double sx = 0.0, sy = 0.0;
  for (int k = 0; k < B; ++k) { sx += sums[2 * k]; sy += sums[2 * k + 1]; }
  loss_out[0] = sx / ((double)B * H * (W - 1)) + sy / ((double)B * (H - 1) * W);
}

// partial: [>= B*tasks][4]; stats [2B]; sums [2B]
// the sweep alone: gn (unnormalised gradient map) and the per-task partials [task][4]; tasks of a sample are contiguous
int smooth_march_sweep(const float* disp, const float* img, int B, int H, int W, float* gn, double* partial,
                       hipStream_t st, int* per_sample) {
  SmoothMarchParams p = {};
  p.disp = disp; p.img = img; p.gn = gn; p.partial = partial; p.B = B; p.H = H; p.W = W;
  p.strips = (W + 61) / 62;
  p.rows = 12;
  p.segs = (H + p.rows - 1) / p.rows;
  p.ntasks = B * p.strips * p.segs;
  p.per_xcd = (p.ntasks + 7) / 8;
  if (per_sample) *per_sample = p.strips * p.segs;
  hipLaunchKernelGGL(smooth_march_kernel, dim3(p.per_xcd * 8), dim3(64), 0, st, p);
  return launch_status();
}
// tasks per sample of one (H, W) map in smooth_march_sweep_batch's decomposition
int smooth_march_batch_tasks(int H, int W) { return ((W + 61) / 62) * ((H + 11) / 12); }
// n <= 8 maps in one launch; per_sample[k] receives the tasks per sample of map k
int smooth_march_sweep_batch(int n, const float* const* disp, const float* const* img, int B, const int* H, const int* W,
                             float* const* gn, double* const* partial, hipStream_t st, int* per_sample) {
  if (n <= 0 || n > kSmoothBatch) return MAL_EINVAL;
  SmoothMarchBatch q = {};
  q.n = n;
  for (int k = 0; k < n; ++k) {
    SmoothMarchParams& p = q.m[k];
    p.disp = disp[k]; p.img = img[k]; p.gn = gn[k]; p.partial = partial[k]; p.B = B; p.H = H[k]; p.W = W[k];
    p.strips = (W[k] + 61) / 62;
    p.rows = 12;
    p.segs = (H[k] + p.rows - 1) / p.rows;
    p.ntasks = B * p.strips * p.segs;
    q.first[k + 1] = q.first[k] + p.ntasks;
    if (per_sample) per_sample[k] = smooth_march_batch_tasks(H[k], W[k]);  // (= p.strips * p.segs)
  }
  q.per_xcd = (q.first[n] + 7) / 8;
  hipLaunchKernelGGL(smooth_march_batch_kernel, dim3(q.per_xcd * 8), dim3(64), 0, st, q);
  return launch_status();
}
int smooth_march(const float* disp, const float* img, int B, int H, int W, int normalise, float* gn, double* partial,
                 double* stats, double* sums, double* loss_out, hipStream_t st) {
  int per_sample = 1;
  int rc = smooth_march_sweep(disp, img, B, H, W, gn, partial, st, &per_sample);
  if (rc) return rc;
  hipLaunchKernelGGL(smooth_march_finish_kernel, dim3(1), dim3(1024), 0, st, partial, per_sample, B, H, W, normalise,
                     stats, sums, loss_out);
  return launch_status();
}

opt_t g_photo_impl{1};  // 1 = marching kernels of this file; 0 = one-pixel-per-thread kernels of mal_photo.hip

static int device_slots() {
  static int slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    slots = cus * 4;  // SIMDs
  }
  return slots;
}

// option "syn_queue": the fused sweep dispatches the tasks that do the full work first (0: in task order).  Measured on
// MI355X (B=12 192x640, 19 % of the 4-row tasks active): the sweep itself 33.8 -> 21.3 us, but the classification launch
// costs 76 us (6336 atomics on two addresses) and even a free one would gain ~6 us: the sweep runs beside the ensemble
// pass, which then waits for the wave slots the expensive tasks hold from the start (0.347 -> 0.417 ms per step).  Off.
opt_t g_syn_queue{0};
opt_t g_syn_rows{4};  // option "syn_rows": rows per task of the fused sweep when a region map makes most tasks leave early
static void decompose(PhotoMarchParams& p, int cw, int waves_per_simd, int rows_min = 8) {
  p.strips = (p.W + cw - 1) / cw;
  int rows = rows_min;
  const long long cap = (long long)device_slots() * waves_per_simd;
  while (rows < p.H && (long long)p.B * p.strips * ((p.H + rows - 1) / rows) > cap) ++rows;
  p.rows = rows;
  p.segs = (p.H + rows - 1) / rows;
  p.ntasks = p.B * p.strips * p.segs;
  p.per_xcd = (p.ntasks + 7) / 8;
}

// forward over n_cand candidates; tmp_min / tmp_arg: scratch maps for the running min when the caller does
// not want min_reproj / argmin.  block_sums: [>= ntasks][2].  *ntasks_out tasks of the last launch.
int photo_march_fwd(const float* target, const float* const* cand, int n_cand, const float* ident, const float* noise,
                    const float* ext_mask, int B, int H, int W, int automask, float* min_reproj, uint8_t* argmin,
                    float* weight_out, double* block_sums, int* ntasks_out, hipStream_t st) {
  const int npairs = (n_cand + 1) / 2;
  if (npairs > 1 && (!min_reproj || !argmin)) return MAL_EINVAL;  // the running min travels in the output maps
  for (int pr = 0; pr < npairs; ++pr) {
    PhotoMarchParams p = {};
    p.target = target; p.B = B; p.H = H; p.W = W;
    p.cand[0] = cand[2 * pr]; p.idx[0] = 2 * pr;
    const bool has2 = 2 * pr + 1 < n_cand;
    p.cand[1] = has2 ? cand[2 * pr + 1] : cand[2 * pr]; p.idx[1] = has2 ? 2 * pr + 1 : -1;
    p.prev_min = pr ? min_reproj : nullptr; p.prev_arg = pr ? argmin : nullptr;
    p.last = pr == npairs - 1; p.automask = automask;
    p.ident = ident; p.noise = noise; p.ext_mask = ext_mask;
    p.min_reproj = min_reproj; p.argmin = argmin; p.weight_out = weight_out; p.block_sums = block_sums;
    decompose(p, 62, 4);
    *ntasks_out = p.ntasks;
    hipLaunchKernelGGL(photo_march_fwd_kernel, dim3(p.per_xcd * 8), dim3(64), 0, st, p);
  }
  return launch_status();
}

int photo_march_bwd(const float* target, const float* const* cand, int n_cand, const uint8_t* argmin, const float* weight,
                    const float* scale, const double* sums, int B, int H, int W, float* const* g_cand, hipStream_t st) {
  if ((long long)B * 3 * H * W * 4 >= (1ll << 32)) return MAL_ESHAPE;  // the sweep addresses (B,3,H,W) images with 32-bit byte offsets
  for (int pr = 0; pr < (n_cand + 1) / 2; ++pr) {
    PhotoMarchParams p = {};
    p.target = target; p.B = B; p.H = H; p.W = W;
    const bool has2 = 2 * pr + 1 < n_cand;
    int i0 = 2 * pr, i1 = has2 ? 2 * pr + 1 : -1;
    if (!g_cand[i0] && i1 >= 0 && g_cand[i1]) { const int t = i0; i0 = i1; i1 = t; }  // slot 0 always writes
    if (!g_cand[i0]) continue;
    p.cand[0] = cand[i0]; p.idx[0] = i0; p.g_cand[0] = g_cand[i0];
    p.cand[1] = i1 >= 0 ? cand[i1] : cand[i0]; p.idx[1] = i1; p.g_cand[1] = i1 >= 0 ? g_cand[i1] : nullptr;
    p.argmin_in = argmin; p.weight_in = weight; p.scale = scale; p.sums = sums;
    decompose(p, 60, 2);
    hipLaunchKernelGGL(photo_march_bwd_kernel<false>, dim3(p.per_xcd * 8), dim3(64), 0, st, p);
  }
  return launch_status();
}

// The temporal hint of the whole-step list in ONE sweep: candidates (cand0, cand1) = indices (idx0, idx0+1) join the
// running min prev_min / prev_arg (read-only; must not alias the outputs), automask against ident (+ noise); the outputs
// min_reproj / argmin / weight_out arrive holding the decision over the earlier candidates and are overwritten where this
// pair is re-decided; per-task partials [task][2] = DIFFERENCES of sum(rp*w), sum(w) against that earlier decision; and
// d sum(rp*w) / d candidate, unnormalised.  order (nullable): >= ntasks device words, order_count: two zeroed words.
static int fused_more_params(PhotoMarchParams& p, const float* target, const float* cand0, const float* cand1, int idx0,
                             const float* ident, const float* noise, const float* prev_min, const uint8_t* prev_arg, int B, int H,
                             int W, float* min_reproj, uint8_t* argmin, float* weight_out, double* block_sums, float* g_cand0,
                             float* g_cand1, const uint8_t* region, float* g_region0, float* g_region1, const float* orig0,
                             const float* orig1, size_t orig_stride, int target_texels, int weight_given);
int photo_march_fused_more(const float* target, const float* cand0, const float* cand1, int idx0, const float* ident,
                           const float* noise, const float* prev_min, const uint8_t* prev_arg, int B, int H, int W,
                           float* min_reproj, uint8_t* argmin, float* weight_out, double* block_sums, float* g_cand0,
                           float* g_cand1, int* per_sample_out, hipStream_t st, const uint8_t* region, float* g_region0,
                           float* g_region1, unsigned* order, unsigned* order_count, const float* orig0, const float* orig1,
                           size_t orig_stride, int target_texels, int weight_given) {
  PhotoMarchParams p = {};
  int rc = fused_more_params(p, target, cand0, cand1, idx0, ident, noise, prev_min, prev_arg, B, H, W, min_reproj, argmin, weight_out,
                             block_sums, g_cand0, g_cand1, region, g_region0, g_region1, orig0, orig1, orig_stride, target_texels,
                             weight_given);
  if (rc) return rc;
  *per_sample_out = p.strips * p.segs;
  unsigned grid = (unsigned)p.per_xcd * 8u;
#ifdef MAL_EXPERIMENTS
  if (region && order && order_count && g_syn_queue) {  // expensive tasks first (see PhotoMarchParams::order)
    p.order = order; p.order_count = order_count;
    grid = (unsigned)p.ntasks;
    hipLaunchKernelGGL(photo_march_classify_kernel, dim3(grid), dim3(64), 0, st, p);
  }
#else
  (void)order; (void)order_count;
#endif
  hipLaunchKernelGGL(photo_march_bwd_kernel<true>, dim3(grid), dim3(64), 0, st, p);
  return launch_status();
}

// n <= MAL_MS_MAX_SCALES fused sweeps of one shape in ONE launch (mal_march.h: FusedMoreArgs = the arguments above)
int photo_march_fused_more_n(int n, const FusedMoreArgs* a, int B, int H, int W, int* per_sample_out, hipStream_t st) {
  if (n < 1 || n > MAL_MS_MAX_SCALES) return MAL_EINVAL;
  PhotoMarchBatch q = {};
  for (int k = 0; k < n; ++k) {
    const FusedMoreArgs& x = a[k];
    const int rc = fused_more_params(q.s[k], x.target, x.cand0, x.cand1, x.idx0, x.ident, x.noise, x.prev_min, x.prev_arg, B, H, W,
                                     x.min_reproj, x.argmin, x.weight_out, x.block_sums, x.g_cand0, x.g_cand1, x.region, x.g_region0,
                                     x.g_region1, x.orig0, x.orig1, x.orig_stride, 1, x.weight_given);
    if (rc) return rc;
    if (q.s[k].ntasks != q.s[0].ntasks) return MAL_EINVAL;  // (a scale without a region map decomposes differently: not batched)
  }
  *per_sample_out = q.s[0].strips * q.s[0].segs;
  hipLaunchKernelGGL(photo_march_bwd_batch_kernel, dim3((unsigned)q.s[0].per_xcd * 8u, (unsigned)n), dim3(64), 0, st, q);
  return launch_status();
}

static int fused_more_params_impl(PhotoMarchParams& p, const float* target, const float* cand0, const float* cand1, int idx0,
                                  const float* ident, const float* noise, const float* prev_min, const uint8_t* prev_arg, int B,
                                  int H, int W, float* min_reproj, uint8_t* argmin, float* weight_out, double* block_sums,
                                  float* g_cand0, float* g_cand1, const uint8_t* region, float* g_region0, float* g_region1,
                                  const float* orig0, const float* orig1, size_t orig_stride, int target_texels, int weight_given) {
  if (prev_min == min_reproj || prev_arg == argmin) return MAL_EINVAL;
  if (weight_given && noise) return MAL_EINVAL;
  if ((orig0 == nullptr) != (orig1 == nullptr) || (orig0 && !region)) return MAL_EINVAL;
  if (!target_texels) return MAL_EINVAL;  // the fused sweep reads the target as texels (compile-time in the kernel)
  if ((long long)B * 3 * H * W * 4 >= (1ll << 32)) return MAL_ESHAPE;  // (B,3,H,W) images are addressed with 32-bit byte offsets
  if (orig0 && (long long)B * (long long)(orig_stride ? orig_stride : (size_t)3 * H * W) * 4 >= (1ll << 32)) return MAL_ESHAPE;  // 32-bit byte offsets
  p = PhotoMarchParams{};
  p.target = target; p.target_texels = target_texels; p.weight_given = weight_given; p.B = B; p.H = H; p.W = W;
  p.cand[0] = cand0; p.cand[1] = cand1; p.idx[0] = idx0; p.idx[1] = idx0 + 1;
  p.prev_min = prev_min; p.prev_arg = prev_arg; p.ident = ident; p.noise = noise;
  p.min_reproj = min_reproj; p.argmin = argmin; p.weight_out = weight_out; p.block_sums = block_sums;
  p.g_cand[0] = g_cand0; p.g_cand[1] = g_cand1; p.region = region;
  p.orig[0] = orig0; p.orig[1] = orig1; p.orig_stride = orig_stride ? orig_stride : (size_t)3 * H * W;
  if (region && g_region0 && g_region1) { p.g_region[0] = g_region0; p.g_region[1] = g_region1; }
  // with a region map the few tasks that do the full work set the kernel's duration (every task is resident at once, and
  // a wavefront alone on its SIMD marches no faster): shorter tasks, four times as many (the workspace holds 2-row tasks)
  if (region && g_syn_rows >= 2 && g_syn_rows < 8) decompose(p, 60, 8, g_syn_rows); else
  decompose(p, 60, 2);
  return MAL_OK;
}
static int fused_more_params(PhotoMarchParams& p, const float* target, const float* cand0, const float* cand1, int idx0,
                             const float* ident, const float* noise, const float* prev_min, const uint8_t* prev_arg, int B, int H,
                             int W, float* min_reproj, uint8_t* argmin, float* weight_out, double* block_sums, float* g_cand0,
                             float* g_cand1, const uint8_t* region, float* g_region0, float* g_region1, const float* orig0,
                             const float* orig1, size_t orig_stride, int target_texels, int weight_given) {
  return fused_more_params_impl(p, target, cand0, cand1, idx0, ident, noise, prev_min, prev_arg, B, H, W, min_reproj, argmin,
                                weight_out, block_sums, g_cand0, g_cand1, region, g_region0, g_region1, orig0, orig1, orig_stride,
                                target_texels, weight_given);
}

}  // namespace mal

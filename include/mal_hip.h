/*
 * mal_hip.h -- C ABI of libmal_hip.so: the MI355X (gfx950) kernels for MAL's photometric
 * reprojection + motion-aware loss hot path.
 *
 * The reference (YuejiangDong/MAL) has no FFI for this path: it sits behind plain Python
 * imports of manydepth/layers.py, manydepth/loss_utils.py and three Trainer methods
 * (manydepth/trainer.py:29-30,44).  These entry points are what a
 * ctypes binding in those modules calls instead of the chain of ATen ops; each one cites
 * the reference lines it replaces.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless the name says otherwise
 *     (u8 = uint8_t, f64 = double); images are NCHW; 4x4 matrices are row-major [B][16].
 *   - every function returns 0 (MAL_OK) or a negative MAL_E* code; nothing throws, nothing
 *     allocates, nothing synchronises: work is enqueued on `stream` (a hipStream_t passed
 *     as void*; NULL = the default stream) and scratch memory comes from the caller's
 *     workspace (`mal_workspace_bytes`).
 *   - `convention`: 0 = ManyDepth / DynamicDepth (grid = (u/(W-1)-0.5)*2 sampled with
 *     align_corners=True, manydepth/layers.py:192-194 + trainer.py:1122-1125);
 *     1 = DualRefine (grid = 2*(u+0.5)/W-1 sampled with align_corners=False,
 *     dualrefine/layers.py:224-225 + dualrefine/trainer.py:444-447).
 *   - F = number of source frames (2: frame ids -1, +1); arrays "[F]" are host arrays of F
 *     device pointers, copied by value at launch (F <= MAL_MAX_FRAMES).
 */
#ifndef MAL_HIP_H
#define MAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAL_VERSION 100 /* 0.1.0 */
#define MAL_MAX_FRAMES 2
#define MAL_MAX_CAND 4 /* warped -1,+1 and the temporal-hint "syn" -1,+1 (loss_utils.py:79-90) */
#define MAL_MAX_INSTANCES 64 /* matched instance masks per sample of the temporal-hint producer */

enum {
  MAL_OK = 0,
  MAL_EINVAL = -1,    /* bad argument (null pointer, non-positive size, F/n_cand out of range) */
  MAL_ESHAPE = -2,    /* image smaller than 2x2 (reflection pad needs >= 2) or too large for int32 indexing */
  MAL_EWORKSPACE = -3,/* workspace too small: call mal_workspace_bytes */
  MAL_ELAUNCH = -4,   /* hipLaunchKernel / hipGetLastError failed */
  MAL_ENODEVICE = -5  /* no HIP device */
};

/* flags for mal_pass_fused / mal_photo_* */
enum {
  MAL_F_AUTOMASK = 1,   /* mask = argmin([min_c r_c, ident + 1e-5*noise]) == 0  (loss_utils.py:105-110) */
  MAL_F_GRAD = 2,       /* also produce the (unnormalised) gradient maps */
  MAL_F_POSE_GRAD = 4,  /* ... and d/dT (teacher pass; the student pass detaches T, trainer.py:1107-1109) */
  MAL_F_NO_SSIM = 8,    /* --no_ssim: r = mean_c |t - p| (trainer.py:1217-1218) */
  MAL_F_AVG = 16,       /* --avg_reprojection (dualrefine/trainer.py:579-583): mean over candidates */
  MAL_F_EPILOGUE = 32,  /* student epilogue: consistency + distillation terms (loss_utils.py:193-254) */
  MAL_F_DUAL_DISTIL = 64,/* --dual_distil (loss_utils.py:232-234): the idx==0 target keeps its graph */
  MAL_F_SRC_PACKED = 128,/* mal_pass_fused: src[f] are texel copies made by mal_pack_texels */
  MAL_F_TGT_PACKED = 256 /* ... and so is target */
};

int mal_version(void);
const char* mal_strerror(int code);
/* 0 if a gfx950-class HIP device is present, else MAL_ENODEVICE. Never called by the loaders. */
int mal_device_check(void);

/* Scratch bytes the fused / warp-backward / photo entry points need for a (B,H,W) batch. */
size_t mal_workspace_bytes(int B, int H, int W);

/* ---- a1: manydepth/layers.py:14-23 ------------------------------------------------ */
int mal_disp_to_depth(const float* disp, size_t n, float min_depth, float max_depth,
                      float* scaled_out /*nullable*/, float* depth_out /*nullable*/, void* stream);
/* g_disp = g_scaled*(1/min-1/max) - g_depth*depth^2*(1/min-1/max) */
int mal_disp_to_depth_bwd(const float* disp, const float* g_scaled /*nullable*/, const float* g_depth /*nullable*/,
                          size_t n, float min_depth, float max_depth, float* g_disp, void* stream);

/* ---- a2: BackprojectDepth.forward, manydepth/layers.py:163-168 ---------------------- */
int mal_backproject(const float* depth /*B,1,H,W*/, const float* inv_K /*B,16*/, int B, int H, int W,
                    float* points /*B,4,HW*/, void* stream);
int mal_backproject_bwd(const float* g_points /*B,4,HW*/, const float* inv_K, int B, int H, int W,
                        float* g_depth /*B,1,H,W*/, void* stream);

/* ---- a3: Project3D.forward, manydepth/layers.py:184-199; dualrefine/layers.py:216-226 -- */
int mal_project3d(const float* points /*B,4,HW*/, const float* K, const float* T, int B, int H, int W,
                  float eps, int convention, float* grid /*B,H,W,2*/, float* z_out /*nullable B,1,H,W*/,
                  void* stream);
/* g_points (B,4,HW) and g_T (B,16, = K^T [g_P;0]); g_z nullable. ws: mal_workspace_bytes. */
int mal_project3d_bwd(const float* points, const float* K, const float* T, const float* g_grid,
                      const float* g_z /*nullable*/, int B, int H, int W, float eps, int convention,
                      float* g_points /*nullable*/, float* g_T /*nullable B,16*/, void* ws, size_t ws_bytes,
                      void* stream);

/* ---- a4: F.grid_sample(bilinear, border), trainer.py:1122-1125; dualrefine/trainer.py:444-447 */
int mal_grid_sample(const float* src /*B,C,H,W*/, const float* grid /*B,Ho,Wo,2*/, int B, int C, int H, int W,
                    int Ho, int Wo, int align_corners, float* out /*B,C,Ho,Wo*/, void* stream);
/* gradient wrt the grid only (the reference's sources carry no gradient) */
int mal_grid_sample_bwd(const float* src, const float* grid, const float* g_out, int B, int C, int H, int W,
                        int Ho, int Wo, int align_corners, float* g_grid /*B,Ho,Wo,2*/, void* stream);

/* ---- a7: SSIM.forward, manydepth/layers.py:243-257 ----------------------------------- */
int mal_ssim(const float* x, const float* y, int B, int C, int H, int W, float* out /*B,C,H,W*/, void* stream);
int mal_ssim_bwd(const float* x, const float* y, const float* g_out, int B, int C, int H, int W,
                 float* g_x /*nullable*/, float* g_y /*nullable*/, void* stream);

/* ---- a8: compute_reprojection_loss, manydepth/loss_utils.py:46-55 -------------------- */
int mal_reprojection_loss(const float* pred, const float* target, int B, int C, int H, int W, int no_ssim,
                          float* out /*B,1,H,W*/, void* stream);
int mal_reprojection_loss_bwd(const float* pred, const float* target, const float* g_out /*B,1,H,W*/,
                              int B, int C, int H, int W, int no_ssim,
                              float* g_pred /*nullable*/, float* g_target /*nullable*/, void* stream);

/* ---- a12: get_smooth_loss on mean-normalised disparity, layers.py:210-223 + loss_utils.py:119-121
 * normalise != 0: disp is divided by (mean_HW disp + 1e-7) per sample first.
 * loss_out[0] (f64, device) = mean_x-term + mean_y-term.  With g_disp != NULL also writes
 * d loss / d disp (exact, including the mean-normalisation coupling).                      */
int mal_smooth_loss(const float* disp /*B,1,H,W*/, const float* img /*B,C,H,W*/, int B, int C, int H, int W,
                    int normalise, double* loss_out, float* g_disp /*nullable*/, void* ws, size_t ws_bytes,
                    void* stream);

/* ---- a5 (materialising): Trainer.generate_images_pred, manydepth/trainer.py:1093-1125 --
 * depth = 1/(1/max + (1/min-1/max)*disp); for f<F: grid_f = Project3D(Backproject(depth), K, T_f);
 * warped_f = grid_sample(src_f, grid_f).  Any of the outputs may be NULL.                  */
/* `convention` of mal_warp_fwd/bwd may be OR'ed with MAL_WARP_TEXELS: src[f] are then the texel copies made by
 * mal_pack_texels (one gather per bilinear tap instead of three); results are bit-identical. */
#define MAL_WARP_TEXELS 256
int mal_warp_fwd(const float* disp /*B,1,H,W, full resolution*/, const float* K, const float* inv_K,
                 const float* const* T /*[F] B,16*/, const float* const* src /*[F] B,3,H,W*/,
                 int B, int H, int W, int F, float min_depth, float max_depth, float eps, int convention,
                 float* depth_out, float* const* grid_out /*[F] B,H,W,2*/, float* const* warped_out /*[F] B,3,H,W*/,
                 void* stream);
/* Backward of the above.  g_warped[f] / g_grid[f] / g_depth may each be NULL (absent).
 * g_disp (B,1,H,W) is overwritten; g_T[f] (B,16) nullable.                                */
int mal_warp_bwd(const float* disp, const float* K, const float* inv_K, const float* const* T,
                 const float* const* src, const float* const* g_warped, const float* const* g_grid,
                 const float* g_depth, int B, int H, int W, int F, float min_depth, float max_depth, float eps,
                 int convention, float* g_disp, float* const* g_T, void* ws, size_t ws_bytes, void* stream);

/* ---- a8-a10 on materialised candidates: the per-pixel min / automask reduction --------
 * cand[c] (B,3,H,W), c < n_cand <= MAL_MAX_CAND.  r_c = compute_reprojection_loss(cand_c, target);
 * rp = min_c r_c (first minimum wins, as torch.min on CPU) or the mean with MAL_F_AVG;
 * weight w = automask * ext_mask (each optional; ext_mask (B,1,H,W)).
 * ident: (B,1,H,W) min over the identity losses (mal_photo_fwd on the raw sources with
 *        flags=0 produces it); noise: (B,1,H,W) N(0,1) draws (CPU generator upstream).
 * Outputs: min_reproj (B,1,H,W, nullable), argmin_u8 (B,1,H,W, nullable),
 *          weight_out (B,1,H,W nullable), sums[0]=sum(rp*w), sums[1]=sum(w)  (f64, device).   */
int mal_photo_fwd(const float* target, const float* const* cand, int n_cand, const float* ident /*nullable*/,
                  const float* noise /*nullable*/, const float* ext_mask /*nullable*/, int B, int H, int W,
                  int flags, float* min_reproj, uint8_t* argmin_u8, float* weight_out, double* sums,
                  void* ws, size_t ws_bytes, void* stream);
/* d(sum(rp*w))/d cand_c -> g_cand[c] (B,3,H,W), scaled by *scale (device f32 scalar, nullable = 1)
 * divided by (sums[1] + 1e-7) when sums != NULL.  weight / argmin come from mal_photo_fwd.  */
int mal_photo_bwd(const float* target, const float* const* cand, int n_cand, const uint8_t* argmin_u8,
                  const float* weight, const float* scale, const double* sums, int B, int H, int W, int flags,
                  float* const* g_cand, void* stream);

/* ---- the fused pass: a2-a4 + a7-a10 (+ the a11 epilogue) in ONE launch ------------------
 * Replaces generate_images_pred + compute_mono_losses / compute_main_losses /
 * generate_images_pred_ensemble when the warped images need not be materialised
 * (manydepth/trainer.py:573-612, loss_utils.py:57-254).
 *
 * Inputs: disp (full res), K, inv_K, T[F], src[F], target; ident/noise (automask), ext_mask
 * (the student's consistency_mask*(1-augmentation_mask), expanded to B,1,H,W).
 * Epilogue inputs (MAL_F_EPILOGUE): mono_depth, mono_reproj, ens_reproj (nullable -> 2-way argmin).
 * Outputs:
 *   min_reproj (B,1,H,W, nullable)         per-pixel min_c r_c
 *   sums (f64[8], device): [0] sum(rp*w) [1] sum(w) [2] sum|d_multi-d_mono|*(1-w) [3] sum|distil-d_multi|*w
 *   with MAL_F_GRAD:
 *     g_reproj (B,1,H,W): d sum(rp*w) / d disp            (caller scales by g/(sums[1]+1e-7))
 *     g_cons, g_distil (B,1,H,W, MAL_F_EPILOGUE): d sums[2] / d disp, d sums[3] / d disp
 *     g_T[f] (B,16, MAL_F_POSE_GRAD): d sum(rp*w) / d T_f
 *   consistency_target (B,1,H,W nullable): 1/(mono*cmask + multi*(1-cmask)), loss_utils.py:212-215
 *   depth_out (B,1,H,W nullable)                                                               */
int mal_pass_fused(const float* disp, const float* disp2 /*nullable: disparity = (disp+disp2)/2, trainer.py:598*/,
                   const float* K, const float* inv_K, const float* const* T,
                   const float* const* src, const float* target, const float* ident, const float* noise,
                   const float* ext_mask, const float* sample_scale /*nullable (B): w *= scale[b], e.g.
                   1-augmentation_mask, loss_utils.py:194*/, const float* mono_depth, const float* mono_reproj,
                   const float* ens_reproj, int B, int H, int W, int F, float min_depth, float max_depth,
                   float eps, int convention, int flags, float* min_reproj, double* sums, float* g_reproj,
                   float* g_cons, float* g_distil, float* const* g_T, float* consistency_target,
                   float* depth_out, void* ws, size_t ws_bytes, void* stream);

/* ---- a11 epilogue on materialised depth (non-fused path): loss_utils.py:193-254 ---------
 * w = ext_mask. sums[2], sums[3] as above; g_multi_cons / g_multi_distil are d/d multi_depth
 * (unnormalised sums); g_mono_distil (nullable) is d sums[3] / d mono_depth for --dual_distil. */
int mal_distil_epilogue(const float* multi_depth, const float* mono_depth, const float* multi_reproj,
                        const float* mono_reproj, const float* ens_reproj /*nullable*/, const float* ext_mask,
                        int B, int H, int W, int flags, double* sums /*f64[8]*/, float* g_multi_cons,
                        float* g_multi_distil, float* g_mono_distil, float* consistency_target,
                        void* ws, size_t ws_bytes, void* stream);
/* ... with --learn_ens (loss_utils.py:240-241; trainer.py:596-600): the ensemble depth is ens_depth = disp_to_depth of the
 * learnt head's output instead of (mono + multi) / 2; g_ens_depth (nullable) = d sums[3] / d ens_depth (unnormalised),
 * non-zero where the ensemble wins the three-way min; ens_reproj is required. */
int mal_distil_epilogue_learned(const float* multi_depth, const float* mono_depth, const float* ens_depth,
                                const float* multi_reproj, const float* mono_reproj, const float* ens_reproj,
                                const float* ext_mask, int B, int H, int W, double* sums /*f64[8]*/, float* g_multi_cons,
                                float* g_multi_distil, float* g_ens_depth, float* consistency_target,
                                void* ws, size_t ws_bytes, void* stream);

/* (B,3,H,W) -> (B,H,W,n) texels, n = mal_texel_floats() (3: r,g,b 12 bytes apart): each bilinear tap of the fused
 * pass becomes one per-lane gather instead of three 4-byte ones.  Sources change per batch, not per pass: pack once.
 * dst holds B*H*W*n floats. */
int mal_texel_floats(void);
int mal_pack_texels(const float* src, int B, int H, int W, float* dst, void* stream);

/* ---- a13: Trainer.compute_matching_mask, manydepth/trainer.py:1066-1076 ----------------
 * out = consistency_mask * [ (1/lowest_cost - mono)/mono < 1 ] * [ (mono - 1/lowest_cost)*lowest_cost < 1 ] */
int mal_matching_mask(const float* lowest_cost /*B,H,W*/, const float* mono_depth /*B,1,H,W*/,
                      const float* consistency_mask /*nullable B,H,W*/, size_t n, float* out, void* stream);

/* ---- gradient assembly: out[i] = sum_k coef_k * maps[k][i], coef_k = (scale_k ? *scale_k : 1) *
 * mult_k / (denom_k ? (*denom_k + denom_eps_k) : 1); all scalars are read on the device so the
 * autograd backward needs no host sync.  n_terms <= 6.                                        */
int mal_axpy_maps(int n_terms, const float* const* maps, const float* const* scale, const double* const* denom,
                  const float* mult, const float* denom_eps, size_t n, float* out, int accumulate, void* stream);

/* f64 device scalar -> f32 device scalars: out[k] = num[k] / (den ? den[k] + eps : 1) */
int mal_finish_scalars(const double* num, const double* den /*nullable*/, float eps, float mult, int n,
                       float* out, void* stream);

/* sum of a float array into a f64 device scalar (the student's sum(m), known before the pass) */
int mal_sum_f64(const float* x, size_t n, double* out, void* ws, size_t ws_bytes, void* stream);

/* ---- a16: transformation_from_parameters, manydepth/layers.py:26-100 -----------------------
 * axisangle[f], translation[f]: (B,3) (the reference's (B,1,3)); invert[f] (host ints): frame ids
 * < 0 are inverted (networks/repdepth.py:159-160).  T[f]: (B,16).  One launch for all f, b. */
int mal_pose_fwd(const float* const* axisangle, const float* const* translation, const int* invert,
                 int B, int F, float* const* T, void* stream);
int mal_pose_bwd(const float* const* axisangle, const float* const* translation, const int* invert,
                 const float* const* g_T, int B, int F, float* const* g_axisangle /*[F] nullable each*/,
                 float* const* g_translation /*[F] nullable each*/, void* stream);

/* ---- the whole loss half of process_batch (manydepth/trainer.py:573-642 with --distil) as one call.
 * Forward enqueues ~18 kernels (pose composition, texel packing, identity term, teacher / ensemble /
 * student passes with the matching mask and the consistency*(1-augmentation) mask formed in the
 * kernel, both smoothness terms, reductions); backward enqueues 2 (gradient assembly, pose backward).
 * losses[16] (device f32): 0 reproj_t 1 smooth_t 2 loss_t(=0+1e-3*1) 3 reproj_s 4 consistency 5 smooth_s
 *   6 distil 7 loss_main(=3+4+1e-3*5) 8 total = w_main*(7+2) + w_distil*6   (what backward differentiates)
 *   9 reproj_s+reproj_t ("reproj_loss/0") 10 (7+6)+2 ("loss" without --loss_blc) 11 7+2 (loss_list[0])
 * Without --loss_blc pass w_main = w_distil = 1; with it w_main = bs*w0, w_distil = bs*w1
 * (loss_utils.py:303-318).  All maps the step needs between kernels live in `ws`
 * (mal_step_workspace_bytes); the same ws must be handed to _bwd. */
enum {
  MAL_STEP_NO_ENS = 1,   /* --no_ens: 2-way distillation argmin, no ensemble pass */
  MAL_STEP_AUG_MASK = 2, /* `augmentation_keep` holds augmentation_mask itself; 1 - mask is formed on the device */
  MAL_STEP_TEMPORAL = 8, /* --temporal (loss_utils.py:84-88): r(syn_f, target) of the two synthesised images joins the
                            teacher's per-pixel min.  Three calls around the producer (dyn_utils.image_synthesis):
                              mal_loss_step_warp  first sweep + the teacher's warped images -> warp_m1 / warp_p1
                              [syn_f = producer(warp_f)]
                              mal_loss_step_fwd   syn_m1 / syn_p1 in; losses out; g_syn_* out = d sum(rp*w) / d syn_f
                                                  (UNNORMALISED: the 1/(sum w + 1e-7) and loss weights are applied in _bwd)
                              [g_warp_f = producer^T(g_syn_f), linear]
                              mal_loss_step_bwd   g_warp_* in (same scale); runs the teacher's gradient sweep with the
                                                  four-way decisions of _fwd and adds what arrives through syn */
  MAL_STEP_NOISE_PHILOX = 4, /* the automask tie-break noise (loss_utils.py:105-106: + 1e-5 * randn) is drawn inside the
                               step's first kernel: N(0,1) by Box-Muller from Philox4x32-10 keyed by noise_seed, counter =
                               (pixel, step); `noise` must be NULL.  No host RNG, no device RNG launch on the step. */
  MAL_STEP_TEXEL_INPUTS = 64, /* color0 / color_m1 / color_p1 are (B,H,W,3) texel images -- a (B,3,H,W) tensor in torch.channels_last
                                memory format is exactly that --: the step gathers from them directly and skips the re-layout of
                                its first sweep (53 MB of stores per step at B=12 192x640); results are bit-identical */
  MAL_STEP_DUAL_DISTIL = 32, /* --dual_distil (loss_utils.py:231-234; only with MAL_STEP_NO_ENS, as upstream: the flag is read on the
                               two-way branch only): where the teacher wins the distillation argmin its depth is NOT detached --
                               g_disp_teacher also receives w_distil/N * sign(mono - multi) * (1 - consistency weight) * d mono / d disp */
  MAL_STEP_SYN_SPARSE = 16, /* with MAL_STEP_TEMPORAL and syn_region: syn_* hold the synthesised images ONLY at pixels whose
                               region byte has bit 0 set (a producer that writes its regions into otherwise untouched
                               buffers); elsewhere they are by definition the warped images mal_loss_step_warp wrote
                               (warp_*, still set), which the sweep reads there.  warp2_* may then be NULL: the pass in
                               front of the producer writes the warped images once instead of twice. */
  MAL_STEP_MAIN_TEMPORAL = 128, /* --main_temporal (manydepth/trainer.py:1164, loss_utils.py:152-155): the STUDENT's pass takes the
                               hint as well -- its own warped images go through the producer and r(syn_f, target) joins the
                               student's per-pixel min (weight: consistency x matching x (1 - augmentation), unchanged).  With
                               or without MAL_STEP_TEMPORAL; the same three calls, the exchange through the *_s_* members:
                                 mal_loss_step_warp  also writes warp_s_m1 / warp_s_p1 (student's disparity, the teacher's poses);
                                                     mal_loss_step_student_ready before the student's producer reads them
                                 mal_loss_step_fwd   syn_s_* in, g_syn_s_* out (unnormalised, as g_syn_*)
                                 mal_loss_step_bwd   g_warp_s_* in: the student's gradient sweep takes the four-way decisions
                                                     of _fwd and adds what arrives through syn */
  MAL_STEP_SYN_S_SPARSE = 256, /* MAL_STEP_SYN_SPARSE for the student's pair (syn_s_* / syn_s_region / warp_s_*) */
  MAL_STEP_NO_MOTION_MASK = 1024, /* mal_loss_multiscale_* only: --disable_motion_masking (manydepth/trainer.py:1321-1323): the student's
                                  weight leaves the consistency mask out (no matching mask is formed either: consistency_mask_out
                                  is not written) */
  MAL_STEP_NO_AUG = 2048,      /* mal_loss_multiscale_* only: --no_matching_augmentation (:1324-1326): ... and (1 - augmentation_mask) */
  MAL_STEP_ENSEMBLE = 4096,    /* mal_loss_multiscale_* only: --ensemble on the non-distillation route (manydepth/trainer.py:1346-1351):
                                  + mean(|(depth_teacher.detach() + depth_student)/2 - depth_student| * mask) per scale for the student;
                                  losses[44 + s] */
  MAL_STEP_NO_SSIM = 512       /* mal_loss_multiscale_* only: --no_ssim, r = mean_c |target - pred| (manydepth/trainer.py:1217-1218 --
                                  upstream reads the flag on this, the non-distillation, route only); not with MAL_STEP_TEMPORAL */
};
typedef struct mal_step_args {
  int B, H, W;
  float min_depth, max_depth;
  int flags;
  float w_main, w_distil;
  const float *color0, *color_m1, *color_p1;      /* (B,3,H,W) */
  const float *K, *inv_K;                         /* (B,16) */
  const float *disp_teacher, *disp_student;       /* (B,1,H,W), full resolution */
  const float *axisangle_m1, *translation_m1, *axisangle_p1, *translation_p1; /* (B,3) */
  const float *consistency_mask;                  /* (B,H,W), before the matching mask */
  const float *augmentation_keep;                 /* (B): 1 - augmentation_mask (or the mask, MAL_STEP_AUG_MASK) */
  const float *lowest_cost;                       /* (B,H,W) */
  const float *noise;                             /* (B,1,H,W) N(0,1), nullable */
  float *losses;                                  /* 16 */
  float *loss_total;                              /* nullable: receives losses[8], the differentiable total */
  float *mono_reproj, *ens_reproj, *multi_reproj; /* (B,1,H,W) nullable outputs */
  float *consistency_mask_out;                    /* (B,H,W) nullable: mask * matching mask */
  const float *g_total;                           /* backward: device scalar d(final)/d(total), nullable = 1 */
  float *g_disp_teacher, *g_disp_student;         /* backward outputs, nullable */
  float *g_axisangle_m1, *g_translation_m1, *g_axisangle_p1, *g_translation_p1;
  void *ws; size_t ws_bytes; void *stream;
  /* Parity instrumentation (tests; NULL in production): the per-pixel DECISIONS the teacher / student pass took,
   * MAL_DEC_PLANES uint32 planes of B*H*W each -- what a CPU checker has to be told to reproduce the discontinuous
   * choices of loss_utils.py:103-113,178-199,237-254 instead of re-deciding near-ties.  Requesting them selects
   * instrumented instantiations of the same kernels (identical arithmetic); H, W < 4096.                       */
  uint32_t *dec_teacher, *dec_student;
  /* MAL_STEP_NOISE_PHILOX: key; step number = *noise_counter when noise_counter != NULL (a device word the step's
   * last kernel advances by one, so a replayed HIP graph draws fresh noise every replay), else noise_step;
   * noise_out: (B,1,H,W) nullable, receives the N(0,1) values that were used (tests). */
  uint64_t noise_seed, noise_step; uint64_t* noise_counter; float* noise_out;
  /* MAL_STEP_TEMPORAL, all (B,3,H,W) planar */
  float *warp_m1, *warp_p1;               /* out of mal_loss_step_warp: outputs[("color", f, 0)] of the teacher */
  const float *syn_m1, *syn_p1;           /* in of _fwd: outputs[("syn", f, 0)] */
  float *g_syn_m1, *g_syn_p1;             /* out of _fwd */
  const float *g_warp_m1, *g_warp_p1;     /* in of _bwd */
  int warp_sample_stride;                 /* floats between two samples of warp_m1 / warp_p1; 0 = 3*H*W.  6*H*W with
                                             warp_p1 = warp_m1 + 3*H*W lays the two images of a sample side by side --
                                             the (2,3,H,W) pair the instance segmenter is fed (dyn_utils.py:139-140) */
  float *warp2_m1, *warp2_p1;             /* nullable, contiguous (B,3,H,W): a second copy of the warped images, written by
                                             mal_loss_step_warp -- the buffers the producer makes syn_f in (it then only
                                             overwrites the pixels its instances touch, mal_dyn_item.prefilled)           */
  const uint8_t *syn_region;              /* nullable, (B,H,W) bytes, bit 0 = the producer touched this pixel (elsewhere
                                             syn_f == warp_f): mal_loss_step_fwd then evaluates the synthesised candidates
                                             only where their 3x3 window can differ -- an exact tie goes to the warped
                                             candidate anyway, torch.min takes the first minimum                          */
  float *g_syn_region_m1, *g_syn_region_p1; /* nullable (with syn_region), (B,3,H,W): g_syn_* once more, written only at
                                             touched pixels -- what a producer's in-place backward gathers from
                                             (mal_dyn_item.region_only) */
  /* --learn_ens (loss_utils.py:240-241, trainer.py:596-600), nullable: the learnt ensemble head's disparity (B,1,H,W).
   * The ensemble pass then warps with it instead of (disp_teacher + disp_student) / 2, its depth is the distillation
   * target where the ensemble wins the three-way min, and g_ens_disp (backward output, nullable) receives
   * d total / d ens_disp.  Not with MAL_STEP_NO_ENS. */
  const float *ens_disp; float *g_ens_disp;
  /* MAL_STEP_MAIN_TEMPORAL: the members above once more for the student's pass (same shapes; warp_s_* use warp_sample_stride) */
  float *warp_s_m1, *warp_s_p1;
  const float *syn_s_m1, *syn_s_p1;
  float *g_syn_s_m1, *g_syn_s_p1;
  const float *g_warp_s_m1, *g_warp_s_p1;
  const uint8_t *syn_s_region;
  float *g_syn_s_region_m1, *g_syn_s_region_p1;
} mal_step_args;
int mal_loss_step_warp(const mal_step_args* args);
/* A MAL_STEP_TEMPORAL step whose producer failed between mal_loss_step_warp and mal_loss_step_fwd: joins the ensemble
 * pass that _warp forked onto the library's side stream back into args->stream (no-op when nothing is pending), so the
 * next step cannot overwrite buffers that pass still reads and a stream capture does not end with unjoined work. */
int mal_loss_step_abort(const mal_step_args* args);
/* Option "tail_overlap" (mal_set_option; --temporal steps): call between the step's forward and the producer's backward.
 * *stream receives the stream the producer's backward is to be enqueued on: the library's side stream -- made to wait for
 * the fused sweep, the producer of d loss / d syn -- when the option applies, args->stream otherwise.  mal_loss_step_bwd
 * then runs the teacher's gradient sweep behind it on that stream (beside the epilogue and the reduction of the forward, which
 * args->stream may still be running) and joins it back in front of the assembly.  _cancel: the caller enqueued nothing there
 * after all (its producer is not one whose backward can be redirected); the side stream is joined back at once. */
int mal_loss_step_tail_begin(const mal_step_args* args, void** stream);
int mal_loss_step_tail_cancel(const mal_step_args* args);
/* MAL_STEP_MAIN_TEMPORAL: call after mal_loss_step_warp (and after the teacher's producer, with MAL_STEP_TEMPORAL) and BEFORE the
 * student's producer reads warp_s_*: with both hints the student's forward pass runs on the library's side stream beside the
 * teacher's producer chain, and args->stream waits for it here (a no-op when the pass ran on args->stream itself). */
int mal_loss_step_student_ready(const mal_step_args* args);
/* Measurement hook (bench.py's `roofline` block): enqueues `launches` (1..4096) back-to-back launches of the teacher's
 * pass -- the fused warp + SSIM + L1 + min + automask forward+backward sweep, manydepth/loss_utils.py:57-113 with
 * trainer.py:1078-1125 -- exactly as mal_loss_step_fwd enqueues it for a step WITHOUT MAL_STEP_TEMPORAL (same parameter
 * block, instantiation and task decomposition) and nothing else.  `args->ws` must hold a finished mal_loss_step_fwd of
 * the same arguments; the launches rewrite what its teacher pass left with the same values.  Captured into a graph and
 * replayed, two events outside the graph time the kernel alone. */
int mal_loss_step_teacher_replay(const mal_step_args* args, int launches);
/* the same noise map on its own (tests; bit-identical to what the step draws for that seed / step) */
int mal_tiebreak_noise(uint64_t seed, uint64_t step, int B, int H, int W, float* out, void* stream);
enum {
  MAL_DEC_WIN = 0,     /* bits 0-1: winning candidate of the per-pixel min (0/1 = warped frame -1/+1, 2/3 = syn -1/+1);
                          bit 2: automask bit `min_c r_c <= identity + 1e-5*noise` (1 when the pass has no automask) */
  MAL_DEC_DISTIL = 1,  /* student: index of the distillation argmin (0 teacher, 1 ensemble, 2 student) */
  MAL_DEC_SMOOTH_X = 2,/* 1 + sign(d[y,x] - d[y,x+1]) of the smoothness term (layers.py:210-223), x < W-1 */
  MAL_DEC_SMOOTH_Y = 3,/* 1 + sign(d[y,x] - d[y+1,x]), y < H-1 */
  MAL_DEC_TAP0 = 4,    /* frame -1: x0 | y0 << 12 (floor of the clipped sampling position) | bit 24: x on/over the
                          border (grid gradient cut), bit 25: same for y */
  MAL_DEC_TAP1 = 5,    /* frame +1 */
  MAL_DEC_L1 = 6,      /* 1 + sign(pred - target) of the winning candidate's L1 term, channels r | g << 2 | b << 4 */
  MAL_DEC_PLANES = 7
};
size_t mal_step_workspace_bytes(int B, int H, int W);
int mal_loss_step_fwd(const mal_step_args* args);
int mal_loss_step_bwd(const mal_step_args* args);

/* ---- the loss half of process_batch WITHOUT --distil and with sclm > 0 (BASELINE.json configs[1] runs the four-scale
 * decoder): manydepth/trainer.py:573-612 calling compute_losses (:1248-1475) for the teacher and for the student over
 * scales 0..sclm, after generate_images_pred (:1078-1170) upsampled each scale's disparity to full resolution
 * (bilinear, align_corners=False) and warped the full-resolution sources with it.  Per scale s:
 *   teacher  reproj_s = sum(min_f r * automask) / (sum automask + 1e-7), identity term from the raw sources + 1e-5 noise_s
 *   student  mask m = consistency_mask (x matching mask, :592-593, when lowest_cost is given) x (1 - augmentation_mask);
 *            reproj_s over m;  consistency_s = mean |depth_student_s - depth_teacher_s| (1 - m)
 *   both     + 1e-3 * smooth(disp_s / mean disp_s, color_s) / 2**s        (disp_s, color_s at the scale's own size)
 * total = (sum_s loss_teacher_s + sum_s loss_student_s) / (sclm + 1).  One host call forward (first sweep, upsampling,
 * 2 marching launches + 2 smoothness sweeps per scale, one reduction), one backward (adjoint upsampling, gathered in a
 * fixed order: no atomics).  --v1_multiscale, --ensemble are not covered (MAL_EINVAL is not how they fail: the Python
 * mirror routes them through the operator-level API); --no_ssim: MAL_STEP_NO_SSIM; the temporal hint: MAL_STEP_TEMPORAL. */
enum { MAL_MS_MAX_SCALES = 4 };
typedef struct mal_ms_args {
  int B, H, W, sclm;                              /* scale s is (H >> s, W >> s); H, W divisible by 2**sclm */
  float min_depth, max_depth;
  int flags;                                      /* MAL_STEP_AUG_MASK, _NOISE_PHILOX, _TEMPORAL, _NO_SSIM, _NO_MOTION_MASK, _NO_AUG;
                                                     --disable_automasking: pass no noise (upstream compares against the identity
                                                     term even then, manydepth/trainer.py:1296-1311: only the noise goes) */
  const float *color0, *color_m1, *color_p1;      /* (B,3,H,W) */
  const float *color0_s[MAL_MS_MAX_SCALES];       /* inputs[("color",0,s)]: (B,3,H>>s,W>>s); [0] NULL = color0 */
  const float *K, *inv_K;                         /* (B,16) */
  const float *disp_teacher[MAL_MS_MAX_SCALES], *disp_student[MAL_MS_MAX_SCALES]; /* (B,1,H>>s,W>>s) */
  const float *axisangle_m1, *translation_m1, *axisangle_p1, *translation_p1;     /* (B,3) */
  const float *consistency_mask;                  /* (B,H,W) */
  const float *augmentation_keep;                 /* (B) */
  const float *lowest_cost;                       /* (B,H,W) nullable: no matching mask */
  const float *noise[MAL_MS_MAX_SCALES];          /* (B,1,H,W) each, all NULL with MAL_STEP_NOISE_PHILOX */
  uint64_t noise_seed, noise_step; uint64_t* noise_counter;
  float *losses;                                  /* 48: [net][scale][reproj, consistency, smooth, loss] (net 0 teacher),
                                                     [32] teacher total, [33] student total, [34] their sum,
                                                     [36+s] reproj student+teacher, [40+s] loss student+teacher,
                                                     [44+s] the student's ensemble term (MAL_STEP_ENSEMBLE, else 0) */
  float *loss_total;                              /* nullable: receives losses[34] */
  float *consistency_mask_out;                    /* (B,H,W) nullable */
  const float *g_total;                           /* backward: device scalar, nullable = 1 */
  float *g_disp_teacher[MAL_MS_MAX_SCALES], *g_disp_student[MAL_MS_MAX_SCALES];   /* backward outputs, nullable */
  float *g_axisangle_m1, *g_translation_m1, *g_axisangle_p1, *g_translation_p1;
  void *ws; size_t ws_bytes; void *stream;
  /* MAL_STEP_TEMPORAL in `flags` (--temporal on this path, trainer.py:1161-1162 + 1279-1283): generate_images_pred calls the
   * producer once PER SCALE on that scale's full-resolution warp of the teacher, and compute_losses adds r(syn_f^s, target)
   * to scale s's per-pixel min when the LAST call reported instances (has_ins is overwritten per scale, :1162).  Three calls
   * around the producers, as mal_loss_step_*: mal_loss_multiscale_warp (everything that does not wait for them: first
   * sweep, upsampling, the teacher's forward pass per scale -> warp_*[s], the student's passes, smoothness),
   * [syn^s = producer(warp^s)], mal_loss_multiscale_fwd (syn_*[s] in; losses, g_syn_*[s] out, unnormalised),
   * [g_warp^s = producer^T(g_syn^s)], mal_loss_multiscale_bwd (g_warp_*[s] in).  Members per scale s <= sclm, all (B,3,H,W)
   * planar, with the meaning of their mal_step_args namesakes; bit s of syn_sparse = MAL_STEP_SYN_SPARSE for scale s. */
  float *warp_m1[MAL_MS_MAX_SCALES], *warp_p1[MAL_MS_MAX_SCALES];
  const float *syn_m1[MAL_MS_MAX_SCALES], *syn_p1[MAL_MS_MAX_SCALES];
  float *g_syn_m1[MAL_MS_MAX_SCALES], *g_syn_p1[MAL_MS_MAX_SCALES];
  const float *g_warp_m1[MAL_MS_MAX_SCALES], *g_warp_p1[MAL_MS_MAX_SCALES];
  const uint8_t *syn_region[MAL_MS_MAX_SCALES];
  float *g_syn_region_m1[MAL_MS_MAX_SCALES], *g_syn_region_p1[MAL_MS_MAX_SCALES];
  int warp_sample_stride, syn_sparse;
  /* parity instrumentation (tests), per scale: the per-pixel decisions of that scale's gradient passes, as mal_step_args.
   * dec_teacher / dec_student (int32 (MAL_DEC_PLANES,B,H,W) each; MAL_DEC_WIN, _TAP0/1, _L1 are written -- the smoothness signs of
   * this path are those of the raw disparity differences at the scale's own size, which a checker forms from the inputs
   * exactly).  NULL (the default): the specialised, uninstrumented kernels run; results are bit-identical either way. */
  uint32_t *dec_teacher[MAL_MS_MAX_SCALES], *dec_student[MAL_MS_MAX_SCALES];
} mal_ms_args;
size_t mal_ms_workspace_bytes(int B, int H, int W, int sclm);
int mal_loss_multiscale_warp(const mal_ms_args* args); /* MAL_STEP_TEMPORAL only */
/* a MAL_STEP_TEMPORAL step whose producer failed between _warp and _fwd: joins the library's side stream (the students' passes
 * run there beside the producers) back into args->stream; a no-op when nothing is pending */
int mal_loss_multiscale_abort(const mal_ms_args* args);
int mal_loss_multiscale_fwd(const mal_ms_args* args);
int mal_loss_multiscale_bwd(const mal_ms_args* args);
/* ---- DualRefine's loss loops over the deq iterations of scale 0 in one call per direction ------------------------------
 * dualrefine/trainer.py:395-451 (generate_images_pred, Project3D convention B + align_corners=False) and :530-633
 * (compute_losses, the per-deq-iteration branch).  Per iteration it = 0..n_iters-1:
 *   reproj_it = sum(min_f r(warp_f) * mask) / (sum mask + 1e-7),  mask = automask(identity term + 1e-5 noise_it)
 *               [x consistency_mask for it > 0, :593-597]
 *   consistency_it = mean |depth_it - depth_0| (1 - mask)  for it > 0 (no gradient to depth_0, :606-618)
 *   loss_it = reproj_it + consistency_it + smooth_weight * smooth(disp_it / mean disp_it, color)
 * and upstream's running loss is added to the total once per iteration (:624-631): total = sum_it (n_iters - it) loss_it.
 * T_m1[it] / T_p1[it] are the 4x4 poses the trainer selects for that iteration (:420-435: the refined pose for frame -1
 * and it > 0, ...); which of them are detached is the caller's business: a NULL g_T_* output is a detached pose.
 * losses: [4*it + {0 reproj, 1 consistency, 2 smooth, 3 running loss after it}], [4*MAL_DR_MAX_ITERS] total,
 * [4*MAL_DR_MAX_ITERS + 1] the final running loss (what every "loss/0_it" entry reads upstream).
 * One call = the iterations of ONE scale (`scale`, below); --avg_reprojection / --no_ssim: MAL_DR_AVG / MAL_DR_NO_SSIM.
 * MAL_DR_POSE_UPDATE (scale 0 only): the same call also evaluates the pose-update losses (:457-480 pose_update_generate_images_pred,
 * :699-767 compute_pose_update_losses) in ONE more marching pass whose two candidates are warped with two different
 * disparities: frame -1 with pu_disp_m1 (the last iteration's disparity -- or iteration 0's, detached, with --Tstar_D0_pair)
 * under the refined pose pu_T_m1 = ("cam_T_cam", 0, -1, 1); frame +1 as ("color", 1, 0, 0) was: pu_disp_p1 = iteration 0's
 * disparity under pu_T_p1 = ("cam_T_cam", 0, 1).  min (or mean) over the two, automask against the identity term + 1e-5 pu_noise,
 * no consistency mask, no smoothness: losses[4*MAL_DR_MAX_ITERS + 2] = sum(rp mask) / (sum mask + 1e-7).  It is NOT part of
 * losses[4*MAL_DR_MAX_ITERS] / loss_total (upstream adds it after the division by len(scales), :337-343): the caller adds it,
 * and hands its cotangent in g_pu_total. */
enum { MAL_DR_MAX_ITERS = 4 };
enum { MAL_DR_AVG = 8,      /* --avg_reprojection (dualrefine/trainer.py:569-583): the MEAN over the two frames of r and of the
                               identity term instead of their min; both frames take half of every gradient */
       MAL_DR_NO_SSIM = 16, /* --no_ssim (:493-494): r = mean_c |target - pred| */
       MAL_DR_NO_AUTOMASK = 1, MAL_DR_NO_MOTION_MASK = 2,
       MAL_DR_NOISE_PHILOX = 4, /* the tie-break noise of every iteration is drawn in the step's first launch (Philox4x32-10
                                  keyed by noise_seed, step number = step * MAL_DR_MAX_ITERS + it: mal_tiebreak_noise with
                                  that step number reproduces iteration it's map); every noise[it] must be NULL.  The
                                  pose-update pass's map: key noise_seed ^ MAL_DR_POSE_NOISE_KEY, step number step * MAL_DR_MAX_ITERS */
       MAL_DR_POSE_UPDATE = 32 /* the pose-update losses ride on this call (see above) */ };
#define MAL_DR_POSE_NOISE_KEY 0x706f73655f757064ull /* "pose_upd" */
typedef struct mal_dr_args {
  int B, H, W, n_iters;
  float min_depth, max_depth, smooth_weight;
  int flags;
  const float *color0, *color_m1, *color_p1;      /* (B,3,H,W) */
  const float *K, *inv_K;                         /* (B,16) */
  const float *disp[MAL_DR_MAX_ITERS];            /* (B,1,H,W) */
  const float *T_m1[MAL_DR_MAX_ITERS], *T_p1[MAL_DR_MAX_ITERS]; /* (B,16) */
  const float *consistency_mask;                  /* (B,H,W); required when n_iters > 1 unless MAL_DR_NO_MOTION_MASK */
  const float *noise[MAL_DR_MAX_ITERS];           /* (B,1,H,W) N(0,1) per iteration, nullable */
  /* MAL_DR_NOISE_PHILOX: key; step = *noise_counter when noise_counter != NULL (a device word the step's last kernel
   * advances by one, so a replayed HIP graph draws fresh noise every replay), else noise_step */
  uint64_t noise_seed, noise_step; uint64_t* noise_counter;
  float *losses;                                  /* 4*MAL_DR_MAX_ITERS + 4 */
  float *loss_total;                              /* nullable: receives the total */
  const float *g_total;                           /* backward: device scalar, nullable = 1 */
  float *g_disp[MAL_DR_MAX_ITERS];                /* backward outputs, nullable each */
  float *g_T_m1[MAL_DR_MAX_ITERS], *g_T_p1[MAL_DR_MAX_ITERS];
  void *ws; size_t ws_bytes; void *stream;
  /* One call covers the iterations of ONE scale of opt.scales (upstream's default list is [0,1,2,3]: scale 0 and 2 with
   * n_losses+1 iterations, scale 1 skipped, scale 3 iteration 0 only, dualrefine/trainer.py:403-407,536-547; the caller sums
   * the calls' totals and divides by len(scales), :694).  scale > 0: disp[it] is the iteration's disparity UPSAMPLED to
   * (H,W) (trainer.py:411-412; mal_upsample_bilinear) -- what is warped and what the consistency term reads --, while the
   * smoothness term (:620-624) is taken on disp_lo[it], (B,1,H>>scale,W>>scale), against color0_s = inputs[("color",0,scale)],
   * (B,3,H>>scale,W>>scale); pass smooth_weight = disparity_smoothness / 2**scale.  Backward then writes g_disp[it] =
   * d total / d (upsampled disparity) WITHOUT the smoothness term and g_disp_lo[it] = the smoothness term's gradient at the
   * scale's own size (the caller adds the adjoint upsampling of the first: mal_upsample_bilinear_adjoint). */
  int scale;
  const float *color0_s;
  const float *disp_lo[MAL_DR_MAX_ITERS];
  float *g_disp_lo[MAL_DR_MAX_ITERS];
  /* nullable: the workspace of an earlier mal_dr_loss_fwd call of the SAME step (same images, B, H, W, flags) -- the texel
   * copies of the three images and the identity term are taken from it instead of being formed again (a step over several
   * scales packs once).  Honoured when this call's smoothness term does not ride on that sweep (scale > 0 or n_iters > 2). */
  const void *texels_from;
  /* parity instrumentation (tests), per iteration: the per-pixel decisions of that iteration's pass, as mal_step_args.dec_teacher
   * (int32 (MAL_DEC_PLANES,B,H,W); MAL_DEC_WIN, _TAP0/1, _L1).  NULL: the uninstrumented kernels; bit-identical results. */
  uint32_t *dec[MAL_DR_MAX_ITERS];
  /* MAL_DR_POSE_UPDATE */
  const float *pu_disp_m1, *pu_disp_p1;           /* (B,1,H,W) */
  const float *pu_T_m1, *pu_T_p1;                 /* (B,16) */
  const float *pu_noise;                          /* (B,1,H,W) N(0,1), nullable; must be NULL with MAL_DR_NOISE_PHILOX */
  const float *g_pu_total;                        /* backward: device scalar, nullable = 1 */
  float *g_pu_disp_m1, *g_pu_disp_p1;             /* backward outputs, nullable each (a NULL output is a detached operand) */
  float *g_pu_T_m1, *g_pu_T_p1;
  uint32_t *pu_dec;                               /* parity instrumentation, as dec[] */
  float *pu_loss_total;                           /* nullable: receives the pose-update loss (= losses[4*MAL_DR_MAX_ITERS + 2]) */
  /* An operand of the pose-update pass that IS an iteration's operand (upstream's default pairing: pu_disp_m1 = disp[n_iters-1],
   * pu_disp_p1 = disp[0], pu_T_m1 = T_m1[n_iters-1], pu_T_p1 = T_p1[0]) can have its pose-update gradient ADDED into that
   * iteration's output by the assembly launch instead of leaving as an output of its own that the caller then adds:
   * k + 1 = "add into iteration k's g_disp / g_T_m1 / g_T_p1" (the matching g_pu_* must then be NULL), 0 = separate output. */
  int pu_disp_m1_into, pu_disp_p1_into, pu_T_m1_into, pu_T_p1_into;
} mal_dr_args;
size_t mal_dr_workspace_bytes(int B, int H, int W, int n_iters);
int mal_dr_loss_fwd(const mal_dr_args* args);
int mal_dr_loss_bwd(const mal_dr_args* args);

/* F.interpolate(x, [H, W], mode="bilinear", align_corners=False) of a (B,1,h,w) map and its adjoint (a gather in a
 * fixed order), as the step uses them (tests) */
int mal_upsample_bilinear(const float* x, int B, int h, int w, int H, int W, float* out, void* stream);
int mal_upsample_bilinear_adjoint(const float* g_out, int B, int h, int w, int H, int W, float* g_x, void* stream);

/* ---- N2: the temporal-hint producer's per-sample arithmetic, manydepth/dyn_utils.py:6-119 --------
 * (fill_dynamic_obj + generate_dynamic_instance), given the matched instance masks of the two warped frames
 * (num,H,W as bytes, non-zero = set; Mask2Former and the matcher stay outside).  Per instance the displacement
 * is half the larger-magnitude difference of the bounding-box edges between the frames (rows: low/top, columns:
 * right/left; index 0 is invisible to the extents, as upstream; half-to-even; replace=1 zeroes |d| < 3); "last"
 * patches move by +d, "next" by -d, overlapping copies add, pixels vacated by an instance show the other frame.
 * Outputs: ori_last / ori_next (C,H,W) = the reference's return values; delta [num][2] (row, column) and
 * flags [H*W] are what mal_dyn_instance_bwd needs.  C <= 4, num <= MAL_MAX_INSTANCES.                     */
size_t mal_dyn_workspace_bytes(int num);
int mal_dyn_instance_fwd(const uint8_t* mask_last, const uint8_t* mask_next, int num, const float* img_last,
                         const float* img_next, int C, int H, int W, int replace, float* ori_last, float* ori_next,
                         int32_t* delta, uint8_t* flags, void* ws, size_t ws_bytes, void* stream);
/* cotangents of (ori_last, ori_next) -> gradients w.r.t. (img_last, img_next), either nullable */
int mal_dyn_instance_bwd(const uint8_t* mask_last, const uint8_t* mask_next, int num, const int32_t* delta,
                         const uint8_t* flags, const float* g_ori_last, const float* g_ori_next, int C, int H, int W,
                         float* g_img_last, float* g_img_next, void* stream);

/* The same for several samples of a batch in one call (three launches forward, one backward for up to 16 samples; more
 * are chunked): one item per sample with matched instances; the fields are the arguments of the two functions above. */
typedef struct mal_dyn_item {
  const uint8_t* mask_last; const uint8_t* mask_next; int num;
  const float* img_last; const float* img_next;  /* forward in */
  float* ori_last; float* ori_next;              /* forward out */
  int32_t* delta; uint8_t* flags;                /* forward out, backward in */
  void* ws; size_t ws_bytes;                     /* forward scratch: mal_dyn_workspace_bytes(num) */
  const float* g_ori_last; const float* g_ori_next; float* g_img_last; float* g_img_next;  /* backward */
  /* optional: instance i is row idx_last[i] / idx_next[i] (device int64) of mask_last / mask_next -- the matcher's
   * selection (dyn_utils.py:147-150) applied inside the kernels instead of by a gather per sample and frame; NULL = row i */
  const int64_t* idx_last; const int64_t* idx_next;
  /* forward: ori_last / ori_next already hold img_last / img_next (the step's warp pass wrote both copies): only the pixels
   * of the instances' region are written */
  int prefilled;
  /* backward IN PLACE (g_img_* == g_ori_*: the cotangent buffers become the gradients): only region pixels change; they
   * are formed in g_tmp_* (C,H,W scratch, contents irrelevant) by one launch and moved by a second */
  float* g_tmp_last; float* g_tmp_next;
  /* backward, third form: g_ori_* are SNAPSHOTS valid at region pixels only (mal_step_args.g_syn_region_*), g_img_* the
   * buffers that hold the cotangent everywhere: one launch writes the region pixels of g_img_* (no scratch, no move) */
  int region_only;
  /* rows of mask_last / mask_next (0 = unknown): with them a selection idx_*[i] outside [0, rows) is clamped into the
   * tensor instead of being dereferenced (the torch indexing it replaces raised IndexError) */
  int n_last, n_next;
} mal_dyn_item;
int mal_dyn_batch_fwd(const mal_dyn_item* items, int n_items, int C, int H, int W, int replace, void* stream);
int mal_dyn_batch_bwd(const mal_dyn_item* items, int n_items, int C, int H, int W, void* stream);

/* ---- N3: ManyDepth's cost volume as MAL's student encoder builds it (forward only; upstream runs it under
 * no_grad): manydepth/networks/resnet_encoder.py:152-233 match_features + :296-312 of the encoder's forward.
 * current_feats (B,C,h,w) and lookup_feats (B,F,C,h,w) planar as the encoder produces them, C = 64 (with option
 * "costvol_impl" 0, the first formulation, both CHANNEL-LAST instead: mal_costvol_channel_last() says which); poses (B,F,16) relative poses (an
 * all-zero pose = missing frame, skipped); K / inv_K (B,16) at the matching resolution; depth_bins (D <= 256).
 * cost_volume (B,D,h,w) = match_features' first return value (mean |warped - current| x border masks, averaged
 * over the frames that hit, missing bins set to the pixel's maximum when set_missing_to_max); nullable outputs:
 * missing_mask (B,D,h,w) = its second return value; masked_cost_volume = cost_volume x confidence (:311);
 * lowest_cost (B,h,w) = 1 / depth of the first minimum, zeros read as 100 (:303-307); confidence_mask (B,h,w). */
int mal_costvol_channel_last(void);
int mal_cost_volume(const float* current_feats, const float* lookup_feats, const float* poses, const float* K,
                    const float* inv_K, const float* depth_bins, int B, int F, int C, int D, int h, int w, float eps,
                    int set_missing_to_max, float* cost_volume, float* missing_mask, float* masked_cost_volume,
                    float* lowest_cost, float* confidence_mask, void* stream);

/* ---- N4 (forward): DualRefine's epipolar correlation lookup, dualrefine/networks/depth_pose.py:433-435.
 * mal_epipolar_coords = Reprojections.depth2epipolarcoords (dualrefine/networks/utils/utils.py:180-217, --gap_factor
 * depth): depth (B,1,h,w), poses (B,16) relative poses, K (B,16) at that resolution, r = --corr_radius, L = --num_levels
 * (<= 4), softplus_delta = softplus of the module's learnable delta, ratio = --gap_factor_depth_ratio ->
 * coords (B,2,L,2r+1,h,w) pixel coordinates in the other view, max_dx (B,1,h,w), depths (B,1,L*(2r+1),h,w).
 * mal_coord_sample_l1 = CoordSampler.__call__ (dualrefine/networks/corr.py:25-50): fmap1 (B,C,h,w), f2_pyramid[l]
 * (B,C,h>>l,w>>l) (avg_pool2d(2,2) levels, corr.py:19-23), coords as above with d1 hypotheses per level ->
 * out (B, L*heads*d1, h, w): per level and channel group (heads) the mean of |fmap1 - bilinear zero-padded sample|. */
int mal_epipolar_coords(const float* depth, const float* poses, const float* K, int B, int h, int w, int r, int L,
                        float softplus_delta, float ratio, float* coords, float* max_dx, float* depths, void* stream);
/* _iproj / pose / _proj for given depth hypotheses (Reprojections.depthbins2coords, utils.py:251-253): depths (B,1,D,h,w)
 * -> coords (B,2,1,D,h,w). */
int mal_epipolar_coords_of_depths(const float* depths, const float* poses, const float* K, int B, int D, int h, int w,
                                 float* coords, void* stream);
int mal_coord_sample_l1(const float* fmap1, const float* const* f2_pyramid, const float* coords, int B, int C, int h, int w,
                        int L, int d1, int heads, float* out, void* stream);
/* VJPs of the two lookup functions (the DEQ solver differentiates through them in training,
 * dualrefine/networks/depth_pose.py:426-455).  mal_epipolar_coords_bwd: cotangents of (coords, max_dx, depths) --
 * the last two nullable -- -> g_depth (B,1,h,w), g_poses (B,16; row 3 zero), g_dd (1) = d / d softplus(delta); fixed
 * summation order.  mal_coord_sample_l1_bwd: cotangent of the lookup's output -> g_fmap1 (B,C,h,w), g_f2_pyramid[l]
 * (level shapes), g_coords; g_fmap1 / g_f2_pyramid[l] are ACCUMULATED into (zero them first); any of the three may be
 * NULL.  With ws and planes that fit the LDS the sums are exact integer (fixed-point) accumulations per (sample, channel)
 * plane -- order-independent --; otherwise float atomics, as ATen's grid_sampler backward. */
size_t mal_epipolar_coords_bwd_workspace_bytes(int B, int h, int w);
int mal_epipolar_coords_bwd(const float* depth, const float* poses, const float* K, const float* g_coords,
                            const float* g_max_dx, const float* g_depths, int B, int h, int w, int r, int L,
                            float softplus_delta, float ratio, float* g_depth, float* g_poses, float* g_dd,
                            void* ws, size_t ws_bytes, void* stream);
int mal_coord_sample_l1_bwd(const float* fmap1, const float* const* f2_pyramid, const float* coords, const float* g_out,
                            int B, int C, int h, int w, int L, int d1, int heads, float* g_fmap1,
                            float* const* g_f2_pyramid, float* g_coords, void* ws, size_t ws_bytes, void* stream);
size_t mal_coord_sample_l1_bwd_workspace_bytes(int B); /* ws nullable: without it the feature cotangents use global atomics */
/* The pose refinement step of the same loop (depth_pose.py:450-455), forward:
 * mal_epipolar_gradcoords = Reprojections.depth2gradcoords (utils.py:219-236): c_p (B,2,1,5,h,w) = the projection and its
 * +-1 px neighbours in x and y, P2 (B,4,h*w) = the transformed points.
 * mal_direct_align_normal_eq = PoseUpdate.direct_align up to the solve (utils.py:303-355; robust != 0: with the
 * --robust_pose_loss weights of :344-355, scaled_barron(0, 0.1) of the squared residual x the in-image test):
 * src_feat / tgt_feat (B,C,h,w), src_w / tgt_w (B,1,h,w) (tgt_w is sampled at the projection as :388-390), weight (B,1,h,w)
 * nullable, K (B,16), p2 = c_p, P2 as above -> H (B,6,6) = sum_px w J^T J and b (B,6) = sum_px -res w J of the
 * feature-metric Gauss-Newton step; ws: mal_direct_align_workspace_bytes(B,h,w). */
int mal_epipolar_gradcoords(const float* depth, const float* poses, const float* K, int B, int h, int w, float* c_p, float* P2,
                            void* stream);
size_t mal_direct_align_workspace_bytes(int B, int h, int w);
/* The rest of direct_align (utils.py:357-368): 6x6 Cholesky solve with upstream's fall-backs (general solve, then the
 * unchanged pose), se3_exp (dualrefine/layers.py:29-55), new_poses (B,16) = exp(update) @ poses; update (B,6). */
int mal_direct_align_update(const float* H, const float* b, const float* poses, int B, float* new_poses, float* update,
                            void* stream);
int mal_direct_align_normal_eq(const float* src_feat, const float* tgt_feat, const float* src_w, const float* tgt_w,
                               const float* weight, const float* K, const float* p2, const float* P2, int B, int C, int h, int w,
                               int robust, float* H, float* b, void* ws, size_t ws_bytes, void* stream);
/* VJPs of the three (the last unrolled solver step differentiates through the pose update, depth_pose.py:450-455).
 * mal_epipolar_gradcoords_bwd: cotangents of c_p and P2 (nullable) -> g_depth (B,1,h,w), g_poses (B,16); fixed order.
 * mal_direct_align_normal_eq_bwd: cotangents of H (B,36) and b (B,6) -> every output nullable: g_src_feat (B,C,h,w),
 * g_tgt_feat (B,C,h,w) and g_tgt_w (B,1,h,w) (ZERO on entry: accumulated with float atomics, as ATen's grid_sampler
 * backward), g_src_w, g_weight (B,1,h,w), g_p2 (B,2,1,5,h,w), g_P2 (B,4,h*w; row 3 zero).  K receives no gradient.
 * mal_direct_align_update_bwd: cotangents of new_poses (B,16) and update (B,6; nullable) -> g_H (B,36; symmetrised on the
 * Cholesky path as torch.linalg.cholesky's backward), g_b (B,6), g_poses (B,16). */
size_t mal_epipolar_gradcoords_bwd_workspace_bytes(int B, int h, int w);
int mal_epipolar_gradcoords_bwd(const float* depth, const float* poses, const float* K, const float* g_c_p, const float* g_P2,
                                int B, int h, int w, float* g_depth, float* g_poses, void* ws, size_t ws_bytes, void* stream);
int mal_direct_align_normal_eq_bwd(const float* src_feat, const float* tgt_feat, const float* src_w, const float* tgt_w,
                                   const float* weight, const float* K, const float* p2, const float* P2, const float* g_H,
                                   const float* g_b, int B, int C, int h, int w, int robust, float* g_src_feat,
                                   float* g_tgt_feat, float* g_src_w, float* g_tgt_w, float* g_weight, float* g_p2, float* g_P2,
                                   void* ws, size_t ws_bytes, void* stream);
size_t mal_direct_align_bwd_workspace_bytes(int B, int h, int w); /* ws nullable: without it g_tgt_feat is scattered with global atomics */
int mal_direct_align_update_bwd(const float* H, const float* b, const float* poses, const float* g_new_poses,
                                const float* g_update, int B, float* g_H, float* g_b, float* g_poses, void* stream);

/* ---- library options:
 * "pass_impl"   formulation of the fused pass: 1 = register-marching (default); 0 / 2 = the LDS-tiled first
 *               versions (256 threads x 4 px, 512 threads x 2 px), kept for A/B;
 * "march_rows"  output rows per wavefront task of the marching kernels (0 = automatic, default);
 * "march_flip"  1 (default): odd row segments of the marching kernels walk bottom-up, so the two tasks sharing a
 *               segment boundary reach it together and the halo rows are served by the L2; 0 = all top-down;
 * "costvol_impl" 1 (default): cost volume with planar features, lane = pixel; 0: channel-last, lane = channel;
 * "photo_impl"  mal_photo_fwd/bwd: 1 = marching kernels, two candidates per launch (default for SSIM + min);
 *               0 = one pixel per thread (ATen's summation order; always used for MAL_F_NO_SSIM / MAL_F_AVG);
 * "epi_bwd_planes" 2 (default): the feature-map cotangents of the N4 VJPs are accumulated per (sample, channel) plane in
 *               LDS when the planes fit, the lookup's with TWO channel planes per workgroup where 2 x the planes fit (round 4:
 *               coordinates, cotangent and tap set of a (pixel, hypothesis) pair serve both); 1: one plane per workgroup
 *               (round 3); 0: global float atomics everywhere (the first formulation, kept for A/B);
 * "step_overlap" 1 (default): with MAL_STEP_TEMPORAL the ensemble pass runs on a side stream beside the producer (forked
 *               after the warp pass, joined before the student pass; events, capturable); 0: in line; 2: beside the
 *               fused sweep (slower: kept for A/B);  "syn_rows": rows per task of the fused sweep given a region map;
 * "student_overlap" 1 (default): with MAL_STEP_TEMPORAL the student's marching pass (without its consistency / distillation
 *               epilogue, which becomes a pointwise launch after the join) is forked beside the producer behind the ensemble
 *               pass; 0: in line after the fused sweep, epilogue inside the pass (round 3, kept for A/B);
 * "march_halo1" 1 (default): the gradient passes of the whole-step lists warp ONE row beyond each end of a task's segment
 *               and hand the boundary rows' missing terms over through scratch rows; 0: two rows, no hand-over (A/B);
 * "march_lean"  1 (default): the teacher / student passes of the whole-step lists run the instantiations without the code
 *               of the optional operands they never pass (march_teacher_kernel, march_student_kernel); 0: generic (A/B);
 * "syn_queue"   1: the fused sweep over the synthesised pair takes its tasks from a classified dispatch order (slower,
 *               default 0);
 * "temporal_spec" 1: with MAL_STEP_TEMPORAL the pass in front of the producer already takes the teacher's gradient and the
 *               sweep behind the producer's backward only redoes the tasks near the region map (slower unless no sample
 *               has instances; default 0);
 * "march3"      1: the teacher's gradient pass as three cooperating wavefronts per strip, rows handed over through LDS
 *               (slower; default 0);
 * "epi_probe", "fwd_waves", "debug": kernel experiments / timing probes. */
int mal_set_option(const char* name, int value);
/* Options are PROCESS-WIDE switches for same-box A/B measurements, not per-call configuration: set them before the first
 * step and leave them; a second trainer in the process sees the same values.  Setting one is serialised by a mutex;
 * launches read them without further synchronisation.  The formulations that were measured slower (pass_impl 0 / 2,
 * march3, temporal_spec, syn_queue) are compiled only into a library built with -DMAL_EXPERIMENTS (MAL_EXPERIMENTS=1
 * python -m mal_amd.build); the default library refuses those switches with MAL_EINVAL.  The one-shot arms below
 * (mal_profile_next_pass, mal_decisions_next_pass) are process-wide too (a step's backward runs on autograd's worker
 * thread, not on the arming one) and are taken with an atomic exchange: exactly one pass consumes an arm. */
int mal_build_has_experiments(void); /* 1 when the library contains the -DMAL_EXPERIMENTS formulations */
/* sizeof of the argument blocks as THIS library was compiled (0 mal_step_args, 1 mal_ms_args, 2 mal_dr_args, 3 mal_dyn_item;
 * else 0): a binding in another language checks its own layout against it once, at load (mal_amd/_lib.py does). */
size_t mal_struct_bytes(int which);

/* ---- measurement hooks (bench.py): HIP events recorded immediately before / after the main
 * kernel of the NEXT mal_pass_fused call, on its stream (one-shot; cleared by that call). */
void* mal_event_create(void);
int mal_event_destroy(void* ev);
int mal_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int mal_profile_next_pass(void* start, void* stop);
/* The task decomposition march_launch picks for a fused pass over (B,H,W) with `flags` (MAL_F_GRAD decides the halo) on
 * the current device: strips x segs wavefront tasks per sample, `rows` output rows per task, `iterations` of the row
 * loop a full task runs (rows + its halo rows).  bench.py prices the kernel's vector-ALU time with it. */
int mal_march_geometry(int B, int H, int W, int flags, int* strips, int* segs, int* rows, int* iterations);
/* Shader clock under load: two wavefronts of packed FMAs on every SIMD for `iters` rounds of 64 instructions; wave 0
 * leaves out3[0] = shader-clock ticks (s_memtime) and out3[1] = ticks of the constant 100 MHz counter (s_memrealtime)
 * around its loop (out3: 3 device words, the third is scratch).  clock_MHz = 100 * out3[0] / out3[1]. */
int mal_clock_probe(unsigned long long* out3, int iters, void* stream);
/* Parity instrumentation for the operator-level fused pass (tests): the next mal_pass_fused call whose flags are
 * GRAD|AUTOMASK|POSE_GRAD (no epilogue) or GRAD|EPILOGUE (no automask, no pose gradient) writes its per-pixel decisions
 * into `planes` (MAL_DEC_PLANES x B*H*W uint32, MAL_DEC_* above; the smoothness planes stay untouched) -- one shot. */
int mal_decisions_next_pass(uint32_t* planes);

#ifdef __cplusplus
}
#endif
#endif /* MAL_HIP_H */

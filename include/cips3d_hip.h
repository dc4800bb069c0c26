/*
 * cips3d_hip.h -- C ABI of libcips3d_hip.so: the MI355X (gfx950) implementation of the
 * CIPS-3D++ generator-forward hot path.
 *
 * Conventions (every entry point):
 *   - plain pointers + sizes, no framework types; all pointers are DEVICE pointers to
 *     contiguous fp32 (or int32 where stated) unless the name ends in `_host`;
 *   - the caller allocates every output and keeps every buffer alive until the stream
 *     has passed the call; nothing is allocated, freed or synchronised inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - returns 0 on success, a hipError_t (> 0) from the launch, or a negative CIPS3D_E_*
 *     code for argument errors (nothing is launched in that case);
 *   - stateless and re-entrant per stream.
 *
 * Reference interfaces replaced (paths relative to /root/reference/exp/):
 *   cips3d_fused_bias_act     op/fused_bias_act.cpp:11-20, op/fused_bias_act_kernel.cu:18-98
 *   cips3d_upfirdn2d          op/upfirdn2d.cpp:12-23, op/upfirdn2d_kernel.cu:49-369
 *   cips3d_linear*            models/model_v3.py:40-65,183-210 (MappingLinear/EqualLinear),
 *                             cips3d/volume_renderer.py:15-35 (LinearLayer gamma/beta heads)
 *   cips3d_camera_params      cips3d/nerf_utils.py:344-436,466-564
 *   cips3d_nerf_*             cips3d/nerf_utils.py:18-338 + cips3d/volume_renderer.py:39-303
 *   cips3d_modconv_* / torgb  models/model_v3.py:218-341,418-482
 */
#ifndef CIPS3D_HIP_H
#define CIPS3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CIPS3D_ABI_VERSION 30  /* bumped with every change of an entry point or of a struct layout below */

#define CIPS3D_E_BADARG   (-1)   /* null pointer / non-positive size */
#define CIPS3D_E_UNSUPP   (-2)   /* configuration outside what the kernels implement */

int cips3d_abi_version(void);
/* human-readable text for a return code of any entry point (static storage) */
const char* cips3d_strerror(int code);

/* ------------------------------------------------------------------ op level */

/* out[i] = act(x[i] + bias[(i / step_b) % size_b]) * scale.
 * act: 1 = linear, 3 = leaky-relu(alpha).  grad: 0 = forward; 1 = first derivative w.r.t. x
 * evaluated with `ref` (the saved forward OUTPUT) deciding the branch; 2 = second derivative (0).
 * bias / ref may be NULL (absent).  Same contract as fused.fused_bias_act(input, bias, refer,
 * act, grad, alpha, scale) with step_b = prod(dims[2:]). */
int cips3d_fused_bias_act(const float* x, const float* bias, const float* ref, float* out,
                          int64_t n, int64_t step_b, int64_t size_b,
                          int act, int grad, float alpha, float scale, void* stream);

/* Up-sample by zero insertion, pad / crop, FIR-filter (true convolution with `kernel`),
 * down-sample.  input [major, in_h, in_w, minor] -> out [major, out_h, out_w, minor],
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kernel_h) / down_y + 1 (same for w); the caller
 * sizes `out` accordingly.  kernel [kernel_h, kernel_w] lives in device memory. */
int cips3d_upfirdn2d(const float* input, const float* kernel, float* out,
                     int64_t major, int in_h, int in_w, int minor, int kernel_h, int kernel_w,
                     int up_x, int up_y, int down_x, int down_y,
                     int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* Image post-step of the demo loops (clamp [-1,1] -> [0,255], round to nearest): out[i] = u8((clamp(x)+1)*127.5).
 * Both pointers 16-byte aligned.  Used before the multi-GPU gather (4x fewer bytes over xGMI). */
int cips3d_rgb_to_uint8(const float* rgb, uint8_t* out, int64_t n, void* stream);

/* ------------------------------------------------------------------ small dense layers */

/* One dense layer on a batch of row vectors:
 *   h      = pixelnorm ? x[b] * rsqrt(mean(x[b]^2) + 1e-8) : x[b]
 *   y      = sum_i h[i] * (W[o][i] * w_scale) + bias[o] * b_scale          (bias may be NULL)
 *   y      = lrelu ? leaky_relu(y, 0.2) * act_gain : y
 *   y      = y * out_scale + out_shift
 *   out    = trunc_mean ? trunc_mean[o] + trunc_psi * (y - trunc_mean[o]) : y
 * and the result is written out_repeat (>= 1) times, copy r at out + r * out_repeat_stride (the
 * `w.unsqueeze(1).repeat(1, n_latent, 1)` of the mapping networks, models/model_v3.py:1367,1416).
 * x [B, x_stride] (first in_dim entries used), W [out_dim, in_dim], out [B, out_stride]. */
int cips3d_linear(const float* x, int64_t x_stride, const float* W, const float* bias, float* out,
                  int64_t out_stride, int B, int in_dim, int out_dim,
                  float w_scale, float b_scale, int pixelnorm, int lrelu, float act_gain,
                  float out_scale, float out_shift, const float* trunc_mean, float trunc_psi,
                  int out_repeat, int64_t out_repeat_stride, void* stream);

/* PixelNorm on its own (models/model_v3.py:32-37): y[b] = x[b] * rsqrt(mean(x[b]^2) + 1e-8).  The generator's mapping
 * path folds it into its first cips3d_linear (pixelnorm = 1); this entry serves `G.style_decoder(z)` called as a module. */
int cips3d_pixel_norm(const float* x, float* y, int B, int C, void* stream);

/* A table of independent dense layers evaluated in ONE launch (FiLM gamma/beta heads and the
 * decoder's per-layer style modulations).  The table itself lives in device memory. */
typedef struct cips3d_linear_desc {
  const float* W;        /* [out_dim, in_dim] */
  const float* bias;     /* [out_dim] or NULL */
  const float* x;        /* row b at x + b * x_stride */
  float* out;            /* row b at out + b * out_stride */
  int64_t x_stride;
  int64_t out_stride;
  int32_t in_dim;
  int32_t out_dim;
  float w_scale;
  float b_scale;
  float out_scale;       /* applied after the bias: y * out_scale + out_shift */
  float out_shift;
  int32_t row_begin;     /* exclusive prefix sum of out_dim over the table */
  int32_t pad_;
} cips3d_linear_desc;

int cips3d_linear_table(const cips3d_linear_desc* table_dev, int n_desc, int total_rows, int B,
                        void* stream);

/* Backward of cips3d_linear_table (heads without activation: y = (w_scale W x + b_scale bias) out_scale + out_shift), all
 * heads in two launches.  The table is the forward's; tensors of the backward sit at the same offsets as the forward's:
 *   dy at dy_base + (desc.out - out_base)   [row b at + b * desc.out_stride]
 *   dx at dx_base + (desc.x - x_base)       [row b at + b * desc.x_stride]; ACCUMULATED with atomics (heads share inputs):
 *                                            the caller zeroes dx; NULL = not wanted
 *   dW flat, head i at dW + w_offsets[i] ([out_dim, in_dim]; w_offsets = device array of int64 element offsets); NULL = not wanted
 *   db flat [total_rows] (rows of heads without bias receive their value all the same); NULL = not wanted
 * Every head must have the same in_dim (% 4 == 0). */
int cips3d_linear_table_bwd(const cips3d_linear_desc* table_dev, int n_desc, int total_rows, int in_dim, int B,
                            const float* out_base, const float* dy_base, const float* x_base, float* dx_base,
                            const int64_t* w_offsets_dev, float* dW, float* db, void* stream);

#ifndef CIPS3D_AMAX_SLOTS
#define CIPS3D_AMAX_SLOTS 8       /* see cips3d_range below: slots of one (tensor, sample) */
#endif
#ifndef CIPS3D_AMAX_STRIDE
#define CIPS3D_AMAX_STRIDE 64     /* floats between slots: 256 bytes, the slots of an array sit in different memory channels */
#endif
#define CIPS3D_AMAX_FLOATS (CIPS3D_AMAX_SLOTS * CIPS3D_AMAX_STRIDE)
#define CIPS3D_FEATURES_EXP (-14)
#define CIPS3D_PLANES_EXP_BLOCK 128   /* pixels that share one exponent of a split-fp16 planes tensor */

/* ------------------------------------------------------------------ camera */

/* locations [B,2] = (azim, elev); fov_deg [B] or NULL (then fov_deg_scalar); up [B,3] or NULL
 * (then (0,1,0)).  Writes extrinsics [B,3,4] = [R^T | T], focal/near/far [B]. */
int cips3d_camera_params(const float* locations, const float* fov_deg, float fov_deg_scalar,
                         const float* up, float dist_radius, int img_size, int B,
                         float* extrinsics, float* focal, float* near_, float* far_, void* stream);

/* ------------------------------------------------------------------ NeRF renderer */

/* Re-pack the FiLM-SIREN weight matrices into the MFMA operand order the render kernel streams
 * (once per weight update).  hidden = H (32, 64, 128 or 256), depth = D >= 1.
 *   w_hidden : D-1 matrices [H,H] (pts_linears.1..D-1.weight), concatenated
 *   w_view   : [H, H+3] (views_linears.weight)
 * packed     : cips3d_nerf_packed_floats(H, D) floats: per layer (hidden layers 1..D-1, then the H x H part of the view
 *              layer) the matrix scaled by a power of two 2^s into fp16's normal range and split into fp16 hi + lo halves
 *              (w 2^s = hi + lo, 22 significant bits) in v_mfma_f32_16x16x32_f16 A-fragment order, 4 bytes per weight;
 *              then (2^s, 2^-s) per layer.  The render kernel accumulates the three exact fp16 products
 *              w_hi x_hi + w_hi x_lo + w_lo x_hi in fp32 (fp32-equivalent results at 16/3 of the fp32 MFMA rate). */
int cips3d_nerf_pack_weights(const float* w_hidden, const float* w_view, float* packed,
                             int hidden, int depth, void* stream);
int64_t cips3d_nerf_packed_floats(int hidden, int depth);

typedef struct cips3d_nerf_params {
  /* geometry, per view */
  const float* cam_poses;   /* [B,3,4] */
  const float* focals;      /* [B] */
  const float* near_;       /* [B] */
  const float* far_;        /* [B] */
  const float* perturb_u;   /* [B, img*img] per-ray uniform in [0,1) or NULL (perturb = False) */
  /* network (all device pointers straight into the module's parameters) */
  const float* w_first;     /* [H,3]    pts_linears.0.weight */
  const float* packed;      /* cips3d_nerf_pack_weights output */
  const float* w_view;      /* [H,H+3]  views_linears.weight (columns H..H+2 = view direction) */
  const float* film;        /* [B, D+1, 2, H]: gamma = 15*g+30, beta = 0.25*b per layer */
  const float* layer_bias;  /* [D+1, H]: pts_linears.i.bias rows, then views_linears.bias */
  const float* w_sigma;     /* [H]      sigma_linear.weight */
  const float* w_rgb;       /* [3,H]    rgb_linear.weight */
  const float* b_sigma;     /* [1] */
  const float* b_rgb;       /* [3] */
  const float* sigmoid_beta;/* [1] */
  int32_t B, img_size, n_samples, hidden, depth;
  int32_t static_viewdirs;
  int32_t n_chunks;         /* sample chunks per ray (partials per ray) */
  int32_t n_rays;           /* 0: R = img_size^2 rays generated from the camera; > 0: explicit-geometry mode with R = n_rays */
  /* outputs */
  float* part;              /* [n_chunks, B, H+8, R] partial composites (see nerf.hip) */
  float* sdf;               /* [B, R, n_samples] or NULL */
  /* explicit-geometry mode = the reference's VolumeFeatureRenderer.forward(pts, rays_d, viewdirs, z_vals, ...)
   * (cips3d/volume_renderer.py:192-303): all four or none; cam_poses / focals / perturb_u are then unused */
  const float* x_pts;       /* [B, R, n_samples, 3] un-normalised sample points */
  const float* x_rays_d;    /* [B, R, 3] */
  const float* x_viewdirs;  /* [B, R, 3] (already normalised) */
  const float* x_z_vals;    /* [B, R, n_samples] */
  /* optional: final maps written by the render kernel itself (all four or none).  When cips3d_nerf_fuses_finish() says
   * yes for this shape, the eight chunk waves of a ray group meet in LDS, combine their partials in sample order (the
   * arithmetic of cips3d_nerf_finish, bit for bit) and write features [B,H,R], thumb_rgb [B,3,R], xyz [B,3,R],
   * mask [B,2,R]; `part` is then not touched and cips3d_nerf_finish must not be called.  Otherwise the pointers are
   * ignored and `part` + cips3d_nerf_finish is the way. */
  float* o_features; float* o_thumb; float* o_xyz; float* o_mask;
  /* != 0 (fused finish only): o_features receives the feature map as split-fp16 planes [B][H/8][hi|lo][R][8] (fp16), the
   * input format of cips3d_modconv1x1_planes, instead of fp32 [B,H,R] (same bytes).  The planes hold features * 2^14: every
   * pixel block has the exponent CIPS3D_FEATURES_EXP (cips3d_range.x_exp_const; a feature is a convex combination of sines,
   * |f| <= 1) */
  int32_t features_planes;
  /* != 0: with_sdf = False (cips3d/nerf_utils.py:288-297, noise-free): the sigma head's output is a raw density,
   * alpha = 1 - exp(-softplus(raw) * delta) (torch's softplus: the identity above 20); sigmoid_beta is then not read.
   * Forward only: the differentiable forward (stash / bwd_*) refuses it */
  int32_t raw_density;
  /* Differentiable forward (camera-driven mode only; all three or none): the render kernel additionally writes what
   * cips3d_nerf_bwd_fused needs, so that the backward does not recompute the forward -- per MFMA layer the fp32 accumulators
   * (`stash`: cips3d_nerf_bwd_fused_stash_floats(B, img_size, n_samples, hidden, depth, n_chunks) floats, the layout of that
   * call, which must then use the same n_chunks) and per point, p = sample * R + ray, the sdf (`bwd_sdf` [B,P]) and the rgb
   * logits (`bwd_crgb` [B,3,P]). */
  float* stash;
  float* bwd_sdf;
  float* bwd_crgb;
  /* != NULL: exact-fp32 arithmetic (the reference's F.linear in IEEE fp32, cips3d/volume_renderer.py:15-35, 74-85): the point
   * MLP's GEMMs run on the fp32 matrix instruction (v_mfma_f32_16x16x4_f32: bit for bit an fmaf chain in k order) over the
   * stream cips3d_nerf_pack_weights32 wrote, instead of three fp16 products per fp32 product over `packed` (which is then not
   * read).  hidden == 256, no stash (CIPS3D_E_UNSUPP otherwise); camera-driven or explicit geometry, fused or separate finish. */
  const float* packed32;
  /* optional (both or neither): n_zero_words floats the launch leaves ZEROED -- scratch that the caller's next launches expect
   * cleared (cips3d_generator_forward: the decoder's measured range rows in a frame of a sequence).  The default kernel's fused
   * finish clears them on its way out, any other form of the launch with one small kernel in front.  Nothing of the render
   * launch itself may live there. */
  float* zero_words;
  int64_t n_zero_words;
  /* != 0 (fused finish only; CIPS3D_E_UNSUPP when the shape does not fuse): o_mask is laid out [2, B, R] -- the background
   * weight of every view, then -|xyz| of every view: the two maps `Generator.forward` returns, each contiguous, without the
   * transposing copy a batch of [B, 2, R] needs */
  int32_t mask_planar;
  int32_t pad3_;
} cips3d_nerf_params;

/* The exact-fp32 weight stream of cips3d_nerf_params.packed32 (cips3d_nerf_packed_floats(hidden, depth) floats, as `packed`):
 * fp32 A fragments of v_mfma_f32_16x16x4_f32 per (layer, 16 output units, 16 input units), unscaled. */
int cips3d_nerf_pack_weights32(const float* w_hidden, const float* w_view, float* packed32, int hidden, int depth, void* stream);

/* 1 when cips3d_nerf_render(p) will write p->o_* itself (o_* set, n_chunks == 8, LDS large enough), else 0 */
int cips3d_nerf_fuses_finish(const cips3d_nerf_params* p);

/* Chunk count the render kernel wants for (B, n_samples): enough workgroups to fill the chip. */
int cips3d_nerf_suggest_chunks(int B, int img_size, int n_samples);
/* floats needed for `part` */
int64_t cips3d_nerf_part_floats(int B, int img_size, int hidden, int n_chunks);

/* rays -> samples -> FiLM-SIREN MLP (fp32 MFMA) -> per-chunk alpha compositing. */
int cips3d_nerf_render(const cips3d_nerf_params* p, void* stream);

/* Ordered combination of the chunk partials into the final maps:
 *   features [B,H,R] (channel-major = NCHW), thumb_rgb [B,3,R], xyz [B,3,R], mask [B,2,R]. */
int cips3d_nerf_finish(const float* part, int n_chunks, int B, int img_size, int hidden,
                       float* features, float* thumb_rgb, float* xyz, float* mask, void* stream);
/* the same for an arbitrary ray count per view (explicit-geometry mode): outputs are [B, C, n_rays] */
int cips3d_nerf_finish_rays(const float* part, int n_chunks, int B, int n_rays, int hidden, float* features,
                            float* thumb_rgb, float* xyz, float* mask, void* stream);

/* ------------------------------------------------------------------ decoder */

/* Range tracking of the split-fp16 modes (CIPS3D_GEMM_SPLIT, planes).  An fp32 activation is carried as two fp16 numbers; fp16
 * has 5 exponent bits, the reference's fp32 convolution (models/model_v3.py:296-312) 8.  Every split therefore happens on
 * x * 2^-e with one power of two per (tensor, sample) that puts a rigorous bound of max|x| into [2^14, 2^15); the consumer
 * undoes it exactly on its accumulators.  Two kinds of small per-sample device arrays carry what the kernels need:
 *   amax  [B][CIPS3D_AMAX_FLOATS] fp32: max |x| of a tensor's true values = the maximum over its CIPS3D_AMAX_SLOTS slots
 *         (slot s at float s * CIPS3D_AMAX_STRIDE: one 64-byte line each, so that the atomics of different workgroups do not
 *         queue on one line).  The producing epilogue raises a slot with one atomicMax per workgroup (the caller zeroes the
 *         array before the producer runs); cips3d_absmax fills it for a tensor that exists already.
 *   exp   [B][ceil(HW / CIPS3D_PLANES_EXP_BLOCK)] int32: the exponents a planes tensor was stored with, one per block of 128
 *         consecutive pixels (stored value = x * 2^-e; GEMM columns are independent, so every pixel block may have its own),
 *         written by its producer.
 *   pmax  [B][ceil(HW / 64)][C / 16] fp32: max |x| of every (16 channels x 64 pixels) patch of a planes tensor, written by its
 *         producer with one plain store per wave (no atomics, nothing to zero).  cips3d_modconv1x1_planes takes max|in| of a
 *         128-pixel block from the 2 x Cin/16 entries that cover it, so a planes OUTPUT's exponent needs no word from any
 *         other workgroup of the launch.
 *   lconst[B][4] fp32 {c0, c1, l1, -}: the layer's bound constants, |out| <= max(c1, sqrt(2) l1) * max|in| + c0 with
 *         c0 = sqrt(2) (|noise_w| * noise_bound + max|bias|), c1 = sqrt(2) * (sqrt(Cin) for a demodulated conv; the FIR's
 *         largest polyphase gain for the 2x up-sampler), l1 = the largest row L1 norm of a NON-demodulated weight (measured
 *         by the modulate kernel, else 0).  Written by cips3d_range_consts or by cips3d_modulate_table.
 * A kernel that must scale values it produces itself (planes output, the intermediates of a fused up-sampling stage) uses
 * bound = c1 * max|in| + c0; a kernel that splits an existing fp32 tensor uses bound = amax(in).
 * Passing rg == NULL keeps e = 0 everywhere (legal only for callers that know max|x| of every operand to lie in
 * [2^-3, 6e4]); the Python host always passes it. */
typedef struct cips3d_range {
  const float* x_amax;     /* [B][CIPS3D_AMAX_FLOATS] of an fp32 input activation (split GEMM, fused stage)        */
  const int32_t* x_exp;    /* planes input: its exponents [B][blocks]; NULL: every block has x_exp_const            */
  int32_t x_exp_const;     /* (the NeRF feature map: CIPS3D_FEATURES_EXP)                                           */
  float x_max_const;       /* planes input without x_pmax: a bound of max|x| the caller vouches for (feature map: 1);
                              0: 2^(e + 15), the bound the input's own exponent encodes (looser by ~2^5)             */
  const float* x_pmax;     /* planes input: its patch maxima [B][ceil(HW/64)][Cin/16], or NULL (x_max_const)        */
  const float* lconst;     /* [B][4] of this layer (planes output; conv1 of a fused stage)                         */
  const float* lconst2;    /* [B][4] of conv2 of a fused stage (needed when wm_next != NULL)                       */
  float* out_amax;         /* [B][CIPS3D_AMAX_FLOATS] or NULL: raised to max |out| (GEMM entry points; not the fused stage)      */
  int32_t* out_exp;        /* planes output: receives the exponents this launch chose, [B][blocks]                  */
  float* next_amax;        /* [B][CIPS3D_AMAX_FLOATS] or NULL: raised to max |y_next| (fused stage with wm_next)                 */
  float* out_pmax;         /* planes output: receives its patch maxima [B][ceil(HW/64)][Cout/16]                    */
  float next_gain;         /* fused stage, > 0: next_amax receives the BOUND next_gain * U2 of |y_next| (U2: the stage's bound of
                              |out2|; next_gain = sqrt(C) for a demodulated next up-conv) in slot 0 instead of the measured
                              maximum -- no workgroup reduction, no atomics; for a consumer that is the decoder's last stage */
  int32_t half_chip;       /* cips3d_modconv1x1_planes, != 0: launches of another, independent view are in flight on another stream
                              -- a launch that fills the chip exactly once (256 tiles of 64 x 128) runs as 128 tiles of 128 x 128
                              on half the CUs instead, side by side with the other view's (same results, same ToRGB slots) */
  const struct cips3d_reduce_job* ride;   /* cips3d_modconv1x1_planes only (HOST pointer, may be NULL): a ToRGB fold that rides on
                              this launch -- see cips3d_reduce_job */
} cips3d_range;
/* The job of one cips3d_torgb_reduce (same arithmetic, same fixed order, same bits) carried by a split-planes GEMM launch that
 * does not depend on it and leaves CUs free: one extra row of workgroups of that launch does the fold, so it costs no launch
 * of its own (4.7 us at 64^2 on MI355X).  out[b][r][n] = skip + sum_s part[s] + sum_k bias[k][r] with
 * n4 = B * 3 * HW / 4 float4 positions, HW4 = HW / 4, slot_stride = B * 3 * HW; n_slots <= 48, n_bias <= CIPS3D_TORGB_FOLD_MAX;
 * part / skip / out must not alias the GEMM's operands.  (CIPS3D_TORGB_FOLD_MAX is defined below: 8.) */
typedef struct cips3d_reduce_job {
  const float* part;
  const float* bias[8];
  const float* skip;       /* may be NULL */
  float* out;
  int64_t n4, HW4, slot_stride;
  int32_t n_slots, n_bias;
} cips3d_reduce_job;
/* amax[b][*] = max_i |x[b][i]|, x [B][n] (zeroes the slots, then one pass; HBM-bound) */
int cips3d_absmax(const float* x, int B, int64_t n, float* amax, void* stream);
/* the same pass over slots the caller has zeroed already (or that hold maxima this tensor's are to be merged with): one launch */
int cips3d_absmax_raise(const float* x, int B, int64_t n, float* amax, void* stream);
/* Test hook for the split itself (n even): words[i] = {hi | lo << 16} of x[i] * k -- even i through the fused form (the exact
 * product t k is split), odd i through the plain form on the fp32 product -- and, per pair (x[2j], x[2j+1]),
 * pairs[2j] = [hi_0 | hi_1 << 16], pairs[2j+1] = [lo_0 | lo_1 << 16] of the values themselves. */
int cips3d_split_words(const float* x, float k, uint32_t* words, uint32_t* pairs, int64_t n, void* stream);
/* CIPS3D_AMAX_SLOTS / CIPS3D_AMAX_STRIDE as the library was built (a binding sizes its amax arrays from these) */
int cips3d_amax_layout(int* slots, int* stride);
/* lconst[b] = {c0, c1, 0, 0} from a layer's parameters (device pointers; noise_w / bias / fir may be NULL):
 * c1 = sqrt(2) * (fir ? largest polyphase gain of the 4x4 FIR : w_gain), w_gain = sqrt(Cin) for a demodulated conv.
 * The bound of |noise| is max(noise_bound, the maximum of noise_amax), noise_amax = the slots of ONE sample written by
 * cips3d_absmax over the whole noise tensor (or NULL): no host round trip for a noise map that exists only on the device. */
int cips3d_range_consts(const float* bias, int n_bias, const float* noise_w, float noise_bound, const float* noise_amax,
                        float w_gain, const float* fir, float* lconst, int B, void* stream);

/* Modulated (and optionally demodulated) weights of one conv for every sample:
 *   wm[b][o][i][t] = scale * W[o][i][t] * s[b][i];  wm[b][o] *= rsqrt(sum wm[b][o]^2 + 1e-8)
 * W [Cout, Cin, k*k], s [B, s_stride] (first Cin used), wm [B, Cout, Cin, k*k].
 * flags: bit 0 = demodulate; bit 1 = emit the MFMA A-fragment order cips3d_modconv1x1 consumes
 * (k = 1, Cout % 16 == 0, Cin % 16 == 0; A operand of v_mfma_f32_16x16x4_f32, four k-steps per 16 bytes):
 *   wm[b][o/16][i/16][((i&3)<<4 | (o&15))*4 + ((i>>2)&3)]. */
#define CIPS3D_MOD_DEMODULATE 1
#define CIPS3D_MOD_PACKED     2
/* with CIPS3D_MOD_PACKED: the A-fragment order whose k-steps follow the MFMA *D* layout, so that a GEMM result still in
 * registers is the B operand of the next GEMM (cips3d_fused_up_conv_next):
 *   wm[b][o/16][i/16][(((i>>2)&3)<<4 | (o&15))*4 + (i&3)]. */
#define CIPS3D_MOD_CHAINED    4
/* with CIPS3D_MOD_PACKED and ksq == 9: tap t is stored in slot 8 - t (the 180-degree rotated kernel the up-sampling branch
 * of cips3d_modconv3x3 correlates with).  Packed ksq == 9 layout: wm[b][tap][o/16][i/16][((i&3)<<4 | (o&15))*4 + ((i>>2)&3)]. */
#define CIPS3D_MOD_FLIP       8
/* with CIPS3D_MOD_PACKED and ksq == 1 (Cin % 32 == 0): split-fp16 A fragments for CIPS3D_GEMM_SPLIT -- 2^8 wm as fp16 hi + lo
 * halves in v_mfma_f32_16x16x32_f16 order, wm[b][o/16][i/32][plane][(((i>>3)&3)<<4 | (o&15))][i&7] (fp16), 4 bytes per weight */
#define CIPS3D_MOD_SPLIT      16
/* with CIPS3D_MOD_PACKED [| CIPS3D_MOD_CHAINED] and ksq == 1: split-fp16 fragments for the fused up-sampling stage in
 * CIPS3D_GEMM_SPLIT mode (16-channel k-groups, v_mfma_f32_16x16x16_f16): the element <-> channel map of the fp32 layout, each
 * lane's 16 bytes = {fp16 hi x 4 | fp16 lo x 4} of 2^8 wm */
#define CIPS3D_MOD_SPLIT16    32
/* with CIPS3D_MOD_PACKED and ksq == 1: the packed form of wm^T ([Cin x Cout]: the A operand of the data-gradient GEMM
 * dx = wm^T dy) instead of wm's -- what cips3d_pack_weights(transpose = 1) makes from the plain matrix; fp32 fragments, or
 * with CIPS3D_MOD_SPLIT (Cout % 32 == 0) the split-fp16 ones (cips3d_pack_weights(transpose = 3)) */
#define CIPS3D_MOD_TRANSPOSE  64
/* with CIPS3D_MOD_PACKED and ksq == 1 (Cin % 32 == 0): bf16 A fragments (bf16(wm), round to nearest even, unscaled) for
 * cips3d_modconv1x1_planes16 -- the bf16 decoder mode's form of CIPS3D_MOD_SPLIT */
#define CIPS3D_MOD_BF16       128
/* OR-ed into `epilogue` of cips3d_modconv1x1 / into `skip_up` of cips3d_fused_up_conv: bf16 compute mode of the GEMM
 * (operands rounded to bf16 in registers, v_mfma_f32_16x16x16_bf16, fp32 accumulate; storage stays fp32).  This is the
 * decoder precision of BASELINE config 3; the default (flag absent) is exact fp32. */
#define CIPS3D_GEMM_BF16      0x100
/* OR-ed into `epilogue` of cips3d_modconv1x1[_torgb]: fp32-EQUIVALENT split-fp16 mode.  `wm` must come from
 * cips3d_modulate_weights(..., CIPS3D_MOD_PACKED | CIPS3D_MOD_SPLIT); activations are split x = fp16(x) + fp16(x - fp16(x)) in
 * registers and each product is accumulated in fp32 as three exact fp16 products (w_lo x_hi + w_hi x_lo + w_hi x_hi) on
 * v_mfma_f32_16x16x32_f16: results agree with the fp32 MFMA path to ~1e-6 relative (below fp32's own accumulation
 * rounding) at a fraction of its matrix time.  Storage and epilogues are unchanged fp32. */
#define CIPS3D_GEMM_SPLIT     0x400
/* bf16 STORAGE of the low-resolution GEMM result of an up-sampling stage (the tensor the 2x FIR reads): OR-ed into `epilogue`
 * of cips3d_modconv1x1 (epilogue 0 only) `out` is written as bf16 [B,Cout,HW]; OR-ed into `skip_up` of cips3d_fused_up_conv[_next]
 * (together with CIPS3D_GEMM_BF16) `y_lo` is read and `y_next` is written as bf16.  FIR, epilogues and accumulation stay
 * fp32.  Halves the activation bytes the >= 128^2 stages move through HBM (BASELINE config 3). */
#define CIPS3D_Y_BF16         0x200
/* OR-ed into `skip_up` of cips3d_fused_up_conv[_next]: `rgb` points to a uint8 [B,3,2H,2W] image and receives
 * cips3d_rgb_to_uint8(rgb) -- clamp to [-1, 1], (c + 1) * 127.5, round to nearest even -- instead of the fp32 values (bit-identical
 * to converting the stored fp32 image; the reference's img_tensor_to_pil step, models/render_video_web_v10.py:1825-1826). */
#define CIPS3D_RGB_U8         0x800
/* OR-ed into `skip_up` of cips3d_fused_up_conv[_next]: a FLAT stage -- a decoder block that does not up-sample,
 * [StyledConv, StyledConv, ToRGB] at one resolution (the blocks above the last entry of `upsample_list`: a 256^2 generator still
 * walks the 512 and 1024 blocks, at 256^2; models/model_v3.py:553-590).  `y_lo` is then conv1's GEMM result at the OUTPUT size
 * H x W, `fir` is not read (may be NULL), the outputs are H x W, `skip` (if any) is [B,3,H,W] and is added as it is (bit 0 of
 * skip_up must be clear); rg->x_amax = the maximum of |y_lo| as always, rg->lconst[b][1] (a row gain relative to conv1's INPUT)
 * is not used.  Supported shapes: cips3d_fused_flat_conv_supported. */
#define CIPS3D_STAGE_FLAT     0x1000
int cips3d_modulate_weights(const float* W, const float* s, int64_t s_stride, float* wm,
                            int B, int Cout, int Cin, int ksq, float scale, int flags,
                            void* stream);

/* The same for a table of convs in ONE launch (the whole decoder).  The table lives in device memory;
 * `s` already points at the conv's slice of the style-modulation buffer. */
typedef struct cips3d_modulate_desc {
  const float* W;        /* [Cout, Cin, ksq] */
  const float* s;        /* row b at s + b * s_stride */
  float* out;            /* [B, Cout*Cin*ksq] (plain or packed per flags) */
  int64_t s_stride;
  int32_t Cout, Cin, ksq, flags;
  float scale;
  int32_t row_begin;     /* exclusive prefix sum of Cout over the table */
  /* range constants of the layer this conv belongs to (all optional; lconst NULL = none wanted): the launch writes
   * lconst[b] = {c0, c1, l1, 0} (see cips3d_range) -- c0 / c1 by the wave of row 0, l1 raised by every row of a
   * non-demodulated conv (lconst must then be zero before the launch) */
  float* lconst;         /* [B][4] */
  const float* bias;     /* [n_bias] FusedLeakyReLU bias of the StyledConv */
  const float* noise_w;  /* [1] */
  const float* fir;      /* [16]: the conv up-samples; c1 follows the FIR, not the weight */
  int32_t n_bias, pad_;
} cips3d_modulate_desc;

/* noise_bound: an upper bound of |noise| over every noise map of the call (cips3d_rng_fill draws stay below 5.77) */
int cips3d_modulate_table(const cips3d_modulate_desc* table_dev, int n_desc, int total_rows, int B, float noise_bound,
                          void* stream);

/* 1 when cips3d_modconv1x1 tiles this shape (Cin % 32 == 0, Cout % 32 == 0, HW % 4 == 0). */
int cips3d_modconv1x1_supported(int Cin, int Cout, int64_t HW);

/* 1x1 modulated convolution as a per-sample GEMM with a fused epilogue:
 *   y[b][o][n] = sum_i wm[b][o][i] * x[b][i][n]
 *   epilogue 0: out = y                                              (raw, feeds the FIR up-sampler)
 *   epilogue 1: out = lrelu(y + noise_w * noise[n] + bias[o], 0.2) * sqrt(2)
 * x [B,Cin,HW], wm in the PACKED order above, out [B,Cout,HW], noise [HW] (noise_bstride = 0) or
 * per-sample [B,HW] (noise_bstride = HW). */
int cips3d_modconv1x1(const float* x, const float* wm, float* out, int B, int Cin, int Cout,
                      int64_t HW, int epilogue, const float* noise, int64_t noise_bstride,
                      const float* noise_w, const float* bias, const cips3d_range* rg, void* stream);

/* The same GEMM with the ToRGB that FOLLOWS this conv folded into its epilogue (models/model_v3.py:602-632: ToRGB reads
 * the conv's output): every workgroup also writes the partial sums of its block of output rows,
 *   rgb_part[row_block][b][r][n] = sum_{o in block} rgb_w[b][r][o] * out[b][o][n]      (rgb_w plain [B,3,Cout])
 * and *n_row_blocks receives the number of row blocks (slots) written (the tiling -- 32 to 128 rows per block -- follows the shape
 * and the batch: size rgb_part for Cout / 32 blocks and fold what *n_row_blocks reports).  cips3d_torgb_reduce folds the slots of one or
 * several such layers, their biases and the skip image in a FIXED order (deterministic), replacing cips3d_torgb launches
 * that re-read the activations.  rgb_w == rgb_part == NULL: plain cips3d_modconv1x1. */
#define CIPS3D_TORGB_FOLD_MAX 8
int cips3d_modconv1x1_torgb(const float* x, const float* wm, float* out, int B, int Cin, int Cout, int64_t HW, int epilogue,
                            const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias,
                            const float* rgb_w, float* rgb_part, int* n_row_blocks, const cips3d_range* rg, void* stream);
/* (rg: CIPS3D_GEMM_SPLIT splits x * 2^-e, e from rg->x_amax; every mode raises rg->out_amax to max |out| when given)
 * out[b][r][n] = skip[b][r][n] + sum_{s < n_slots} part[s][b][r][n] + sum_{k < n_bias} biases[k][r]
 * (part = consecutive slots of [B,3,HW]; biases = HOST array of device pointers to [3]; skip may be NULL) */
int cips3d_torgb_reduce(const float* part, int n_slots, const float* const* biases, int n_bias, const float* skip, float* out,
                        int B, int64_t HW, void* stream);

/* 2x FIR up-sampling of a low-resolution conv result fused with the StyledConv epilogue:
 *   u   = upfirdn2d(y_lo, fir, up=2, pad=(2,1))      (fir = outer([1,3,3,1])/64*4, [4,4] device)
 *   out = lrelu(u + noise_w * noise + bias[c], 0.2) * sqrt(2)
 * y_lo [B,C,H,W] -> out [B,C,2H,2W].  out_amax: per-sample amax rows of `out` (cips3d_range; raised, not zeroed) or NULL. */
int cips3d_up2_fir_act(const float* y_lo, const float* fir, float* out, int B, int C, int H, int W,
                       const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias,
                       float* out_amax, void* stream);

/* StyledConv epilogue on its own (used after the k x k path):
 *   out = lrelu(x + noise_w * noise + bias[c], 0.2) * sqrt(2);  x/out [B,C,HW]. */
int cips3d_noise_bias_act(const float* x, const float* noise, int64_t noise_bstride, const float* noise_w,
                          const float* bias, float* out, int B, int C, int64_t HW, void* stream);

/* ToRGB: out[b][c][n] = sum_i wm[b][c][i] * x[b][i][n] + bias[c] + skip_term, where skip_term is
 * absent (skip NULL), skip[b][c][n] (skip_up = 0) or upfirdn2d(skip, fir, up=2, pad=(2,1)) taken
 * from the half-resolution skip [B,3,H/2,W/2] (skip_up = 1).  x [B,Cin,H,W], wm [B,3,Cin]. */
int cips3d_torgb(const float* x, const float* wm, const float* bias, const float* skip, int skip_up,
                 const float* fir, float* out, int B, int Cin, int H, int W, void* stream);


/* One kernel for a whole up-sampling stage after its low-resolution GEMM (models/model_v3.py:618-630 for a
 * stage in `upsample_list`):
 *   act1 = lrelu(upfirdn2d(y_lo, fir, up=2, pad=(2,1)) + nw1*noise1 + bias1) * sqrt(2)     (never stored)
 *   out2 = lrelu(wm2 (1x1) act1 + nw2*noise2 + bias2) * sqrt(2)            -> out2 [B,C,2H,2W] unless NULL
 *   rgb  = wm_rgb (3 x C) out2 + bias_rgb + (skip_up ? upfirdn2d(skip) : skip)   -> rgb [B,3,2H,2W] unless wm_rgb NULL
 * y_lo [B,C,H,W]; wm2 in the PACKED order of cips3d_modulate_weights, wm_rgb plain [B,3,C]; skip [B,3,H,W] when
 * skip_up else [B,3,2H,2W] (or NULL).  C in {32,64,128,256}, W % 32 == 0, H % 2 == 0. */
int cips3d_fused_up_conv_supported(int C, int H, int W);
/* ... and for a FLAT stage (CIPS3D_STAGE_FLAT in `skip_up`; H x W = the one resolution of the block): C in {32,64,128,256},
 * W % 64 == 0, H % 4 == 0.  The same kernel without the FIR: act1 = lrelu(y_lo + nw1*noise1 + bias1) * sqrt(2), skip added as is. */
int cips3d_fused_flat_conv_supported(int C, int H, int W);
/* The stage above plus the low-resolution GEMM of the NEXT up-sampling stage (its `StyledConv(up)` 1x1 conv, C -> C/2,
 * which reads out2): y_next[b] = wm_next[b] (C/2 x C) out2[b], taken from the registers that hold out2, so out2 need not be
 * stored (pass NULL) and the next stage starts at its own cips3d_fused_up_conv with y_lo = y_next.  wm_next in the
 * CIPS3D_MOD_PACKED | CIPS3D_MOD_CHAINED order, y_next [B,C/2,2H,2W].  wm_next == y_next == NULL: plain cips3d_fused_up_conv.
 * cips3d_fused_up_conv_chains(C): 1 for the widths that have the chained form. */
int cips3d_fused_up_conv_chains(int C);
int cips3d_fused_up_conv_next(const float* y_lo, const float* fir, const float* noise1, int64_t noise1_bstride,
                              const float* noise_w1, const float* bias1, const float* wm2, const float* noise2,
                              int64_t noise2_bstride, const float* noise_w2, const float* bias2, float* out2,
                              const float* wm_rgb, const float* bias_rgb, const float* skip, int skip_up, float* rgb,
                              const float* wm_next, float* y_next, int B, int C, int H, int W, const cips3d_range* rg,
                              void* stream);
int cips3d_fused_up_conv(const float* y_lo, const float* fir, const float* noise1, int64_t noise1_bstride,
                         const float* noise_w1, const float* bias1, const float* wm2, const float* noise2,
                         int64_t noise2_bstride, const float* noise_w2, const float* bias2, float* out2,
                         const float* wm_rgb, const float* bias_rgb, const float* skip, int skip_up, float* rgb,
                         int B, int C, int H, int W, const cips3d_range* rg, void* stream);
/* (rg, CIPS3D_GEMM_SPLIT mode: x_amax = amax of y_lo, lconst = conv1's constants, lconst2 = conv2's; next_amax <- max |y_next|.
 * out2 is not tracked -- the register budget of the C = 32 stage; use cips3d_absmax before a split GEMM reads a stored out2) */

/* General k x k modulated convolution (k odd, padding k/2), direct form; used for k = 3 configs.
 * transpose2 = 1 computes conv_transpose2d(stride 2, padding 0): out is (2H-1+k-1)^2. */
int cips3d_modconv_kxk(const float* x, const float* wm, float* out, int B, int Cin, int Cout,
                       int H, int W, int k, int transpose2, void* stream);

/* The same 1x1 modulated conv on activations that are STORED split ("planes": x = fp16 hi + fp16 lo, laid out
 * [B][C/8][plane hi|lo][HW][8] fp16 = the consumer's MFMA B fragments, 4 bytes per value like fp32).  The producer's epilogue
 * splits every output once, so the main loop is LDS reads + MFMA only (csrc/chain.hip); used for the run of equal-resolution
 * StyledConvs at the NeRF resolution.  wm: CIPS3D_MOD_PACKED | CIPS3D_MOD_SPLIT.  out_format: 0 fp32 [B,Cout,HW], 1 planes,
 * 2 bf16 [B,Cout,HW].  epilogue / noise / bias / rgb_w / rgb_part / n_row_blocks as cips3d_modconv1x1_torgb (row blocks of 64).
 * cips3d_planes_supported: Cin % 64 == 0, Cout % 64 == 0. */
int cips3d_planes_supported(int Cin, int Cout, int64_t HW);
/* x_amax [B][CIPS3D_AMAX_FLOATS] (cips3d_absmax) -> planes of x * 2^-e with one e per sample, written to every pixel block's
 * entry of exp_out [B][blocks]; both NULL: e = 0.  pmax_out (optional, C % 16 == 0): the sample's maximum in every entry of the
 * tensor's patch maxima [B][ceil(HW/64)][C/16].  C % 8 == 0 */
int cips3d_to_planes(const float* x, void* planes, int B, int C, int64_t HW, const float* x_amax, int32_t* exp_out,
                     float* pmax_out, void* stream);
int cips3d_from_planes(const void* planes, float* x, int B, int C, int64_t HW, const int32_t* exp, void* stream);
/* rg: x_exp / x_exp_const (input planes); for a planes output x_pmax / x_max_const + lconst (-> its exponents, written to out_exp)
 * and out_pmax; out_amax (fp32 / bf16 output only: raised to max |out| per sample) */
int cips3d_modconv1x1_planes(const void* x_planes, const float* wm, void* out, int out_format, int B, int Cin, int Cout,
                             int64_t HW, int epilogue, const float* noise, int64_t noise_bstride, const float* noise_w,
                             const float* bias, const float* rgb_w, float* rgb_part, int* n_row_blocks, const cips3d_range* rg,
                             void* stream);

/* The bf16 decoder mode's form of the same run (BASELINE config 3: both GEMM operands rounded to bf16, fp32 accumulate --
 * the reference has no bf16 mode; the semantics are oracle/path.py:modulated_conv2d(bf16_gemm=True) over
 * models/model_v3.py:218-314).  "planes16" = [B][C/8][HW][8] bf16: one plane of the layout above, 2 bytes per value.  The stored
 * activation IS the rounded operand of the next layer's GEMM, so results equal the fp32-stored bf16 mode's.
 * wm: CIPS3D_MOD_PACKED | CIPS3D_MOD_BF16.  out_format: 0 fp32 [B,Cout,HW], 2 bf16 [B,Cout,HW], 3 planes16.  The other
 * arguments as cips3d_modconv1x1_planes (rgb_part row blocks of 64). */
int cips3d_to_planes16(const float* x, void* planes16, int B, int C, int64_t HW, void* stream);  /* C % 8 == 0 */
int cips3d_from_planes16(const void* planes16, float* x, int B, int C, int64_t HW, void* stream);
int cips3d_modconv1x1_planes16(const void* x_planes16, const float* wm, void* out, int out_format, int B, int Cin, int Cout,
                               int64_t HW, int epilogue, const float* noise, int64_t noise_bstride, const float* noise_w,
                               const float* bias, const float* rgb_w, float* rgb_part, int* n_row_blocks, const cips3d_range* rg,
                               void* stream);
/* (planes16: bf16 has fp32's exponent range, no scaling; rg->out_amax is still raised when given -- an fp32 / bf16 exit of the
 * run feeds a fused stage that may run split) */

/* 3x3 ModulatedConv2d (models/model_v3.py:264-314, decoder_cfg.kernel_size = 3) as an LDS-tiled implicit GEMM on MFMA.
 *   up = 0: out [B,Cout,H,W] = conv2d(x, wm, padding 1)                                           (:296-311)
 *   up = 1: out [B,Cout,2H,2W] = Blur(conv_transpose2d(x, wm, stride 2)), fir = the Blur's 4x4 taps (:280-291); the FIR is
 *           applied to the input tile in LDS (the two convolutions commute), nothing full-size is materialised
 * wm: cips3d_modulate_weights(..., ksq = 9, CIPS3D_MOD_PACKED [| CIPS3D_MOD_FLIP for up = 1]).
 * epilogue = 1 fuses NoiseInjection + bias + leaky ReLU * sqrt(2) as in cips3d_modconv1x1 (noise at the OUTPUT size).
 * epilogue | CIPS3D_GEMM_SPLIT: fp32-equivalent split-fp16 products (v_mfma_f32_16x16x32_f16, two taps per 32-deep step); wm then
 * additionally CIPS3D_MOD_SPLIT-packed (tap-pair fragments: B * Cout * Cin * 10 floats are written); rg->x_amax = the measured
 * per-sample maximum of x (cips3d_absmax; NULL: x is split unscaled -- for data of O(1)).  rg is not read otherwise.
 * cips3d_modconv3x3_supported: Cin % 16 == 0, Cout % 16 == 0, output width % 4 == 0 (plain: W % 4 == 0); other shapes use
 * cips3d_modconv_kxk. */
int cips3d_modconv3x3_supported(int Cin, int Cout, int H, int W, int up);
int cips3d_modconv3x3(const float* x, const float* wm, float* out, int B, int Cin, int Cout, int H, int W, int up,
                      const float* fir, int epilogue, const float* noise, int64_t noise_bstride, const float* noise_w,
                      const float* bias, const cips3d_range* rg, void* stream);

/* ------------------------------------------------------------------ whole forward, one call */

/* The generator forward (models/model_v3.py:875-1042) as ONE host call that enqueues every kernel of
 * the path on `stream`: mapping networks -> FiLM heads -> fused NeRF render -> finish -> all decoder
 * style modulations + weight modulations (two table launches) -> per layer GEMM / FIR / toRGB.
 * The plan is built once per (module, batch size, N_samples) by the host binding: it holds device
 * pointers into the module's parameters and into a persistent workspace; the per-call tensors travel
 * in cips3d_forward_io.  Only 1x1 convolutions whose shapes cips3d_modconv1x1_supported accepts can be
 * planned (every released v10 config); other configurations use the per-op entry points. */
#define CIPS3D_MAX_MAP_LAYERS 8
#define CIPS3D_MAX_DEC_LAYERS 40

typedef struct cips3d_dec_layer {
  int32_t kind;            /* 0 StyledConv, 1 StyledConv with 2x up-sampling, 2 ToRGB, 3 ToRGB + up-sampled skip */
  int32_t Cin, Cout, H, W; /* H, W = INPUT resolution of the layer (ToRGB: its own resolution) */
  int32_t noise_index;     /* index into cips3d_forward_io.noise (StyledConv) or -1 */
  int32_t flags;           /* bit 0 (kind 1, or kind 0 with bit 7): wm is in the CIPS3D_MOD_CHAINED order and this conv's GEMM is
                              computed by the previous stage's kernel (cips3d_fused_up_conv_next);
                              bit 1: wm is CIPS3D_MOD_SPLIT-packed and this layer's stand-alone GEMM runs in CIPS3D_GEMM_SPLIT mode;
                              bit 2: the layer's input is stored as split-fp16 planes, bit 3: its output is (cips3d_modconv1x1_planes);
                              bit 5 (with bit 2 / 3): the planes are bf16 planes16 and wm is CIPS3D_MOD_BF16-packed (cips3d_modconv1x1_planes16);
                              bit 4: wm is CIPS3D_MOD_SPLIT16-packed (conv2 / chained up-conv of a fused stage run in CIPS3D_GEMM_SPLIT mode);
                              bit 6: the conv is demodulated (its rows have unit norm: the sqrt(Cin) gain of the range bounds);
                              bit 7 (kind 0): head of a FLAT stage -- this conv, the next StyledConv and the ToRGB behind them (equal
                              widths, one resolution) run as one launch of the fused stage kernel (CIPS3D_STAGE_FLAT) */
  int32_t pad_;
  const float* wm;         /* this layer's modulated weights (workspace, written by the modulate table) */
  const float* bias;       /* activate.bias [Cout] or ToRGB.bias [3] */
  const float* noise_w;    /* NoiseInjection.weight [1] (StyledConv) */
  const float* fir;        /* 4x4 FIR (blur.kernel / upsample.kernel) for kinds 1 and 3 */
  /* range workspace rows of this layer (cips3d_range; StyledConvs of a plan whose decoder runs split-fp16, else NULL) */
  float* amax;             /* [B][CIPS3D_AMAX_FLOATS] max |output| (kind 1: of the low-resolution GEMM result y_lo) */
  int32_t* aexp;           /* [B][ceil(H*W / CIPS3D_PLANES_EXP_BLOCK)] exponents of a planes output */
  float* pmax;             /* [B][ceil(H*W / 64)][Cout / 16] patch maxima of a planes output */
  float* lconst;           /* [B][4] written by the modulate table */
} cips3d_dec_layer;

typedef struct cips3d_generator_plan {
  int32_t B, z_dim, n_map_r, n_map_d, style_dim_r, style_dim_d, n_latent, n_dec_layers;
  /* mapping networks: MappingLinear x n_map_r (lrelu gain 1), PixelNorm + EqualLinear x n_map_d (gain sqrt 2) */
  const float* map_r_w[CIPS3D_MAX_MAP_LAYERS]; const float* map_r_b[CIPS3D_MAX_MAP_LAYERS];
  const float* map_d_w[CIPS3D_MAX_MAP_LAYERS]; const float* map_d_b[CIPS3D_MAX_MAP_LAYERS];
  int32_t map_d_in[CIPS3D_MAX_MAP_LAYERS];
  float map_d_lr_mul;
  int32_t decoder_bf16;          /* != 0: every decoder GEMM runs in the bf16 compute mode (CIPS3D_GEMM_BF16); 2: the fused
                                    up-sampling stages also keep y_lo / y_next as bf16 (CIPS3D_Y_BF16) */
  float* lat[4];                 /* ping-pong [B, max(style_dim_r, style_dim_d, z_dim)]: [0,1] renderer chain, [2,3] decoder chain */
  float* styles_r;               /* [B, D+1, style_dim_r] */
  float* styles_d;               /* [B, n_latent, style_dim_d] */
  /* style heads */
  const cips3d_linear_desc* film_table; int32_t film_n, film_rows;     /* -> nerf.film */
  const cips3d_linear_desc* mod_table; int32_t mod_n, mod_rows;        /* -> s_buf */
  const cips3d_modulate_desc* wm_table; int32_t wm_n, wm_rows;         /* s_buf -> wm buffers */
  /* renderer: everything but the per-call camera / perturbation pointers is pre-filled */
  cips3d_nerf_params nerf;
  float* features;               /* [B, H, S, S] NeRF feature map = decoder input */
  /* decoder */
  cips3d_dec_layer layers[CIPS3D_MAX_DEC_LAYERS];
  float* act[2];                 /* activation ping-pong, each >= B * max(C*H*W) floats */
  float* y_lo;                   /* low-resolution GEMM result feeding the FIR up-sampler */
  float* y_lo2;                  /* second one (a chained stage writes the next stage's while reading its own) or NULL */
  float* skip[2];                /* RGB skip ping-pong, each >= B*3*Hout*Wout floats */
  float* rgb_part;               /* ToRGB partial-sum slots, each [B,3,H0*W0] at the input resolution (or NULL: no folding) */
  int64_t rgb_part_slots;        /* capacity in slots */
  /* range workspace (NULL: the decoder does not run split-fp16): zeroed by every forward before its first use */
  float* range_ws;               /* range_ws_words 32-bit words holding every layer's amax / aexp / lconst rows */
  int64_t range_ws_words;
  float* feat_amax;              /* [B][CIPS3D_AMAX_FLOATS] of the NeRF feature map when it takes the fp32 -> planes conversion pass */
  int32_t* feat_exp;             /* [B][blocks] exponents written by that pass */
  float* feat_pmax;              /* [B][ceil(S*S/64)][hidden/16] patch maxima written by that pass */
  float* tmp_amax;               /* [B][CIPS3D_AMAX_FLOATS] scratch for tensors no producer tracked */
  /* workspace of the one-launch style phase (cips3d_style_phase) or NULL (the phase then runs as launches): */
  void* style_xch;               /* 2 * CIPS3D_MAX_MAP_LAYERS * B * style_xch_dim 8-byte {value, tag} granules, zeroed once */
  void* style_sync;              /* 256 bytes, zeroed once: [0] generation of the last launch, [1] waits that gave up (stays 0),
                                    [3..] diagnostics: 10 ns ticks from the start of workgroup 0 to its end / to each stage */
  int32_t style_xch_dim;         /* >= every width of the two mapping networks, % 4 == 0 */
  int32_t range_volatile_words;  /* the first range_volatile_words words of range_ws hold what every forward measures anew (amax /
                                    aexp rows, the feature map's rows); the lconst rows the modulate table writes lie behind them.
                                    A forward with io.styles_resident zeroes only these (0: such forwards are refused on a ranged plan) */
} cips3d_generator_plan;

typedef struct cips3d_forward_io {
  const float* z_r;        /* [B, z_dim] or NULL when styles_r of the plan was filled by the caller */
  const float* z_d;        /* [B, z_dim] or NULL (same for styles_d) */
  const float* mean_r;     /* [style_dim_r] truncation mean or NULL (truncation = 1) */
  const float* mean_d;     /* [style_dim_d] */
  float trunc_psi;
  int32_t pad_;
  const float* cam_poses; const float* focals; const float* near_; const float* far_;   /* [B,...] */
  const float* perturb_u;  /* [B, R] or NULL */
  float* sdf;              /* [B, R, N] or NULL */
  const float* noise[CIPS3D_MAX_DEC_LAYERS];      /* per StyledConv: [1 or B][Hout*Wout] or NULL */
  int64_t noise_bstride[CIPS3D_MAX_DEC_LAYERS];   /* 0 (shared) or Hout*Wout */
  float* rgb;              /* [B, 3, Hout, Wout] final image (written by the last ToRGB); with rgb_is_u8 a uint8 image of that shape */
  float* thumb;            /* [B, 3, S, S] */
  float* xyz;              /* [B, 3, S, S] */
  float* mask;             /* [B, 2, S, S]: background weight, -|xyz| */
  void* ev_nerf_start;     /* optional hipEvent_t recorded on `stream` right before / after the render kernel */
  void* ev_nerf_stop;
  /* optional: the call makes the fresh draws itself -- exactly cips3d_rng_fill(rng_seed, rng_base, rng_normal, rng_n_normal,
   * rng_uniform, rng_n_uniform), its threads spread over the mapping networks' launches (whose few latency-bound work groups
   * leave the chip idle) or, without those, as a launch of its own; the caller points noise[] / perturb_u into the two arrays.
   * Both counts 0: no draw. */
  uint64_t rng_seed, rng_base;
  float* rng_normal; int64_t rng_n_normal;
  float* rng_uniform; int64_t rng_n_uniform;
  float noise_bound;       /* upper bound of |noise[i][...]| over the call (0: the rng draw's own bound, 5.77, is used) */
  int32_t views_in_flight; /* > 1: the caller keeps that many independent forwards in flight on different streams (a hint: the 64^2
                              chain then launches half-chip tiles, cips3d_range.half_chip; results do not depend on it) */
  /* optional timeline of the decoder (measurement only; bench.py's roofline.kernels): ev_marks[0] and ev_marks[1] are
   * recorded back to back right before the first decoder launch (their distance calibrates what a record costs) and
   * ev_marks[k + 1] right after the k-th decoder launch of the call (hipEvent_t handles, at most
   * n_ev_marks of them; further launches are not marked); ev_info[4 k .. 4 k + 3] receives what launch k was:
   * {CIPS3D_MARK_* kind, input channels, output channels, output height}.  *ev_count = marks recorded.  A record between two dependent
   * launches costs ~1-2 us of queue drain: intervals are upper bounds of the kernel times. */
  void** ev_marks; int32_t* ev_info; int32_t* ev_count; int32_t n_ev_marks;
  /* != 0: a frame of a SEQUENCE (multi-view rendering of one latent: models/render_video_web_v10.py:1792-1824 calls G with the same
   * sample_z / truncation / noise_bufs for every frame).  The plan's style tables -- styles_r / styles_d, the FiLM table, the
   * modulation table and all modulated / demodulated decoder matrices with their range constants -- are taken as left by the last
   * full forward of this plan: the mapping networks, the style heads and the modulate table are NOT run (z_r / z_d / mean_* /
   * trunc_psi are ignored); only the measured range rows are zeroed.  The reference offers the same hoist through its
   * style_render= / style_decoder= arguments (models/model_v3.py:875-914).  Results are bit-identical to a full forward with the
   * same inputs.  The caller owns the promise that styles, truncation, weights and the bound of |noise| have not changed. */
  int32_t styles_resident;
  /* != 0: `rgb` points to uint8 [B, 3, Hout, Wout] and receives cips3d_rgb_to_uint8 of the image (CIPS3D_RGB_U8 of the last
   * up-sampling stage).  Only plans whose last layers form a fused up-sampling stage can do it: CIPS3D_E_UNSUPP otherwise (the
   * caller then renders fp32 and converts). */
  int32_t rgb_is_u8;
  /* != 0: `mask` is [2, B, S, S] (cips3d_nerf_params.mask_planar); only for plans whose render launch fuses its finish */
  int32_t mask_planar;
} cips3d_forward_io;

#define CIPS3D_MARK_START 0
#define CIPS3D_MARK_PLANES_GEMM 1   /* a 1x1 StyledConv of the planes run (chain_gemm_kernel), epilogue + folded ToRGB included */
#define CIPS3D_MARK_GEMM 2          /* cips3d_modconv1x1 / _torgb (fp32 activations in HBM) */
#define CIPS3D_MARK_LOWRES_GEMM 3   /* the low-resolution GEMM of an up-sampling StyledConv as a launch of its own */
#define CIPS3D_MARK_FUSED_STAGE 4   /* cips3d_fused_up_conv_next: FIR + act -> conv2 + act -> ToRGB + skip [-> next low-res GEMM] */
#define CIPS3D_MARK_TORGB 5         /* cips3d_torgb / cips3d_torgb_reduce */
#define CIPS3D_MARK_OTHER 6

int cips3d_generator_forward(const cips3d_generator_plan* plan, const cips3d_forward_io* io, void* stream);

/* The part of cips3d_generator_forward in front of the modulated weights: Generator.mapping_networks (models/model_v3.py:
 * 1299-1418) -> plan.styles_r / styles_d, the FiLM heads (cips3d/volume_renderer.py:66-67) and every
 * ModulatedConv2d.modulation (models/model_v3.py:254,268) through the plan's tables, the call's fresh draws (io.rng_*) and the
 * zeroing of the range workspace.  mode 0: a chain of dependent launches (two layers or a layer and a table per launch);
 * mode 1: ONE launch of resident workgroups that hand each layer's outputs to each other through plan.style_xch
 * (CIPS3D_E_UNSUPP when the plan has no such workspace, a chain is absent, B > 8 or a width exceeds style_xch_dim);
 * mode -1: what cips3d_generator_forward does -- mode 0, or mode 1 (where supported) when the environment says
 * CIPS3D_STYLE_PHASE=1: the one launch measured slower on MI355X.  Both modes give bit-identical
 * results. */
int cips3d_style_phase(const cips3d_generator_plan* plan, const cips3d_forward_io* io, int mode, void* stream);
/* sizeof() of the two structs as the library sees them (layout check for foreign-language bindings) */
int64_t cips3d_sizeof_plan(void);
int64_t cips3d_sizeof_io(void);
/* sizeof() of every struct that crosses the boundary (a binding checks its own layout against these at load time):
 * which = 0 cips3d_generator_plan, 1 cips3d_forward_io, 2 cips3d_nerf_params, 3 cips3d_linear_desc,
 * 4 cips3d_modulate_desc, 5 cips3d_dec_layer, 6 cips3d_nerf_bwd_geom, 7 cips3d_nerf_bwd_fused_params;
 * -1 for an unknown index */
int64_t cips3d_sizeof_struct(int which);

/* ------------------------------------------------------------------ stand-alone renderer steps
 * The fused kernel does these in registers; as separate entry points they give every method of the reference's `Render`
 * class (cips3d/nerf_utils.py:11-338) a callable counterpart with the reference's tensor layouts. */
/* Render.get_rays_in_world (:18-66): rays_o / rays_d / viewdirs [B,S,S,3] */
int cips3d_rays_in_world(const float* cam_poses, const float* focals, int img_size, int static_viewdirs, int B,
                         float* rays_o, float* rays_d, float* viewdirs, void* stream);
/* Render.get_z_vals, offset-sampling branch (:69-121): z [B,R,N]; perturb_u [B,R] per-ray uniform or NULL */
int cips3d_z_vals(const float* near_, const float* far_, const float* perturb_u, int B, int R, int N, float* z, void* stream);
/* the classic stratified branch of Render.get_z_vals (offset_sampling = False, cips3d/nerf_utils.py:98-117; `mlp_init_pass`):
 * t = linspace(0, 1, N); perturb_t [B,R,N] (one uniform per SAMPLE, drawn inside the interval between the neighbouring
 * midpoints) or NULL (perturb = False) */
int cips3d_z_vals_stratified(const float* near_, const float* far_, const float* perturb_t, int B, int R, int N, float* z,
                             void* stream);
/* Render.get_points (+ Render.normalize_points when pts_normalized != NULL) (:124-170): [B,R,N,3]; either output may be NULL */
int cips3d_ray_points(const float* rays_o, const float* rays_d, const float* z, const float* near_, const float* far_, int B,
                      int R, int N, float* pts, float* pts_normalized, void* stream);
/* Render.volume_integration (:231-338): rgb [n_rays,N,3], sdf [n_rays,N], features [n_rays,N,C] or NULL,
 * z_vals [n_rays,N], rays_d [n_rays,3], pts [n_rays,N,3] -> rgb_map [n_rays,3], feature_map [n_rays,C], xyz [n_rays,3],
 * mask [n_rays,2] = (last weight, -|xyz|).  N <= 256, C % 4 == 0.
 * flags: 0 = the with_sdf branch (density = sigmoid(-sdf / beta) / beta, :276-286);
 *   CIPS3D_VI_RAW_DENSITY      `sdf` holds the network's raw density output, density = softplus(raw) (:288-297; the caller adds
 *                              raw_noise_std * randn beforehand); sigmoid_beta may be NULL
 *   CIPS3D_VI_FORCE_BACKGROUND the last sample takes the weight the others leave: w[N-1] = 1 - sum_{k<N-1} w[k] (:309-310) */
#define CIPS3D_VI_RAW_DENSITY 1
#define CIPS3D_VI_FORCE_BACKGROUND 2
int cips3d_volume_integration(const float* rgb, const float* sdf, const float* features, const float* z_vals,
                              const float* rays_d, const float* pts, const float* sigmoid_beta, int64_t n_rays, int N, int C,
                              int flags, float* rgb_map, float* feature_map, float* xyz, float* mask, void* stream);

/* Dense layer over a point-major tensor x [n_points, in] -> y [n_points, out] for the per-point module forwards of the
 * reference when they are called directly (LinearLayer / FiLMSiren, cips3d/volume_renderer.py:15-85):
 *   mode 0: y = out_scale * (x W^T + bias) + out_shift
 *   mode 1: y = sin(gamma * (x W^T + bias) + beta), film = [B][2][out] (gamma, beta), batch = point / points_per_batch */
int cips3d_points_linear(const float* x, const float* W, const float* bias, const float* film, int64_t n_points,
                         int64_t points_per_batch, int in_dim, int out_dim, int mode, float out_scale, float out_shift,
                         float* y, void* stream);

/* ------------------------------------------------------------------ backward of the path
 * The flip-inversion loop (reference models/projector_v10.py:211-277,1058-1216) calls loss.backward() through
 * Generator.forward; the reference gets these gradients from PyTorch autograd over cuBLAS/cuDNN and from the backward
 * wiring of its two native ops (op/fused_act.py:20-84, op/upfirdn2d.py:20-143).  Each entry below is the explicit
 * backward of one forward entry point; torch.autograd.Function objects on the host chain them.
 * Gradient outputs are OVERWRITTEN (not accumulated); NULL output pointers skip that gradient. */

/* Backward of cips3d_linear (pixelnorm = 0, no truncation):  y = act(x W^T w_scale + bias b_scale) * out_scale + shift,
 * act = leaky-ReLU(0.2) * act_gain when lrelu (then `out` = the forward output, whose sign selects the slope).
 * dx [B,in], dW [out,in], dbias [out].  (models/model_v3.py:40-65,183-210; cips3d/volume_renderer.py:15-35) */
int cips3d_linear_bwd(const float* x, int64_t x_stride, const float* W, const float* out, int64_t out_stride,
                      const float* dout, int64_t dout_stride, int B, int in_dim, int out_dim, float w_scale,
                      float b_scale, int lrelu, float act_gain, float out_scale, float* dx, int64_t dx_stride,
                      float* dW, float* dbias, void* stream);

/* Backward of cips3d_modulate_weights (models/model_v3.py:267-278).  dwm [B,Cout,Cin*ksq] holds dL/dwm on entry and is
 * used as scratch (it holds dL/du on return, u = the modulated weight before demodulation).
 * dW [Cout,Cin*ksq] (summed over the batch), ds [B,Cin] with row stride ds_stride. */
int cips3d_modulate_bwd(float* dwm, const float* W, const float* s, int64_t s_stride, int B, int Cout, int Cin, int ksq,
                        float scale, int demodulate, float* dW, float* ds, int64_t ds_stride, void* stream);

/* wm [B,M,K] plain -> the MFMA A-fragment order cips3d_modconv1x1 consumes; transpose bit 0 packs wm[b]^T (K x M),
 * which turns cips3d_modconv1x1 into the data-gradient GEMM dx = wm^T dy.  M, K multiples of 16.  transpose bit 1: split-fp16
 * fragments (2^8 w = fp16 hi + lo) for the CIPS3D_GEMM_SPLIT mode; the packed matrix's K a multiple of 32, |w| < 255. */
int cips3d_pack_weights(const float* wm, float* packed, int B, int M, int K, int transpose, void* stream);

/* Weight gradient of the 1x1 convolution: dwm[b][m][k] = sum_p dy[b][m][p] * x[b][k][p]  (dy [B,M,P], x [B,K,P]).
 * M, K multiples of 32, P multiple of 4, 16-byte aligned pointers.  Pixel chunks are summed with fp32 atomics, so the
 * last bits depend on the execution order (as cuDNN's default weight-gradient algorithms do). */
int cips3d_gemm_wgrad(const float* dy, const float* x, float* dwm, int B, int M, int K, int64_t P, void* stream);
/* ---- one-call decoder backward: the data-gradient GEMM with the previous layer's activation backward as its epilogue.
 * The reference gets all of this from autograd over `StyledConv.forward` / `ToRGB.forward` (models/model_v3.py:444-482).
 *
 * g [B,Cin,HW] = gradient w.r.t. the pre-activation of StyledConv l (at the resolution of its GEMM), wm_t = its modulated
 * weights packed TRANSPOSED (cips3d_modulate_table with CIPS3D_MOD_TRANSPOSE, + CIPS3D_MOD_SPLIT16 for flags =
 * CIPS3D_GEMM_SPLIT): the product wm^T g [B,Cout,HW] is the gradient w.r.t. layer l's input = the output y of StyledConv
 * l-1 (Cout here = that layer's channel count).  The epilogue makes it the gradient w.r.t. layer l-1's pre-activation:
 *   dpre = (wm^T g + sum_c rgb_w[b][c][o] drgb[b][c][p]) * (y > 0 ? sqrt2 : 0.2 sqrt2)
 * and adds, with one fp32 atomic per row and workgroup,
 *   d_bias[o]      += sum_{b,p} dpre                     (FusedLeakyReLU bias, fused_act.py:20-84)
 *   d_noise_w[o]   += sum_{b,p} dpre * noise[b][p]       (per-channel partials of NoiseInjection.weight's gradient; the
 *                                                         caller sums the Cout partials)
 *   d_rgb_w[b][c][o] += sum_p drgb[b][c][p] * y[b][o][p] (the modulated ToRGB weights' gradient)
 * The accumulators must be zeroed by the caller.  rg->x_amax scales the split of g (CIPS3D_GEMM_SPLIT), rg->out_amax
 * records max|dpre| per sample (the next GEMM's and the weight-gradient GEMM's operand scale). */
typedef struct cips3d_actbwd {
  const float* y;        /* [B,Cout,HW] stored output of layer l-1 (only its sign is used) */
  const float* rgb_w;    /* [B,3,Cout] plain modulated weights of the ToRGB that read y, or NULL (both or neither with drgb) */
  const float* drgb;     /* [B,3,HW] gradient w.r.t. that ToRGB's output */
  float* d_bias;         /* [Cout] or NULL */
  float* d_noise_w;      /* [Cout] or NULL (needs `noise`) */
  float* d_rgb_w;        /* [B,3,Cout] or NULL */
  /* Slotted accumulators.  The L2 executes atomics per cache line and instruction: at 256^2 a row's accumulator line would
   * take one add from each of ~2000 workgroups, ~50 ns apiece (measured: 108 us for a 40 us GEMM).  With slots = S (a power
   * of two, 0 / 1 = none) workgroup w adds into copy w % S: d_bias / d_noise_w copy s at + s * slot_stride floats, d_rgb_w
   * copy s at + s * rgb_slot_stride floats; the caller sums the copies (cips3d_slot_reduce). */
  int32_t slots, slot_stride, rgb_slot_stride, pad_;
} cips3d_actbwd;
int cips3d_modconv1x1_actbwd(const float* g, const float* wm_t, float* dpre, int B, int Cin, int Cout, int64_t HW, int flags,
                             const cips3d_actbwd* ab, const float* noise, int64_t noise_bstride, const cips3d_range* rg,
                             void* stream);

/* Activation backward without a GEMM in front of it (the LAST StyledConv: its output is read by the last ToRGB only):
 * dpre = (g_in + rgb_w^T drgb) * lrelu'(y), same reductions as cips3d_actbwd; g_in may be NULL.  out_amax: amax row or NULL. */
int cips3d_act_tail_bwd(const float* g_in, const float* y, const float* rgb_w, const float* drgb, const float* noise,
                        int64_t noise_bstride, float* dpre, float* d_bias, float* d_noise_w, float* d_rgb_w, float* out_amax,
                        int B, int C, int64_t HW, int slots, int slot_stride, int rgb_slot_stride, void* stream);
/* dst[i] = sum_{s < S} src[s * stride + i] for a table of jobs in one launch (dst may be src: copy 0 then holds the sum) */
typedef struct cips3d_slot_job {
  const float* src; float* dst;
  int32_t n, slots, stride;
  int32_t row_begin;     /* exclusive prefix sum of ceil(n / 256) (workgroups) over the table */
} cips3d_slot_job;
int cips3d_slot_reduce(const cips3d_slot_job* table_dev, int n_jobs, int total_blocks, void* stream);
/* Transpose of the 2x FIR up-sampler (upfirdn2d up = 2, pad = (2, 1), 4 x 4 taps; op/upfirdn2d.py:20-143 reaches it through
 * upfirdn2d with swapped factors): g_hi [B,C,2H,2W] -> g_lo [B,C,H,W]; out_amax: per-sample amax rows of g_lo or NULL. */
int cips3d_up2_fir_bwd(const float* g_hi, const float* fir, float* g_lo, float* out_amax, int B, int C, int H, int W,
                       void* stream);
/* Modulation backward (models/model_v3.py:267-278, k = 1) of a table of layers in ONE launch: cips3d_modulate_bwd's
 * arithmetic; dW is stored, ds is ADDED to (the caller zeroes it).  Cin <= 512, B <= 4, n_desc <= 64. */
typedef struct cips3d_modbwd_desc {
  const float* d_wm;     /* [B, Cout, Cin] gradient of the modulated weights */
  const float* W;        /* [Cout, Cin] */
  const float* s;        /* row b at s + b * s_stride */
  float* dW;             /* [Cout, Cin] or NULL */
  float* ds;             /* row b at ds + b * ds_stride, or NULL */
  int64_t s_stride, ds_stride;
  int32_t Cout, Cin;
  float scale;
  int32_t demodulate;
  int32_t row_begin;     /* exclusive prefix sum of ceil(Cout / 32) (workgroups) over the table */
  int32_t pad_;
} cips3d_modbwd_desc;
int cips3d_modulate_table_bwd(const cips3d_modbwd_desc* table_dev, int n_desc, int total_blocks, int B, void* stream);

/* Adam step over a list of parameter tensors in one launch per 48 tensors (csrc/optim.hip): the update rule of
 * torch.optim.Adam (amsgrad off, no weight decay) that the reference's inversion loop runs three of per step
 * (models/projector_v10.py:279-390, 1210-1216).  `entries` is a HOST array (copied into the kernel arguments); p / m / v are
 * updated in place; step = the 1-based step count of these tensors (bias corrections). */
typedef struct cips3d_adam_entry {
  float* p; const float* g; float* m; float* v;
  int64_t n;
} cips3d_adam_entry;
int cips3d_adam_step(const cips3d_adam_entry* entries, int n_entries, float lr, float beta1, float beta2, float eps, int step,
                     void* stream);
/* The same update for the tensors of SEVERAL parameter groups / optimisers in shared launches (48 tensors and up to 8 distinct
 * hyper-parameter sets per launch): entry i takes hypers[group_of[i]].  The three optimisers of an inversion step (camera,
 * NeRF W+, decoder W+ + decoder parameters: 125 tensors in 4 groups) are then 3 launches instead of 6. */
typedef struct cips3d_adam_hyper { float lr, beta1, beta2, eps; int32_t step; } cips3d_adam_hyper;
int cips3d_adam_step_groups(const cips3d_adam_entry* entries, const int* group_of, int n_entries, const cips3d_adam_hyper* hypers,
                            int n_hypers, void* stream);

/* The two squared-difference terms of the inversion loss (models/projector_v10.py:1173-1174:
 * `(target - synth).square().sum() * rgb_weight + (target_thumb - synth_thumb).square().sum() * thumb_weight`; with
 * c_k = weight_k / n_k the F.mse_loss form of :1178):  loss[0] = c0 sum (a0 - b0)^2 + c1 sum (a1 - b1)^2, deterministic
 * (per-workgroup partial sums, then their total in a fixed order: two launches).  partial: scratch of
 * 2 * cips3d_sqdiff_pair_partials(n0, n1) floats.  _bwd: d_k = gloss[0] * 2 c_k (a_k - b_k), one launch; gloss = the
 * loss' incoming gradient, a device scalar.  Either tensor may be empty (n = 0). */
int cips3d_sqdiff_pair_partials(int64_t n0, int64_t n1);
int cips3d_sqdiff_pair(const float* a0, const float* b0, int64_t n0, float c0, const float* a1, const float* b1, int64_t n1,
                       float c1, float* partial, float* loss, void* stream);
int cips3d_sqdiff_pair_bwd(const float* a0, const float* b0, int64_t n0, float c0, float* d0, const float* a1, const float* b1,
                           int64_t n1, float c1, float* d1, const float* gloss, void* stream);

/* The decoder as ONE differentiable node (csrc/decoder_grad.hip): Decoder.forward with every StyledConv output kept, and the
 * whole backward -- gradients of the features, the W+ styles and every decoder parameter -- in one call each.
 * models/model_v3.py:592-637 under `loss.backward()` (models/projector_v10.py:1203-1209).  kernel_size 1. */
#define CIPS3D_GRAD_MAX_LAYERS 48
typedef struct cips3d_grad_layer {
  int32_t kind;            /* 0 StyledConv, 1 StyledConv with up-sampling, 2 ToRGB, 3 ToRGB whose skip is up-sampled first */
  int32_t Cin, Cout;
  int32_t H, W;            /* resolution of the layer's input */
  int32_t noise_index;     /* StyledConv: entry of io->noise; -1 otherwise */
  int32_t flags;           /* bit 0: split-fp16 GEMMs (wm / wm_t packed with CIPS3D_MOD_SPLIT16) */
  int32_t pad_;
  const float* bias;       /* FusedLeakyReLU bias [Cout] / ToRGB bias [3] */
  const float* noise_w;    /* [1] (StyledConv) */
  const float* fir;        /* [4,4] (kinds 1, 3) */
  const float* wm;         /* StyledConv: packed modulated weights; ToRGB: plain [B,3,Cin] (both written by mod_table) */
  const float* wm_t;       /* StyledConv: the packed transpose */
  float* y;                /* StyledConv: the stored output [B,Cout,Ho,Wo] */
  float* y_amax;           /* amax rows of y      (inside plan->amax_base .. + amax_bytes) */
  float* g_amax;           /* amax rows of the gradient w.r.t. this layer's pre-activation (same region) */
  float* glo_amax;         /* kind 1: amax rows of that gradient after the FIR transpose (same region) */
  float* d_wm;             /* [B,Cout,Cin] gradient of the modulated weights      (inside plan->zero_base .. + zero_bytes) */
  float* d_bias;           /* [Cout] / [3]                                         (same region) */
  float* d_nw_part;        /* StyledConv: row of plan->nw_parts                    (same region) */
  /* slotted accumulators (cips3d_actbwd): copies of d_bias / d_nw_part at + s * slot_stride, of d_wm (ToRGB: the
   * accumulators of the epilogue that reads this ToRGB's input) at + s * rgb_slot_stride; slots <= 1: none */
  int32_t slots, slot_stride, rgb_slot_stride, pad2_;
} cips3d_grad_layer;

typedef struct cips3d_decoder_grad_plan {
  int32_t B, n_layers, style_dim, pad_;
  const cips3d_linear_desc* style_table;      /* style -> s of every layer (cips3d_linear_table) */
  int32_t style_n, style_rows;
  const cips3d_modulate_desc* mod_table;      /* s -> wm in every form the layers name */
  int32_t mod_n, mod_rows;
  const cips3d_modbwd_desc* modbwd_table;
  int32_t modbwd_n, modbwd_blocks;
  const int64_t* style_w_offsets;             /* cips3d_linear_table_bwd's w_offsets */
  const float* styles;                        /* [B, n_latent, style_dim]: the style table's input (filled by the caller) */
  const float* s_all;                         /* the style table's output base */
  const float* ds_all;                        /* its gradient, same layout (zero region; the modbwd table adds to it) */
  float* d_styles;                            /* [B, n_latent, style_dim] or NULL (zero region) */
  float* d_style_W; float* d_style_b;         /* flat gradients of the style heads or NULL */
  float* amax_base; int64_t amax_bytes;       /* every amax row of the step: zeroed by the forward call */
  float* zero_base; int64_t zero_bytes;       /* every accumulator of the backward: zeroed by the backward call */
  float* feat_amax;                           /* amax rows of the features (amax region) */
  float* y_lo;                                /* low-resolution GEMM result of an up-sampling layer (largest one) */
  float* g[2];                                /* two gradient buffers of the largest activation */
  float* g_lo;                                /* FIR-transposed gradient of an up-sampling layer (largest one) */
  float* drgb_lo[4];                          /* gradient images of the skip chain below the output resolution */
  float* rgb[2];                              /* intermediate rgb images */
  const float* nw_parts; int32_t nw_stride, pad2_;    /* [n StyledConvs][nw_stride] noise-weight partials (all slots of a row) */
  const cips3d_slot_job* slot_table;          /* sums the slot copies of d_bias / ToRGB d_wm before they are used, or NULL */
  int32_t slot_n, slot_blocks;
  float* d_noise_w;                           /* [n StyledConvs] or NULL */
  cips3d_grad_layer layers[CIPS3D_GRAD_MAX_LAYERS];
} cips3d_decoder_grad_plan;

typedef struct cips3d_decoder_grad_io {
  const float* features;     /* [B, Cin0, H0, W0] */
  const float* noise[CIPS3D_GRAD_MAX_LAYERS];
  int64_t noise_bstride[CIPS3D_GRAD_MAX_LAYERS];
  float* rgb;                /* forward: [B,3,Hf,Wf] */
  const float* d_rgb;        /* backward: gradient of rgb */
  float* d_features;         /* backward: [B, Cin0, H0, W0] or NULL */
  /* optional (both or neither): the plan's style table re-pointed at the caller's own contiguous W+ tensor `styles`
   * [B, n_latent, style_dim] -- forward and backward then read the styles where they are instead of from the plan's copy */
  const cips3d_linear_desc* style_table;
  const float* styles;
} cips3d_decoder_grad_io;
int cips3d_decoder_grad_forward(const cips3d_decoder_grad_plan* plan, const cips3d_decoder_grad_io* io, void* stream);
int cips3d_decoder_grad_backward(const cips3d_decoder_grad_plan* plan, const cips3d_decoder_grad_io* io, void* stream);
int cips3d_sizeof_grad_plan(void);
int cips3d_sizeof_grad_io(void);

/* The same contraction on split-fp16 products (three fp16 MFMA products per fp32 product).  dy_amax / x_amax: the measured
 * per-sample maxima of the operands ([B][CIPS3D_AMAX_FLOATS] slot arrays as cips3d_range uses them, or NULL: unscaled split,
 * for tests on O(1) data); each operand is split as v * 2^-e with max|v| 2^-e in [2^14, 2^15) and the sums are scaled back
 * exactly, so gradients of any magnitude keep fp32-level accuracy.  accumulate != 0: dwm is added to (the caller zeroed it).
 * M % 32 == K % 32 == P % 32 == 0, 16-byte aligned operands, else CIPS3D_E_UNSUPP. */
int cips3d_gemm_wgrad_split(const float* dy, const float* x, float* dwm, int B, int M, int K, int64_t P, const float* dy_amax,
                            const float* x_amax, int accumulate, void* stream);

/* Backward of cips3d_noise_bias_act / the epilogue of StyledConv (models/model_v3.py:327-341; op/fused_act.py:20-84):
 * y = lrelu(x + noise_w*noise + bias_c)*sqrt2.  dx [B,C,HW] (may alias dy), dnoise [1 or B][HW] (layout of `noise`),
 * dnoise_w [1], dbias [C]; scratch_c [C] is required with dnoise_w (per-channel partial sums). */
int cips3d_noise_bias_act_bwd(const float* dy, const float* y, const float* noise, int64_t noise_bstride,
                              const float* noise_w, float* dx, float* dnoise, float* dnoise_w, float* dbias,
                              float* scratch_c, int B, int C, int64_t HW, void* stream);

/* Backward of cips3d_torgb without the skip (its gradient is drgb itself; the FIR up-sampling of the skip is
 * cips3d_upfirdn2d): dx [B,C,HW], dwm [B,3,C], dbias [3].  (models/model_v3.py:469-482) */
int cips3d_torgb_bwd(const float* drgb, const float* x, const float* wm, float* dx, float* dwm, float* dbias, int B, int C,
                     int64_t HW, void* stream);

/* ---- NeRF half, materialised backward (csrc/nerf_bwd.hip).  Replaces autograd through Render.prepare_nerf_inputs,
 * SirenGenerator.forward and Render.volume_integration (cips3d/nerf_utils.py:18-338, cips3d/volume_renderer.py:39-160).
 * Activations are recomputed layer by layer in the decoder's layout act[b][c][p], p = sample * R + ray (R = img_size^2,
 * P = R * n_samples), so the hidden-layer GEMMs and their data gradients are cips3d_modconv1x1 (weights packed with
 * cips3d_pack_weights, transposed for the gradient).  Sequence: cips_3dplusplus_amd/autograd.py:NerfRenderFn.backward. */
typedef struct cips3d_nerf_bwd_geom {
  const float* cam_poses;  /* [B,3,4] */
  const float* focals;     /* [B] */
  const float* near_;      /* [B] */
  const float* far_;       /* [B] */
  const float* perturb_u;  /* [B,R] or NULL */
  int32_t B, img_size, n_samples, static_viewdirs;
} cips3d_nerf_bwd_geom;

/* rays -> ptsn [B,3,P] (normalised points), pre0 / h0 [B,H,P] (layer-0 pre-activation incl. bias, and its FiLM sine),
 * viewdirs [B,3,R].  film points at layer 0 of the [B,L,2,H] table, film_bstride = L*2*H floats. */
int cips3d_nerf_bwd_points(const cips3d_nerf_bwd_geom* geom, const float* w_first, const float* bias0, const float* film,
                           int film_bstride, int H, float* ptsn, float* pre0, float* h0, float* viewdirs, void* stream);
/* acc [B,H,P] (GEMM output) -> pre = acc + bias_c (+ wd[c*wd_stride + 0..2] . viewdir[ray]) in place; h = sin(gamma pre + beta) */
int cips3d_nerf_bwd_film(float* acc, float* h, const float* bias, const float* film, int film_bstride, const float* wd,
                         int wd_stride, const float* viewdirs, int B, int H, int R, int64_t P, void* stream);
/* out[b][r][p] = sum_c Wm[r*row_stride + c*col_stride] * x[b][c][p] + bias[r], n_rows <= 4 (bias may be NULL) */
int cips3d_nerf_bwd_heads(const float* x, const float* Wm, int row_stride, int col_stride, const float* bias, int n_rows,
                          int B, int H, int64_t P, float* out, void* stream);
/* g[b][p] = sum_c dF[b][c][ray(p)] * f[b][c][p] */
int cips3d_nerf_bwd_dot(const float* dF, const float* f, int B, int H, int R, int64_t P, float* g, void* stream);
/* volume integration forward + backward per ray: sdf [B,P], crgb [B,3,P] (rgb logits), g [B,P], dthumb [B,3,R]
 * -> w [B,P] (compositing weights), dsdf [B,P], dcrgb [B,3,P], ddnorm [B,R] (d loss / d |rays_d|, which scales the
 * sample spacing); T_scratch [B,P].  sigmoid_beta == NULL: with_sdf = False -- `sdf` is the raw density, sigma = softplus(sdf)
 * (cips3d/nerf_utils.py:288-297); no dbeta_ray then. */
int cips3d_nerf_bwd_composite(const cips3d_nerf_bwd_geom* geom, const float* sdf, const float* crgb, const float* g,
                              const float* dthumb, const float* sigmoid_beta, float* w, float* T_scratch, float* dsdf,
                              float* dcrgb, float* ddnorm, float* dbeta_ray, void* stream);
/* (dbeta_ray [B,R] or NULL: per-ray d loss / d sigmoid_beta, for `optim_render_params`, models/projector_v10.py:848-872)
 * out[row][c] += sum_{b,p} a[b][row][p] * x[b][c][p mod Px] (c < nx <= 3), out[row][3] += sum_{b,p} a[b][row][p]; a [B,rows,P],
 * x [B,nx,Px], out [rows,4] zeroed by the caller: the narrow weight gradients and the biases of the point MLP. */
int cips3d_nerf_bwd_row_dots(const float* a, const float* x, int nx, int64_t Px, float* out, int B, int rows, int64_t P,
                             void* stream);
/* FiLM-sine backward, in place on buf [B,H,P]:
 *   mode 0: buf = dh (+ w_sigma[c]*dsdf[p] when w_sigma != NULL)            -> buf = d(pre)
 *   mode 1: upstream = weights[p]*dF[c][ray] + sum_r w_rgb[r][c]*dcrgb[r][p] -> buf = d(pre)   (buf's old content unused)
 * dfilm (same layout / stride as film, pointing at this layer) += sums for d(gamma), d(beta); zero it before the first call. */
int cips3d_nerf_bwd_film_grad(float* buf, const float* pre, const float* film, int film_bstride, int mode,
                              const float* w_sigma, const float* dsdf, const float* weights, const float* dF,
                              const float* w_rgb, const float* dcrgb, float* dfilm, int B, int H, int R, int64_t P,
                              void* stream);
/* dptsn [B,3,P], dvd_pt [B,3,P] (d loss / d viewdir per point), ddnorm [B,R] -> dcam [B,3,4] */
int cips3d_nerf_bwd_camera(const cips3d_nerf_bwd_geom* geom, const float* dptsn, const float* dvd_pt, const float* ddnorm,
                           float* dcam, void* stream);
/* the same, ADDED to a dcam the caller has zeroed (cips3d_nerf_bwd_fused clears it in its preparation launch) */
int cips3d_nerf_bwd_camera_acc(const cips3d_nerf_bwd_geom* geom, const float* dptsn, const float* dvd_pt, const float* ddnorm,
                               float* dcam, void* stream);
/* ---- NeRF half, fused backward (csrc/nerf_bwd_fused.hip): the same gradients as the sequence above -- d loss / d film
 * [B,L,2,H] and d loss / d cam_poses [B,3,4] from d_features [B,H,R] and d_thumb [B,3,R] -- with the point MLP kept in the
 * register file in both directions (the reference gets them from autograd through cips3d/volume_renderer.py:39-160 and
 * cips3d/nerf_utils.py:264-338).  Two MFMA kernels with the task shape of cips3d_nerf_render (16 rays x a chunk of samples
 * per wave, weights streamed through the LDS ring, split-fp16 products):
 *   recompute: the forward MLP again, writing per MFMA layer the fp32 accumulators (`stash`, in the wave's own register
 *              order: 16 * H * 4 bytes per layer and 16-point tile, fully coalesced) and per point sdf, rgb logits and
 *              <d_features[:, ray], feature>;
 *   (cips3d_nerf_bwd_composite: volume integration forward + backward per ray, as in the materialised sequence)
 *   backward : per 16-point tile the layers in reverse -- upstream * cos(arg) from the stash, the d(gamma) / d(beta) sums
 *              (cross-lane reduce-scatter + LDS accumulation, one global atomic per workgroup and table entry), the data
 *              gradient W^T d(pre) on the matrix cores with the transposed packed stream (the gradient operand is scaled per
 *              point by a power of two into fp16's range before it is split, and the result unscaled exactly), the
 *              view-direction and point gradients on the VALU;
 *   (cips3d_nerf_bwd_camera: d points / d viewdirs -> d cam_poses)
 * HBM traffic per point and MFMA layer: one write + one read of H floats (the materialised sequence moves ~10). */
typedef struct cips3d_nerf_bwd_fused_params {
  cips3d_nerf_bwd_geom geom;
  const float* w_first;      /* [H,3] */
  const float* packed;       /* cips3d_nerf_pack_weights */
  const float* packed_t;     /* cips3d_nerf_pack_weights_t */
  const float* w_view;       /* [H,H+3] */
  const float* film;         /* [B,L,2,H], L = depth + 1 */
  const float* layer_bias;   /* [L,H] */
  const float* w_sigma;      /* [H] */
  const float* b_sigma;      /* [1] */
  const float* w_rgb;        /* [3,H] */
  const float* b_rgb;        /* [3] */
  const float* sigmoid_beta; /* [1] */
  const float* d_features;   /* [B,H,R] */
  const float* d_thumb;      /* [B,3,R] */
  float* stash;              /* cips3d_nerf_bwd_fused_stash_floats() floats */
  float* scratch;            /* cips3d_nerf_bwd_fused_scratch_floats() floats */
  float* dfilm;              /* out [B,L,2,H] */
  float* dcam;               /* out [B,3,4] */
  int32_t hidden, depth, n_chunks, pad_;
  /* NULL: the call recomputes the forward (nerf_stash_kernel fills `stash`).  Both set: `stash` was filled by
   * cips3d_nerf_render (cips3d_nerf_params.stash / bwd_sdf / bwd_crgb, same n_chunks) and these are its per-point outputs;
   * only g = <d_features, feature> is then rebuilt, from the view layer's stash. */
  const float* fwd_sdf;      /* [B,P] */
  const float* fwd_crgb;     /* [B,3,P] */
} cips3d_nerf_bwd_fused_params;

/* 1 when the fused kernels cover the shape (hidden 32/64/128/256, R % 16 == 0, tables within the LDS) */
int cips3d_nerf_bwd_fused_supported(int hidden, int depth, int img_size, int n_samples);
int64_t cips3d_nerf_bwd_fused_stash_floats(int B, int img_size, int n_samples, int hidden, int depth, int n_chunks);
int64_t cips3d_nerf_bwd_fused_scratch_floats(int B, int img_size, int n_samples, int hidden, int depth);
/* the transposed weight stream of the backward kernel, in its consumption order (view layer first, then hidden layers
 * depth-1 .. 1), each matrix W_l^T packed like cips3d_nerf_pack_weights packs W_l and with W_l's power-of-two scale (read from
 * `packed`); packed_t holds cips3d_nerf_packed_floats(hidden, depth) floats */
int cips3d_nerf_pack_weights_t(const float* w_hidden, const float* w_view, const float* packed, float* packed_t, int hidden,
                               int depth, void* stream);
int cips3d_nerf_bwd_fused(const cips3d_nerf_bwd_fused_params* p, void* stream);

/* Backward of cips3d_camera_params w.r.t. locations (azim, elev): dextrinsics [B,3,4] -> dlocations [B,2]
 * (cips3d/nerf_utils.py:344-436,466-564; focal / near / far do not depend on the angles). */
int cips3d_camera_params_bwd(const float* locations, const float* up, const float* dextrinsics, int B, float* dlocations,
                             void* stream);

/* Fresh noise of one forward in one launch: n_normal N(0,1) values (every NoiseInjection map, models/model_v3.py:334-336
 * `image.new_empty(batch, 1, height, width).normal_()`) and n_uniform U[0,1) values (the per-ray jitter, cips3d/nerf_utils.py:110
 * `torch.rand(B, h, w, 1)`).  The reference takes both from torch's global generator and pins no stream; this draws them from
 * Philox4x32-10 with key = seed and counter = (base + thread, 'CIPS', 0) -- csrc/rng.hip, restated in oracle/rng.py.  A call
 * consumes cips3d_rng_fill_threads(n_normal, n_uniform) counter values from `base` on; the caller advances its offset by that
 * (the Python binding keeps it in torch's generator state: seed = initial_seed(), base = get_offset()).  Either count may be 0.
 * cips3d_rng_words: the raw 4 x 32-bit words of threads 0 .. n_threads-1 (tests). */
int cips3d_rng_fill(uint64_t seed, uint64_t base, float* normal, int64_t n_normal, float* uniform, int64_t n_uniform,
                    void* stream);
int64_t cips3d_rng_fill_threads(int64_t n_normal, int64_t n_uniform);
int cips3d_rng_words(uint64_t seed, uint64_t base, uint32_t* out, int64_t n_threads, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CIPS3D_HIP_H */

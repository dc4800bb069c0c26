// cips3d_generator_forward: the whole generator forward (reference models/model_v3.py:875-1042) enqueued
// by ONE host call.  At batch 1 a 1024^2 view is ~60 kernels of 5-300 us; issuing them from Python costs
// ~20 us each (more than the GPU time of most of them), issuing them from here costs a hipLaunchKernel
// each, and the call can be captured into a hipGraph by the caller because nothing here allocates,
// synchronises or touches anything but `stream`.
#include <stdlib.h>

#include "common.h"

#define TRY(expr)            \
  do {                       \
    const int rc_ = (expr);  \
    if (rc_ != 0) return rc_; \
  } while (0)

extern "C" int64_t cips3d_sizeof_plan(void) { return (int64_t)sizeof(cips3d_generator_plan); }
extern "C" int64_t cips3d_sizeof_io(void) { return (int64_t)sizeof(cips3d_forward_io); }
extern "C" int64_t cips3d_sizeof_struct(int which) {
  switch (which) {
    case 0: return (int64_t)sizeof(cips3d_generator_plan);
    case 1: return (int64_t)sizeof(cips3d_forward_io);
    case 2: return (int64_t)sizeof(cips3d_nerf_params);
    case 3: return (int64_t)sizeof(cips3d_linear_desc);
    case 4: return (int64_t)sizeof(cips3d_modulate_desc);
    case 5: return (int64_t)sizeof(cips3d_dec_layer);
    case 6: return (int64_t)sizeof(cips3d_nerf_bwd_geom);
    case 7: return (int64_t)sizeof(cips3d_nerf_bwd_fused_params);
    case 8: return (int64_t)sizeof(cips3d_range);
    case 9: return (int64_t)sizeof(cips3d_reduce_job);
    default: return -1;
  }
}

int cips3d_style_phase_one_launch(const cips3d_linear_args* ar, int nr, const cips3d_linear_args* ad, int nd,
                                  const cips3d_linear_desc* film, int film_n, int film_rows, const cips3d_linear_desc* mod,
                                  int mod_n, int mod_rows, const float* styles_r, int sd_r, const float* styles_d, int sd_d,
                                  void* xch, void* sync, int dim, const cips3d_rng_job* job, float* zero_ptr, int zero_n,
                                  void* stream);

extern "C" int cips3d_style_phase(const cips3d_generator_plan* plan, const cips3d_forward_io* io, int mode, void* stream) {
  if (!plan || !io || mode < -1 || mode > 1) return CIPS3D_E_BADARG;
  const cips3d_generator_plan& P = *plan;
  const cips3d_forward_io& IO = *io;
  if (P.B <= 0 || P.n_map_r > CIPS3D_MAX_MAP_LAYERS || P.n_map_d > CIPS3D_MAX_MAP_LAYERS) return CIPS3D_E_BADARG;
  const int B = P.B;
  const int D = P.nerf.depth;
  const bool trunc = IO.mean_r && IO.mean_d && IO.trunc_psi < 1.f;
  const bool ranged = P.range_ws != nullptr;
  if (ranged && (P.range_ws_words <= 0 || P.range_ws_words > (1 << 30))) return CIPS3D_E_BADARG;

  // ---- mapping networks (model_v3.py:1299-1418); the last layer broadcasts w to every style slot.
  cips3d_linear_args ar[CIPS3D_MAX_MAP_LAYERS], ad[CIPS3D_MAX_MAP_LAYERS];
  const int nr = IO.z_r ? P.n_map_r : 0, nd = IO.z_d ? P.n_map_d : 0;
  {
    const float* x = IO.z_r;
    int64_t xs = P.z_dim;
    for (int i = 0; i < nr; ++i) {
      const bool last = i == nr - 1;
      float* out = last ? P.styles_r : P.lat[i & 1];
      ar[i] = cips3d_linear_args{x, xs, P.map_r_w[i], P.map_r_b[i], out,
                                 last ? (int64_t)(D + 1) * P.style_dim_r : (int64_t)P.style_dim_r, B,
                                 i == 0 ? P.z_dim : P.style_dim_r, P.style_dim_r, 1.f, 1.f, 0, 1, 1.f, 1.f, 0.f,
                                 (last && trunc) ? IO.mean_r : nullptr, IO.trunc_psi, last ? D + 1 : 1, P.style_dim_r};
      x = out; xs = P.style_dim_r;
    }
    x = IO.z_d;
    xs = P.z_dim;
    for (int i = 0; i < nd; ++i) {
      const bool last = i == nd - 1;
      const int in_dim = P.map_d_in[i];
      float* out = last ? P.styles_d : P.lat[2 + (i & 1)];
      ad[i] = cips3d_linear_args{x, xs, P.map_d_w[i], P.map_d_b[i], out,
                                 last ? (int64_t)P.n_latent * P.style_dim_d : (int64_t)P.style_dim_d, B, in_dim,
                                 P.style_dim_d, (1.f / sqrtf((float)in_dim)) * P.map_d_lr_mul, P.map_d_lr_mul,
                                 i == 0 ? 1 : 0, 1, 1.41421356237309515f, 1.f, 0.f,
                                 (last && trunc) ? IO.mean_d : nullptr, IO.trunc_psi, last ? P.n_latent : 1, P.style_dim_d};
      x = out; xs = P.style_dim_d;
    }
  }
  // fresh draws of this call: slices of one cips3d_rng_fill ride on the phase's launch(es); none of those: a launch of its own
  const bool want_rng = IO.rng_n_normal > 0 || IO.rng_n_uniform > 0;
  if (IO.rng_n_normal < 0 || IO.rng_n_uniform < 0 || (IO.rng_n_normal > 0 && !IO.rng_normal) ||
      (IO.rng_n_uniform > 0 && !IO.rng_uniform))
    return CIPS3D_E_BADARG;
  const long long rng_threads = want_rng ? cips3d_rng_fill_threads(IO.rng_n_normal, IO.rng_n_uniform) : 0;
  cips3d_rng_job whole{(unsigned)IO.rng_seed, (unsigned)(IO.rng_seed >> 32), (unsigned long long)IO.rng_base, IO.rng_normal,
                       (long long)IO.rng_n_normal, IO.rng_uniform, (long long)IO.rng_n_uniform, 0, rng_threads};

  // ---- ONE launch: resident workgroups, layer outputs handed over as tagged granules (linear.hip: style_phase_kernel).
  // Opt-in: measured 9 us SLOWER per forward than the launches below (~14 dependent memory round trips of
  // ~2.2 us inside the launch against six kernel boundaries).
  bool one = mode == 1;
  if (mode == -1) {
    const char* e = getenv("CIPS3D_STYLE_PHASE");
    one = e && e[0] == '1';
  }
  if (one) {
    const int rc = cips3d_style_phase_one_launch(ar, nr, ad, nd, P.film_table, P.film_n, P.film_rows, P.mod_table, P.mod_n, P.mod_rows,
                                                 P.styles_r, P.style_dim_r, P.styles_d, P.style_dim_d, P.style_xch, P.style_sync,
                                                 P.style_xch_dim, want_rng ? &whole : nullptr, ranged ? P.range_ws : nullptr,
                                                 ranged ? (int)P.range_ws_words : 0, stream);
    if (rc != CIPS3D_E_UNSUPP || mode == 1) return rc;
  }

  // ---- launches.  The two chains are independent: their i-th layers share a launch (8 dependent ~4.5 us launches -> 5),
  // and the FiLM heads of the (shorter) renderer chain ride on the decoder chain's next layer.
  bool film_done = false;
  {
    const int n_pair = nr < nd ? nr : nd;
    if (want_rng && n_pair == 0)
      TRY(cips3d_rng_fill(IO.rng_seed, IO.rng_base, IO.rng_normal, IO.rng_n_normal, IO.rng_uniform, IO.rng_n_uniform, stream));
    for (int i = 0; i < (nr > nd ? nr : nd); ++i) {
      if (i < nr && i < nd) {
        cips3d_rng_job job = whole;
        job.t1 = 0;
        if (want_rng) {      // slice i of n_pair, in whole blocks of 256 threads
          const long long per = ((rng_threads + n_pair - 1) / n_pair + 255) / 256 * 256;
          job.t0 = per * i < rng_threads ? per * i : rng_threads;
          job.t1 = per * (i + 1) < rng_threads ? per * (i + 1) : rng_threads;
        }
        TRY(cips3d_linear_pair(ar[i], ad[i], stream, want_rng ? &job : nullptr));
      } else if (i == nr && nr > 0 && P.film_n > 0) {
        // the renderer's W+ is complete: its FiLM heads share the launch of the decoder chain's next layer
        TRY(cips3d_linear_and_table(ad[i], P.film_table, P.film_n, P.film_rows, stream));
        film_done = true;
      } else {
        const cips3d_linear_args& a = i < nr ? ar[i] : ad[i];
        TRY(cips3d_linear(a.x, a.x_stride, a.W, a.bias, a.out, a.out_stride, a.B, a.in_dim, a.out_dim, a.w_scale, a.b_scale,
                          a.pixelnorm, a.lrelu, a.act_gain, a.out_scale, a.out_shift, a.trunc_mean, a.trunc_psi,
                          a.out_repeat, a.out_repeat_stride, stream));
      }
    }
  }
  // ---- style heads: FiLM gamma/beta of every SIREN layer; every decoder modulation
  if (!film_done) TRY(cips3d_linear_table(P.film_table, P.film_n, P.film_rows, B, stream));
  // Range tracking of the split-fp16 decoder (cips3d_range): the workspace -- amax slots every producing epilogue raises with
  // atomicMax, planes exponents, layer constants -- is zeroed by the launch of the modulation heads.
  if (ranged) TRY(cips3d_linear_table_zero(P.mod_table, P.mod_n, P.mod_rows, B, P.range_ws, (int)P.range_ws_words, stream));
  else TRY(cips3d_linear_table(P.mod_table, P.mod_n, P.mod_rows, B, stream));
  return 0;
}

extern "C" int cips3d_generator_forward(const cips3d_generator_plan* plan, const cips3d_forward_io* io,
                                        void* stream) {
  if (!plan || !io) return CIPS3D_E_BADARG;
  const cips3d_generator_plan& P = *plan;
  const cips3d_forward_io& IO = *io;
  if (P.B <= 0 || P.n_map_r > CIPS3D_MAX_MAP_LAYERS || P.n_map_d > CIPS3D_MAX_MAP_LAYERS ||
      P.n_dec_layers > CIPS3D_MAX_DEC_LAYERS || P.n_dec_layers < 2)
    return CIPS3D_E_BADARG;
  if (!IO.cam_poses || !IO.focals || !IO.near_ || !IO.far_ || !IO.rgb || !IO.thumb || !IO.xyz || !IO.mask)
    return CIPS3D_E_BADARG;
  const int B = P.B;
  if (IO.rgb_is_u8) {      // a uint8 image leaves through a fused stage only: decided before anything is enqueued
    const int n = P.n_dec_layers;
    bool ok = n >= 3 && P.layers[n - 2].kind == 0 && P.layers[n - 2].Cin == P.layers[n - 3].Cout &&
              P.layers[n - 2].Cout == P.layers[n - 3].Cout && P.layers[n - 1].Cin == P.layers[n - 3].Cout;
    if (ok) {
      const cips3d_dec_layer& h = P.layers[n - 3];
      ok = (h.kind == 1 && P.layers[n - 1].kind == 3 && cips3d_fused_up_conv_supported(h.Cout, h.H, h.W)) ||
           (h.kind == 0 && (h.flags & 128) && P.layers[n - 1].kind == 2 && cips3d_fused_flat_conv_supported(h.Cout, h.H, h.W));
    }
    if (!ok) return CIPS3D_E_UNSUPP;
  }
  const int gemm_flag = P.decoder_bf16 ? CIPS3D_GEMM_BF16 : 0;
  // decoder_bf16 == 2: the low-resolution GEMM results of the fused up-sampling stages (y_lo / y_next) live in HBM as bf16
  const int ybf_flag = P.decoder_bf16 == 2 ? CIPS3D_Y_BF16 : 0;

  // ---- mapping networks, FiLM heads, modulation heads, the call's draws (cips3d_style_phase); then the modulated weights:
  // the modulate table also writes every layer's range constants with this call's bound of |noise| (6: cips3d_rng_fill's
  // draws stay below 5.89).
  const bool ranged = P.range_ws != nullptr;
  if (ranged && (P.range_ws_words <= 0 || P.range_ws_words > (1 << 30) || !P.feat_amax || !P.feat_exp || !P.tmp_amax)) return CIPS3D_E_BADARG;
  if (IO.styles_resident) {
    // a frame of a sequence: styles, FiLM / modulation tables and the modulated weights (with their range constants) are the
    // plan's, as the last full forward left them.  What every forward measures anew is zeroed; the call's draws (if any) take
    // a launch of their own.
    // (the zeroing rides on the render launch: cips3d_nerf_params.zero_words, below)
    if (ranged && (P.range_volatile_words <= 0 || P.range_volatile_words > P.range_ws_words)) return CIPS3D_E_BADARG;
    if (IO.rng_n_normal < 0 || IO.rng_n_uniform < 0 || (IO.rng_n_normal > 0 && !IO.rng_normal) || (IO.rng_n_uniform > 0 && !IO.rng_uniform))
      return CIPS3D_E_BADARG;
    if (IO.rng_n_normal > 0 || IO.rng_n_uniform > 0)
      TRY(cips3d_rng_fill(IO.rng_seed, IO.rng_base, IO.rng_normal, IO.rng_n_normal, IO.rng_uniform, IO.rng_n_uniform, stream));
  } else {
    TRY(cips3d_style_phase(plan, io, -1, stream));
    TRY(cips3d_modulate_table(P.wm_table, P.wm_n, P.wm_rows, B, IO.noise_bound > 0.f ? IO.noise_bound : 6.f, stream));
  }

  // ---- NeRF: rays -> samples -> FiLM-SIREN -> compositing
  cips3d_nerf_params np = P.nerf;
  np.cam_poses = IO.cam_poses; np.focals = IO.focals; np.near_ = IO.near_; np.far_ = IO.far_;
  np.perturb_u = IO.perturb_u; np.sdf = IO.sdf;
  if (IO.ev_nerf_start) hipEventRecord(reinterpret_cast<hipEvent_t>(IO.ev_nerf_start), as_stream(stream));
  // the chunk partials are combined inside the render kernel when the shape allows it, else by cips3d_nerf_finish
  np.o_features = P.features; np.o_thumb = IO.thumb; np.o_xyz = IO.xyz; np.o_mask = IO.mask;
  np.mask_planar = IO.mask_planar;
  const bool fused_finish = cips3d_nerf_fuses_finish(&np) != 0;
  // the first decoder layer reads split-fp16 planes (flags bit 2): the render kernel's fused finish writes them directly,
  // the stand-alone finish writes fp32 into the spare activation buffer and a conversion pass follows.  bf16 planes16
  // (flags bit 5) always take the conversion pass.
  const bool feat_planes = (P.layers[0].flags & 4) != 0;
  const bool feat_p16 = feat_planes && (P.layers[0].flags & 32) != 0;
  np.features_planes = (feat_planes && !feat_p16 && fused_finish) ? 1 : 0;
  if (feat_planes && !feat_p16 && !ranged) return CIPS3D_E_BADARG;      // split-fp16 planes carry exponents: the plan owes the rows
  float* feat32 = (feat_planes && !np.features_planes) ? P.act[1] : P.features;
  np.o_features = feat32;
  // a frame of a sequence: what every forward measures anew (the volatile range rows; the style phase's zeroing launch in a full
  // forward) is cleared by the render launch on its way out -- nothing of that launch lives there, everything behind it does
  np.zero_words = nullptr; np.n_zero_words = 0;
  if (IO.styles_resident && ranged) { np.zero_words = P.range_ws; np.n_zero_words = P.range_volatile_words; }
  TRY(cips3d_nerf_render(&np, stream));
  if (IO.ev_nerf_stop) hipEventRecord(reinterpret_cast<hipEvent_t>(IO.ev_nerf_stop), as_stream(stream));
  if (!fused_finish)
    TRY(cips3d_nerf_finish(np.part, np.n_chunks, B, np.img_size, np.hidden, feat32, IO.thumb, IO.xyz, IO.mask, stream));
  if (feat_planes && !np.features_planes) {
    const int64_t hw0 = (int64_t)np.img_size * np.img_size;
    if (feat_p16) TRY(cips3d_to_planes16(feat32, P.features, B, np.hidden, hw0, stream));
    else {
      if (ranged) TRY(cips3d_absmax(feat32, B, (int64_t)np.hidden * hw0, P.feat_amax, stream));
      TRY(cips3d_to_planes(feat32, P.features, B, np.hidden, hw0, ranged ? P.feat_amax : nullptr, ranged ? P.feat_exp : nullptr,
                           ranged ? P.feat_pmax : nullptr, stream));
    }
  }

  // ---- decoder (model_v3.py:592-637)
  int n_marks = 0;
  auto mark = [&](int kind, int Ci, int Co, int H) {          // timeline for bench.py (cips3d_forward_io.ev_marks)
    if (!IO.ev_marks || n_marks >= IO.n_ev_marks) return;
    hipEventRecord(reinterpret_cast<hipEvent_t>(IO.ev_marks[n_marks]), as_stream(stream));
    if (IO.ev_info) {
      IO.ev_info[4 * n_marks] = kind; IO.ev_info[4 * n_marks + 1] = Ci; IO.ev_info[4 * n_marks + 2] = Co; IO.ev_info[4 * n_marks + 3] = H;
    }
    ++n_marks;
    if (IO.ev_count) *IO.ev_count = n_marks;
  };
  mark(CIPS3D_MARK_START, 0, 0, 0);
  mark(CIPS3D_MARK_START, 0, 0, 0);     // (two records back to back: their distance is what a record itself costs)
  const float* x = P.features;
  // range rows of x (ranged plans).  fp32 x: the measured maximum of its values (x_amax; nullptr: no producer tracked this tensor
  // -- a split GEMM that reads it measures it first, amax_of).  Planes x: the exponents of its pixel blocks (x_exp; nullptr: the
  // render kernel's feature planes, CIPS3D_FEATURES_EXP everywhere)
  const float* x_amax = nullptr;
  const int32_t* x_exp = (ranged && feat_planes && !np.features_planes && !feat_p16) ? P.feat_exp : nullptr;
  const float* x_pmax = x_exp ? P.feat_pmax : nullptr;      // patch maxima of planes x (nullptr: the feature planes, |f| <= 1)
  auto amax_of = [&](const float* t, int C, int64_t hw) -> int {
    if (!ranged || x_amax) return 0;
    const int rc = cips3d_absmax(t, B, (int64_t)C * hw, P.tmp_amax, stream);
    x_amax = P.tmp_amax;
    return rc;
  };
  const float* skip = nullptr;
  int act_i = 0, skip_i = 0;
  float* ylo_cur = P.y_lo;                       // low-resolution GEMM result of the stage being run
  float* ylo_alt = P.y_lo2;                      // ... of the next stage, when the current stage's kernel computes it
  bool ylo_ready = false;                        // ylo_cur was already filled by the previous stage's kernel
  bool u8_done = false;                          // the final image left through a fused stage (the only form that writes uint8)
  // ToRGB folding: a non-up-sampling ToRGB that follows a StyledConv is computed from that conv's registers (partial sums
  // per row block, cips3d_modconv1x1_torgb); the slots of consecutive such layers are folded by ONE cips3d_torgb_reduce
  // when their sum is first needed (the skip chain at an unchanged resolution is a plain sum).
  int fold_slots = 0, fold_nb = 0, fold_H = 0, fold_W = 0;
  const float* fold_bias[CIPS3D_TORGB_FOLD_MAX];
  auto fold_flush = [&](float* dst) -> int {
    if (fold_slots == 0) return 0;
    const int rc = cips3d_torgb_reduce(P.rgb_part, fold_slots, fold_bias, fold_nb, skip, dst, B, (int64_t)fold_H * fold_W, stream);
    mark(CIPS3D_MARK_TORGB, 0, 3, fold_H);
    skip = dst;
    fold_slots = fold_nb = 0;
    return rc;
  };
  for (int li = 0; li < P.n_dec_layers; ++li) {
    const cips3d_dec_layer& L = P.layers[li];
    const bool last = li == P.n_dec_layers - 1;
    if (L.kind == 0 || L.kind == 1) {
      const float* nz = L.noise_index >= 0 ? IO.noise[L.noise_index] : nullptr;
      const int64_t nbs = L.noise_index >= 0 ? IO.noise_bstride[L.noise_index] : 0;
      float* out = P.act[act_i];
      // resolution changes: the folded sum becomes the skip of this stage.  When the stage's low-resolution GEMM is the next
      // launch and runs on split planes, the fold rides on it (cips3d_reduce_job) instead of taking a launch of its own
      cips3d_reduce_job ride_job{};
      bool ride = false;
      if (L.kind == 1 && fold_slots) {
        static const int ride_on = getenv("CIPS3D_TORGB_RIDE") ? atoi(getenv("CIPS3D_TORGB_RIDE")) : 1;      // A/B knob
        const bool fused_next = li + 2 < P.n_dec_layers && P.layers[li + 1].kind == 0 && P.layers[li + 2].kind == 3 &&
                                P.layers[li + 1].Cin == L.Cout && P.layers[li + 1].Cout == L.Cout && P.layers[li + 2].Cin == L.Cout &&
                                cips3d_fused_up_conv_supported(L.Cout, L.H, L.W);
        const int64_t fhw = (int64_t)fold_H * fold_W;
        if (ride_on && ranged && fused_next && !ylo_ready && (L.flags & 4) && !(L.flags & 32) && fold_slots <= 48 && fhw % 4 == 0 &&
            (int64_t)B * 3 * fhw / 4 <= ((int64_t)L.H * L.W + 127) / 128 * B * 128) {
          ride_job.part = P.rgb_part;
          for (int k = 0; k < fold_nb; ++k) ride_job.bias[k] = fold_bias[k];
          ride_job.skip = skip;
          ride_job.out = P.skip[skip_i];
          ride_job.n4 = (int64_t)B * 3 * fhw / 4;
          ride_job.HW4 = fhw / 4;
          ride_job.slot_stride = (int64_t)B * 3 * fhw;
          ride_job.n_slots = fold_slots;
          ride_job.n_bias = fold_nb;
          ride = true;
          skip = P.skip[skip_i];
          fold_slots = fold_nb = 0;
        } else {
          TRY(fold_flush(P.skip[skip_i]));
        }
        skip_i ^= 1;
      }
      // up-sampling stage [StyledConv(up), StyledConv, ToRGB(up)] with equal widths: low-res GEMM, then ONE fused
      // kernel for FIR + act -> conv2 + act -> ToRGB (+ FIR-upsampled skip); the full-resolution intermediate
      // never reaches HBM and the last stage stores only the image.
      // FLAT stage (flags bit 7, set by the plan): [StyledConv, StyledConv, ToRGB] of equal widths at one resolution -- a block
      // above the last up-sampling one (model_v3.py:553-590 builds every block up to size_end) -- through the same kernel
      // without the FIR (CIPS3D_STAGE_FLAT): conv1's GEMM, then one launch; chained with its neighbours like the others.
      const bool same_w = li + 2 < P.n_dec_layers && P.layers[li + 1].kind == 0 && P.layers[li + 1].Cin == L.Cout &&
                          P.layers[li + 1].Cout == L.Cout && P.layers[li + 2].Cin == L.Cout;
      const bool flat = L.kind == 0 && (L.flags & 128);
      if (flat && (!same_w || P.layers[li + 2].kind != 2 || (L.flags & 4) || !cips3d_fused_flat_conv_supported(L.Cout, L.H, L.W)))
        return CIPS3D_E_BADARG;                      // the plan promised a block this kernel takes
      if (flat && fold_slots) {                      // (a pending fold of ToRGB partial sums becomes this block's skip image)
        TRY(fold_flush(P.skip[skip_i]));
        skip_i ^= 1;
      }
      if (flat || (L.kind == 1 && same_w && P.layers[li + 2].kind == 3 && cips3d_fused_up_conv_supported(L.Cout, L.H, L.W))) {
        const int OHs = flat ? L.H : 2 * L.H, OWs = flat ? L.W : 2 * L.W;       // the stage's output resolution
        const cips3d_dec_layer& L2 = P.layers[li + 1];
        const cips3d_dec_layer& L3 = P.layers[li + 2];
        const bool stage_last = li + 2 == P.n_dec_layers - 1;
        const float* nz2 = L2.noise_index >= 0 ? IO.noise[L2.noise_index] : nullptr;
        const int64_t nbs2 = L2.noise_index >= 0 ? IO.noise_bstride[L2.noise_index] : 0;
        if ((L.flags & 1) && !ylo_ready) return CIPS3D_E_BADARG;     // chained weights without the stage that chains them
        // ranged: y_lo's maximum lives in the up-conv's amax row (written by whoever computes y_lo)
        const bool stage_ranged = ranged && L.amax && L.lconst && L2.lconst;
        if (!ylo_ready) {
          cips3d_range rg{};
          rg.out_amax = stage_ranged ? L.amax : nullptr;
          if ((L.flags & 4) && (L.flags & 32))       // the run's last activation arrives as planes16 (bf16 mode)
            TRY(cips3d_modconv1x1_planes16(x, L.wm, ylo_cur, ybf_flag ? 2 : 0, B, L.Cin, L.Cout, (int64_t)L.H * L.W, 0, nullptr, 0,
                                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream));
          else if (L.flags & 4) {  // ... as split-fp16 planes
            rg.x_exp = x_exp;
            rg.x_exp_const = CIPS3D_FEATURES_EXP;
            rg.ride = ride ? &ride_job : nullptr;
            TRY(cips3d_modconv1x1_planes(x, L.wm, ylo_cur, ybf_flag ? 2 : 0, B, L.Cin, L.Cout, (int64_t)L.H * L.W, 0, nullptr, 0,
                                         nullptr, nullptr, nullptr, nullptr, nullptr, ranged ? &rg : nullptr, stream));
            ride = false;
          } else {
            if (L.flags & 2) TRY(amax_of(x, L.Cin, (int64_t)L.H * L.W));
            rg.x_amax = x_amax;
            TRY(cips3d_modconv1x1(x, L.wm, ylo_cur, B, L.Cin, L.Cout, (int64_t)L.H * L.W,
                                  0 | gemm_flag | ybf_flag | ((L.flags & 2) ? CIPS3D_GEMM_SPLIT : 0), nullptr, 0, nullptr, nullptr,
                                  ranged ? &rg : nullptr, stream));
          }
          mark(CIPS3D_MARK_LOWRES_GEMM, L.Cin, L.Cout, L.H);
        }
        // the next stage's 1x1 up-conv reads nothing but this stage's output: when the plan packed its weights for it
        // (flags bit 0) this kernel computes that GEMM from its registers and the activations are never stored
        const cips3d_dec_layer* LN = li + 3 < P.n_dec_layers ? &P.layers[li + 3] : nullptr;
        const bool ln_head = LN && (LN->kind == 1 || (LN->kind == 0 && (LN->flags & 128)));
        const bool chain = ln_head && (LN->flags & 1) && LN->Cin == L.Cout && LN->Cout * 2 == L.Cout &&
                           LN->H == OHs && LN->W == OWs && ylo_alt && cips3d_fused_up_conv_chains(L.Cout);
        if (ln_head && (LN->flags & 1) && !chain) return CIPS3D_E_BADARG;
        float* out2 = (stage_last || chain) ? nullptr : P.act[act_i];
        float* rgb = stage_last ? IO.rgb : P.skip[skip_i];
        const int u8_flag = (stage_last && IO.rgb_is_u8) ? CIPS3D_RGB_U8 : 0;
        if (stage_last) u8_done = true;
        // flags bit 4: conv2's (and the chained up-conv's) weights are CIPS3D_MOD_SPLIT16-packed: the stage runs split-fp16
        const int stage_split = (L2.flags & 16) ? CIPS3D_GEMM_SPLIT : 0;
        if (chain && ((LN->flags & 16) != (L2.flags & 16))) return CIPS3D_E_BADARG;
        if (stage_split && ranged && (!stage_ranged || (chain && !LN->amax))) return CIPS3D_E_BADARG;   // the plan owes the rows
        cips3d_range srg{};
        if (stage_ranged) {
          srg.x_amax = L.amax; srg.lconst = L.lconst; srg.lconst2 = L2.lconst;
          srg.next_amax = chain ? LN->amax : nullptr;
          // The stage that feeds the LAST up-sampling stage publishes the bound of |y_next| instead of measuring it: that
          // consumer splits once more (act1) and nothing after it, so ~2^8 of slack is harmless there, and the measurement is
          // dearest exactly here (one reduction + atomic per workgroup, 4096 workgroups at C = 64: 4 us).  A non-demodulated
          // next up-conv has no sqrt(C) row bound: measured then.
          const bool ln_last = chain && li + 6 >= P.n_dec_layers;
          srg.next_gain = (ln_last && (LN->flags & 64)) ? sqrtf((float)L.Cout) : 0.f;
        }
        TRY(cips3d_fused_up_conv_next(ylo_cur, L.fir, nz, nbs, L.noise_w, L.bias, L2.wm, nz2, nbs2, L2.noise_w, L2.bias, out2,
                                      L3.wm, L3.bias, skip, (flat ? CIPS3D_STAGE_FLAT : 1) | gemm_flag | ybf_flag | stage_split | u8_flag, rgb, chain ? LN->wm : nullptr,
                                      chain ? ylo_alt : nullptr, B, L.Cout, L.H, L.W, stage_ranged ? &srg : nullptr, stream));
        mark(CIPS3D_MARK_FUSED_STAGE, L.Cout, chain ? LN->Cout : 0, OHs);
        ylo_ready = chain;
        if (chain) { float* t = ylo_cur; ylo_cur = ylo_alt; ylo_alt = t; }
        x = out2;
        x_amax = nullptr;                // (a stored out2 is not tracked: its reader measures it)
        x_exp = nullptr;
        act_i ^= 1;
        skip = rgb;
        skip_i ^= 1;
        li += 2;
        continue;
      }
      if (L.kind == 0 && (L.flags & 4)) {
        // a layer of the split-fp16 run (csrc/chain.hip): planes in; planes out (flags bit 3) with the ToRGB that follows
        // folded from the registers -- nothing else may read this activation
        const cips3d_dec_layer* T = li + 1 < P.n_dec_layers ? &P.layers[li + 1] : nullptr;
        const int64_t hw = (int64_t)L.H * L.W;
        const bool has_rgb = T && T->kind == 2;
        const bool fold = has_rgb && li + 1 != P.n_dec_layers - 1 && T->Cin == L.Cout && P.rgb_part &&
                          fold_nb < CIPS3D_TORGB_FOLD_MAX && fold_slots + 16 <= P.rgb_part_slots &&
                          (fold_slots == 0 || (fold_H == L.H && fold_W == L.W));
        const bool p16 = (L.flags & 32) != 0;           // bf16 mode: planes16 and CIPS3D_MOD_BF16 weights
        const int fmt = (L.flags & 8) ? (p16 ? 3 : 1) : 0;   // planes for the next layer of the run, or fp32 when the run ends here
        if ((!p16 && !(L.flags & 2)) || (fmt != 0 && has_rgb && !fold)) return CIPS3D_E_BADARG;   // the plan promised otherwise
        int nblk = 0;
        cips3d_range rg{};
        const bool lr = ranged && !p16;
        if (lr) {
          if (!L.amax || !L.aexp || !L.lconst) return CIPS3D_E_BADARG;      // the plan owes the rows
          if (fmt == 1 && !L.pmax) return CIPS3D_E_BADARG;
          // every other layer of the run leaves patch maxima for its consumer; the layers in between derive max|in| from the
          // input's exponent (chain.hip: a bound of a bound, ~2^5 looser -- and half the tracking cost of the run)
          rg.x_exp = x_exp; rg.x_exp_const = CIPS3D_FEATURES_EXP; rg.x_pmax = x_pmax; rg.x_max_const = x_exp ? 0.f : 1.f;
          rg.lconst = L.lconst; rg.out_exp = L.aexp; rg.out_pmax = (fmt == 1 && !x_pmax) ? L.pmax : nullptr;
          rg.out_amax = fmt == 1 ? nullptr : L.amax;       // (the exit's per-sample maximum, for the fused stage that reads it)
          rg.half_chip = IO.views_in_flight > 1;           // (another view's launches beside this one's: chain.hip)
        }
        if (p16)
          TRY(cips3d_modconv1x1_planes16(x, L.wm, out, fmt, B, L.Cin, L.Cout, hw, 1, nz, nbs, L.noise_w, L.bias,
                                         fold ? T->wm : nullptr, fold ? P.rgb_part + (int64_t)fold_slots * B * 3 * hw : nullptr,
                                         &nblk, nullptr, stream));
        else
          TRY(cips3d_modconv1x1_planes(x, L.wm, out, fmt, B, L.Cin, L.Cout, hw, 1, nz, nbs, L.noise_w, L.bias,
                                       fold ? T->wm : nullptr, fold ? P.rgb_part + (int64_t)fold_slots * B * 3 * hw : nullptr, &nblk,
                                       lr ? &rg : nullptr, stream));
        mark(CIPS3D_MARK_PLANES_GEMM, L.Cin, L.Cout, L.H);
        x_amax = (lr && fmt != 1) ? L.amax : nullptr;
        x_exp = (lr && fmt == 1) ? L.aexp : nullptr;
        x_pmax = (lr && fmt == 1) ? rg.out_pmax : nullptr;
        if (fold) {
          fold_slots += nblk;
          fold_bias[fold_nb++] = T->bias;
          fold_H = L.H; fold_W = L.W;
          ++li;
        }
        x = out;
        act_i ^= 1;
        continue;
      }
      if (L.kind == 0) {
        const cips3d_dec_layer* T = li + 1 < P.n_dec_layers ? &P.layers[li + 1] : nullptr;
        const int64_t hw = (int64_t)L.H * L.W;
        const bool fold = T && T->kind == 2 && li + 1 != P.n_dec_layers - 1 && T->Cin == L.Cout && hw % 4 == 0 && P.rgb_part &&
                          fold_nb < CIPS3D_TORGB_FOLD_MAX && fold_slots + 16 <= P.rgb_part_slots &&
                          (fold_slots == 0 || (fold_H == L.H && fold_W == L.W));
        const int split_flag = (L.flags & 2) ? CIPS3D_GEMM_SPLIT : 0;
        cips3d_range rg{};
        if (split_flag) TRY(amax_of(x, L.Cin, hw));
        rg.x_amax = x_amax;
        rg.out_amax = (ranged && L.amax) ? L.amax : nullptr;
        x_amax = rg.out_amax;            // (of this layer's output, from here on)
        x_exp = nullptr;
        if (fold) {
          int nblk = 0;
          TRY(cips3d_modconv1x1_torgb(x, L.wm, out, B, L.Cin, L.Cout, hw, 1 | gemm_flag | split_flag, nz, nbs, L.noise_w, L.bias, T->wm,
                                      P.rgb_part + (int64_t)fold_slots * B * 3 * hw, &nblk, ranged ? &rg : nullptr, stream));
          mark(CIPS3D_MARK_GEMM, L.Cin, L.Cout, L.H);
          fold_slots += nblk;
          fold_bias[fold_nb++] = T->bias;
          fold_H = L.H; fold_W = L.W;
          x = out;
          act_i ^= 1;
          ++li;                                  // the ToRGB layer is done (its sum is pending in the slots)
          continue;
        }
        TRY(cips3d_modconv1x1(x, L.wm, out, B, L.Cin, L.Cout, hw, 1 | gemm_flag | split_flag, nz, nbs, L.noise_w, L.bias,
                              ranged ? &rg : nullptr, stream));
        mark(CIPS3D_MARK_GEMM, L.Cin, L.Cout, L.H);
      } else {
        if (L.flags & 1) return CIPS3D_E_BADARG;   // chained packs only exist for stages that take the fused route above
        if (L.flags & 4) return CIPS3D_E_BADARG;   // planes reach an up-conv only on the fused route (the plan guarantees it)
        cips3d_range rg{};
        if (L.flags & 2) TRY(amax_of(x, L.Cin, (int64_t)L.H * L.W));
        rg.x_amax = x_amax;
        TRY(cips3d_modconv1x1(x, L.wm, P.y_lo, B, L.Cin, L.Cout, (int64_t)L.H * L.W,
                              0 | gemm_flag | ((L.flags & 2) ? CIPS3D_GEMM_SPLIT : 0), nullptr, 0, nullptr, nullptr,
                              ranged ? &rg : nullptr, stream));
        float* fir_amax = (ranged && L.amax) ? L.amax : nullptr;      // (this route leaves the row unused: y_lo is not tracked here)
        TRY(cips3d_up2_fir_act(P.y_lo, L.fir, out, B, L.Cout, L.H, L.W, nz, nbs, L.noise_w, L.bias, fir_amax, stream));
        mark(CIPS3D_MARK_OTHER, L.Cout, L.Cout, 2 * L.H);
        x_amax = fir_amax;               // recorded by the FIR kernel itself (else its reader measures it)
        x_exp = nullptr;
      }
      x = out;
      act_i ^= 1;
    } else if (L.kind == 2 || L.kind == 3) {
      if (fold_slots) {                          // a ToRGB that could not be folded: settle the pending sum first
        TRY(fold_flush(P.skip[skip_i]));
        skip_i ^= 1;
      }
      if (last && IO.rgb_is_u8) return CIPS3D_E_UNSUPP;     // a stand-alone ToRGB writes fp32 only
      float* out = last ? IO.rgb : P.skip[skip_i];
      TRY(cips3d_torgb(x, L.wm, L.bias, skip, L.kind == 3 ? 1 : 0, L.fir, out, B, L.Cin, L.H, L.W, stream));
      mark(CIPS3D_MARK_TORGB, L.Cin, 3, L.kind == 3 ? 2 * L.H : L.H);
      skip = out;
      skip_i ^= 1;
    } else {
      return CIPS3D_E_BADARG;
    }
  }
  if (IO.rgb_is_u8 && !u8_done) return CIPS3D_E_UNSUPP;
  return 0;
}

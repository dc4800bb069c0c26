"""Host side of cips3d_generator_forward: builds the plan (device pointers into a Generator's
parameters + a persistent workspace) once per (batch, img_size, N_samples, static_viewdirs) and fills
the small per-call io struct.  One ctypes call then enqueues the whole forward.
"""
import ctypes as C
import os

import torch

from . import _lib, hip
from ._lib import dev_ptr

MAX_MAP = 8
MAX_DEC = 40


class DecLayer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("noise_index", C.c_int32), ("flags", C.c_int32), ("pad_", C.c_int32), ("wm", C.c_void_p),
                ("bias", C.c_void_p), ("noise_w", C.c_void_p), ("fir", C.c_void_p),
                ("amax", C.c_void_p), ("aexp", C.c_void_p), ("pmax", C.c_void_p), ("lconst", C.c_void_p)]


class GeneratorPlan(C.Structure):
    _fields_ = [("B", C.c_int32), ("z_dim", C.c_int32), ("n_map_r", C.c_int32), ("n_map_d", C.c_int32),
                ("style_dim_r", C.c_int32), ("style_dim_d", C.c_int32), ("n_latent", C.c_int32),
                ("n_dec_layers", C.c_int32),
                ("map_r_w", C.c_void_p * MAX_MAP), ("map_r_b", C.c_void_p * MAX_MAP),
                ("map_d_w", C.c_void_p * MAX_MAP), ("map_d_b", C.c_void_p * MAX_MAP),
                ("map_d_in", C.c_int32 * MAX_MAP), ("map_d_lr_mul", C.c_float), ("decoder_bf16", C.c_int32),
                ("lat", C.c_void_p * 4), ("styles_r", C.c_void_p), ("styles_d", C.c_void_p),
                ("film_table", C.c_void_p), ("film_n", C.c_int32), ("film_rows", C.c_int32),
                ("mod_table", C.c_void_p), ("mod_n", C.c_int32), ("mod_rows", C.c_int32),
                ("wm_table", C.c_void_p), ("wm_n", C.c_int32), ("wm_rows", C.c_int32),
                ("nerf", _lib.NerfParams), ("features", C.c_void_p),
                ("layers", DecLayer * MAX_DEC),
                ("act", C.c_void_p * 2), ("y_lo", C.c_void_p), ("y_lo2", C.c_void_p), ("skip", C.c_void_p * 2),
                ("rgb_part", C.c_void_p),
                ("rgb_part_slots", C.c_int64),
                ("range_ws", C.c_void_p), ("range_ws_words", C.c_int64), ("feat_amax", C.c_void_p), ("feat_exp", C.c_void_p),
                ("feat_pmax", C.c_void_p), ("tmp_amax", C.c_void_p),
                ("style_xch", C.c_void_p), ("style_sync", C.c_void_p), ("style_xch_dim", C.c_int32), ("range_volatile_words", C.c_int32)]


class ForwardIO(C.Structure):
    _fields_ = [("z_r", C.c_void_p), ("z_d", C.c_void_p), ("mean_r", C.c_void_p), ("mean_d", C.c_void_p),
                ("trunc_psi", C.c_float), ("pad_", C.c_int32),
                ("cam_poses", C.c_void_p), ("focals", C.c_void_p), ("near_", C.c_void_p), ("far_", C.c_void_p),
                ("perturb_u", C.c_void_p), ("sdf", C.c_void_p),
                ("noise", C.c_void_p * MAX_DEC), ("noise_bstride", C.c_int64 * MAX_DEC),
                ("rgb", C.c_void_p), ("thumb", C.c_void_p), ("xyz", C.c_void_p), ("mask", C.c_void_p),
                ("ev_nerf_start", C.c_void_p), ("ev_nerf_stop", C.c_void_p),
                ("rng_seed", C.c_uint64), ("rng_base", C.c_uint64), ("rng_normal", C.c_void_p), ("rng_n_normal", C.c_int64),
                ("rng_uniform", C.c_void_p), ("rng_n_uniform", C.c_int64),
                ("noise_bound", C.c_float), ("views_in_flight", C.c_int32),
                ("ev_marks", C.c_void_p), ("ev_info", C.c_void_p), ("ev_count", C.c_void_p), ("n_ev_marks", C.c_int32),
                ("styles_resident", C.c_int32), ("rgb_is_u8", C.c_int32), ("mask_planar", C.c_int32)]


# A fused up-sampling stage also computes the next stage's low-resolution GEMM (cips3d_fused_up_conv_next); 0 = every
# stage launches its own (A/B knob)
CHAIN_STAGES = os.environ.get("CIPS3D_CHAIN_STAGES", "1") != "0"
# The run of StyledConvs at the NeRF resolution keeps its activations as split-fp16 planes (csrc/chain.hip); 0 = every layer
# reads / writes fp32 and splits in registers (A/B knob)
PLANES_RUN = os.environ.get("CIPS3D_PLANES", "1") != "0"
# ... and as one bf16 plane in the bf16 decoder modes (cips3d_modconv1x1_planes16); 0 = fp32 activations, rounded in registers
PLANES16_RUN = os.environ.get("CIPS3D_PLANES16", "1") != "0"
# Decoder blocks ABOVE the NeRF resolution that do not up-sample ([StyledConv, StyledConv, ToRGB] at one resolution: the 512 / 1024
# blocks of a 256^2 generator) run as ONE launch of the fused stage kernel in its flat form (CIPS3D_STAGE_FLAT), chained like the
# up-sampling stages; 0 = two GEMM launches + a ToRGB launch per block (A/B knob)
FLAT_STAGES = os.environ.get("CIPS3D_FLAT_STAGES", "1") != "0"


class PlanUnsupported(RuntimeError):
    """The configuration cannot be planned (k != 1 or channel counts the MFMA GEMM does not tile)."""


def _upload(array):
    raw = bytes(memoryview(array))
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8)


class ForwardPlan:
    def __init__(self, G, B, img_size, N_samples, static_viewdirs, n_chunks=None, lane=0):
        """lane: which set of the modules' style tables (FiLM table, decoder modulations, their staging buffers) the plan runs
        on.  Lane 0 is the one every other path of the package shares; a lane k > 0 is private to the plans of that lane, so that
        forwards issued on different streams (Generator.forward keys its plans by the current stream) can be in flight together."""
        lib = _lib.load()
        self.lane = lane
        if lib.cips3d_sizeof_plan() != C.sizeof(GeneratorPlan) or lib.cips3d_sizeof_io() != C.sizeof(ForwardIO):
            raise RuntimeError("cips3d_generator_plan / cips3d_forward_io layout mismatch between python and the library")
        dev = G.renderer.sigmoid_beta.device
        ren, dec = G.renderer, G.decoder
        D, H = ren.N_layers_renderer, ren.hidden_dim
        self.B, self.img_size, self.N, self.static = B, img_size, N_samples, bool(static_viewdirs)
        self.device = dev
        self._keep = []
        p = GeneratorPlan()
        p.B, p.z_dim = B, G.z_dim
        p.style_dim_r, p.style_dim_d = ren.style_dim, dec.style_dim
        p.n_latent = dec.n_latent
        p.decoder_bf16 = (2 if getattr(dec, "bf16_storage", False) else 1) if getattr(dec, "bf16", False) else 0

        # ---- mapping networks
        map_r = list(G.style)
        map_d = list(G.style_decoder)[1:]
        if len(map_r) > MAX_MAP or len(map_d) > MAX_MAP:
            raise PlanUnsupported("too many mapping layers")
        p.n_map_r, p.n_map_d = len(map_r), len(map_d)
        for i, l in enumerate(map_r):
            p.map_r_w[i], p.map_r_b[i] = dev_ptr(l.weight), dev_ptr(l.bias)
        for i, l in enumerate(map_d):
            p.map_d_w[i], p.map_d_b[i] = dev_ptr(l.weight), dev_ptr(l.bias)
            p.map_d_in[i] = l.weight.shape[1]
        p.map_d_lr_mul = float(map_d[0].lr_mul)
        lat_w = max(p.style_dim_r, p.style_dim_d, p.z_dim)
        lat = torch.empty(4, B, lat_w, device=dev)
        for i in range(4):
            p.lat[i] = lat[i].data_ptr()
        # workspace of the one-launch style phase (cips3d_style_phase): tagged granules + the generation word
        xdim = (max([lat_w] + [l.weight.shape[1] for l in map_d]) + 3) // 4 * 4
        self.style_xch = torch.zeros(2 * MAX_MAP * B * xdim, dtype=torch.int64, device=dev)
        self.style_sync = torch.zeros(64, dtype=torch.int32, device=dev)
        p.style_xch, p.style_sync, p.style_xch_dim = self.style_xch.data_ptr(), self.style_sync.data_ptr(), xdim

        # ---- FiLM heads (renderer owns styles staging + film + table)
        styles_r, film, film_tab = ren._film_table(B, dev, lane)
        if film_tab._dev is None:
            film_tab._upload()
        p.styles_r = styles_r.data_ptr()
        p.film_table, p.film_n, p.film_rows = film_tab._dev.data_ptr(), len(film_tab._descs), film_tab._rows
        self.styles_r = styles_r

        # ---- decoder modulations
        styles_d, s_buf, mod_tab, offs, total = dec._style_table(B, dev, lane)
        if mod_tab._dev is None:
            mod_tab._upload()
        p.styles_d = styles_d.data_ptr()
        p.mod_table, p.mod_n, p.mod_rows = mod_tab._dev.data_ptr(), len(mod_tab._descs), mod_tab._rows
        self.styles_d = styles_d

        # ---- decoder layer list + modulated-weight table
        from .decoder import StyledConv
        seq = dec._mod_layers()
        if len(seq) > MAX_DEC:
            raise PlanUnsupported("too many decoder layers")
        res = img_size
        wm_sizes, layer_info = [], []
        noise_idx = 0
        max_act, max_lo, max_skip = 0, 1, 1
        for idx, (m, _) in enumerate(seq):
            conv = m.conv
            if conv.kernel_size != 1:
                raise PlanUnsupported("kernel_size != 1")
            if isinstance(m, StyledConv):
                if not conv.fast(res * res):
                    raise PlanUnsupported(f"conv {conv.in_channel}->{conv.out_channel} @ {res}^2 not tiled by the MFMA GEMM")
                kind = 1 if conv.upsample else 0
                info = dict(kind=kind, Cin=conv.in_channel, Cout=conv.out_channel, H=res, W=res, noise_index=noise_idx,
                            bias=m.activate.bias, noise_w=m.noise.weight, fir=conv.blur.kernel if conv.upsample else None,
                            packed=True, conv=conv)
                noise_idx += 1
                if conv.upsample:
                    max_lo = max(max_lo, B * conv.out_channel * res * res)
                    res *= 2
                info["Hout"] = res
                max_act = max(max_act, B * conv.out_channel * res * res)
            else:
                up = bool(m.upsample)
                if res % 4 != 0:
                    raise PlanUnsupported("toRGB resolution not a multiple of 4")
                info = dict(kind=3 if up else 2, Cin=conv.in_channel, Cout=3, H=res, W=res, noise_index=-1, bias=m.bias,
                            noise_w=None, fir=m.upsample.kernel if up else None, packed=False, conv=conv, Hout=res)
                max_skip = max(max_skip, B * 3 * res * res)
            wm_sizes.append(B * conv.out_channel * conv.in_channel)
            layer_info.append(info)
        self.out_res = res
        self._layer_info = layer_info
        self.noise_sizes = [li["Hout"] for li in layer_info if li["kind"] in (0, 1)]
        # An up-sampling stage [StyledConv(up), StyledConv, ToRGB(up)] of equal widths runs as low-res GEMM + one fused kernel
        # (forward.hip).  When the NEXT stage is one too and the fused kernel has the chained form for this width, that kernel
        # also computes the next stage's low-res GEMM from its registers: the next up-conv's weights are then packed in the
        # chained order and its own GEMM launch disappears.
        # A block of the same shape that does NOT up-sample, above the NeRF resolution (the blocks at the NeRF resolution belong
        # to the planes run below), takes the same kernel in its flat form: stage_kind 2.  (At the NeRF resolution itself -- a 64^2
        # generator's 128 .. 1024 blocks -- the four flat launches take what the eight planes launches they replace do:
        # 0.456 against 0.457 ms per view, measured; left with the run.)
        def stage_kind(i):
            if i + 2 >= len(layer_info):
                return 0
            a, b_, c = layer_info[i], layer_info[i + 1], layer_info[i + 2]
            if not (b_["kind"] == 0 and b_["Cin"] == a["Cout"] and b_["Cout"] == a["Cout"] and c["Cin"] == a["Cout"]):
                return 0
            if a["kind"] == 1 and c["kind"] == 3 and lib.cips3d_fused_up_conv_supported(a["Cout"], a["H"], a["W"]):
                return 1
            if (FLAT_STAGES and a["kind"] == 0 and c["kind"] == 2 and a["H"] > img_size
                    and lib.cips3d_fused_flat_conv_supported(a["Cout"], a["H"], a["W"])):
                return 2
            return 0

        def fused_stage(i):
            return stage_kind(i) == 1
        for i, li in enumerate(layer_info):
            if stage_kind(i) == 2:
                li["flat_head"] = True
                max_lo = max(max_lo, B * li["Cout"] * li["H"] * li["W"])       # conv1's GEMM result, at the block's resolution
        # the image can leave as uint8 (cips3d_forward_io.rgb_is_u8) when the decoder ends in a fused stage
        self.u8_capable = len(layer_info) >= 3 and stage_kind(len(layer_info) - 3) != 0
        for i in range(len(layer_info) - 3):
            a, nx = layer_info[i], layer_info[i + 3]
            if (CHAIN_STAGES and stage_kind(i) and stage_kind(i + 3) and lib.cips3d_fused_up_conv_chains(a["Cout"])
                    and nx["Cin"] == a["Cout"] and nx["Cout"] * 2 == a["Cout"] and nx["H"] == a["Hout"]):
                nx["chained"] = True
        # Which packed layers the stand-alone GEMM consumes (everything but conv2 of a fused stage and chained up-convs): in the
        # default "fp32" precision those are packed as split-fp16 fragments and run in CIPS3D_GEMM_SPLIT mode.
        use_split = bool(getattr(dec, "split", False)) and not getattr(dec, "bf16", False)
        for i, li in enumerate(layer_info):
            if not (use_split and li["packed"]):
                continue
            conv2_of_fused = li["kind"] == 0 and i >= 1 and stage_kind(i - 1) != 0
            chained = bool(li.get("chained"))
            li["split"] = not conv2_of_fused and not chained
            li["split16"] = conv2_of_fused or chained           # consumed inside cips3d_fused_up_conv_next
        # The run of equal-resolution StyledConvs at the NeRF resolution keeps its activations as split-fp16 planes
        # (csrc/chain.hip): the render kernel writes the feature map as planes, every layer of the run reads and writes them,
        # the ToRGBs in between are folded from registers, and the first up-sampling conv's low-resolution GEMM reads them.
        # In the bf16 modes the run keeps ONE bf16 plane ("planes16", cips3d_modconv1x1_planes16): the stored activation is the
        # rounded operand of the next GEMM, so the mode's results do not change while its operand bytes halve.
        use_p16 = bool(getattr(dec, "bf16", False)) and PLANES16_RUN
        if (use_split and PLANES_RUN) or use_p16:
            hw = img_size * img_size
            mine = (lambda c: c["packed"] and not c.get("chained")) if use_p16 else (lambda c: c.get("split"))  # noqa: E731
            ok_conv = lambda c: (c is not None and mine(c) and c["kind"] in (0, 1) and c["H"] == img_size   # noqa: E731
                                 and bool(lib.cips3d_planes_supported(c["Cin"], c["Cout"], hw)))
            i, n_fold = 0, 0
            while i < len(layer_info) and ok_conv(layer_info[i]):
                li = layer_info[i]
                if li["kind"] == 1:
                    if fused_stage(i):
                        li["planes_in"], li["p16"] = True, use_p16
                    break
                nxt = layer_info[i + 1] if i + 1 < len(layer_info) else None
                has_rgb = nxt is not None and nxt["kind"] == 2
                foldable = (has_rgb and i + 1 != len(layer_info) - 1 and nxt["Cin"] == li["Cout"] and hw % 4 == 0 and n_fold < 8)
                j = i + (2 if has_rgb else 1)
                nconv = layer_info[j] if j < len(layer_info) else None
                next_takes = ok_conv(nconv) and (nconv["kind"] == 0 or fused_stage(j))
                li["planes_in"], li["p16"] = True, use_p16
                li["planes_out"] = bool(next_takes and (not has_rgb or foldable))
                if has_rgb and foldable:
                    n_fold += 1
                if not li["planes_out"]:
                    break
                i = j
        wm_buf = torch.empty(sum(wm_sizes), device=dev)
        wm_tab = (_lib.ModulateDesc * len(seq))()
        rows, woff = 0, 0
        # Range workspace of a split-fp16 decoder (cips3d_range): per StyledConv an amax array, the exponent of its planes
        # output and its bound constants; the feature map's rows; one scratch amax array.  One buffer, zeroed by every forward.
        AF = hip.amax_floats()
        nblk = (img_size * img_size + _lib.PLANES_EXP_BLOCK - 1) // _lib.PLANES_EXP_BLOCK      # planes exponents per sample
        # Layout: [per StyledConv: amax | aexp] [feature-map rows] -- what every forward measures anew, the "volatile" words -- then
        # [per StyledConv: lconst], which the modulate table writes: a frame of a sequence (io.styles_resident) zeroes only the former
        per_layer = B * (AF + nblk)
        n_sc = sum(1 for li in layer_info if li["kind"] in (0, 1))
        volatile = n_sc * per_layer + B * (2 * AF + nblk)
        range_ws = torch.zeros(volatile + n_sc * B * 4, device=dev) if use_split else None
        self.ranged = range_ws is not None
        rw = range_ws.data_ptr() if self.ranged else 0
        if self.ranged:
            p.range_ws, p.range_ws_words, p.range_volatile_words = rw, range_ws.numel(), volatile
            tail = rw + 4 * n_sc * per_layer
            p.feat_amax, p.tmp_amax, p.feat_exp = tail, tail + 4 * B * AF, tail + 4 * 2 * B * AF
            # patch maxima of the planes tensors (one array per layer of the run + the feature map's for the conversion pass):
            # fully rewritten by their producer every forward, never zeroed
            n_half = (img_size * img_size + 63) // 64
            pm_words = B * n_half * (max(li["Cout"] for li in layer_info if li["kind"] in (0, 1)) // 16 + 1)
            n_pl = sum(1 for li in layer_info if li.get("planes_out"))
            pmax_ws = torch.zeros((n_pl + 1) * pm_words, device=dev)
            p.feat_pmax = pmax_ws.data_ptr() + 4 * n_pl * pm_words
            self._keep += [range_ws, pmax_ws]
        pl_i = 0
        sc_i = 0
        for idx, (info, off) in enumerate(zip(layer_info, offs)):
            conv = info["conv"]
            d = wm_tab[idx]
            d.W = dev_ptr(conv.weight)
            d.s = s_buf.data_ptr() + 4 * off
            d.out = wm_buf.data_ptr() + 4 * woff
            d.s_stride = total
            d.Cout, d.Cin, d.ksq = conv.out_channel, conv.in_channel, 1
            d.flags = (hip.MOD_DEMODULATE if conv.demodulate else 0) | (hip.MOD_PACKED if info["packed"] else 0) | (
                hip.MOD_CHAINED if info.get("chained") else 0) | (hip.MOD_SPLIT if info.get("split") else 0) | (
                hip.MOD_SPLIT16 if info.get("split16") else 0) | (hip.MOD_BF16 if info.get("p16") else 0)
            d.scale = conv.scale
            d.row_begin = rows
            rows += conv.out_channel
            L = p.layers[idx]
            L.kind, L.Cin, L.Cout, L.H, L.W, L.noise_index = (info["kind"], info["Cin"], info["Cout"], info["H"],
                                                              info["W"], info["noise_index"])
            L.flags = ((1 if info.get("chained") else 0) | (2 if info.get("split") else 0) | (4 if info.get("planes_in") else 0) |
                       (8 if info.get("planes_out") else 0) | (16 if info.get("split16") else 0) |
                       (32 if info.get("p16") else 0) | (64 if conv.demodulate else 0) | (128 if info.get("flat_head") else 0))
            L.wm = d.out
            L.bias = dev_ptr(info["bias"])
            L.noise_w = dev_ptr(info["noise_w"], allow_none=True)
            L.fir = dev_ptr(info["fir"], allow_none=True)
            if self.ranged and info["kind"] in (0, 1):
                base = rw + 4 * sc_i * per_layer
                L.amax, L.aexp, L.lconst = base, base + 4 * B * AF, rw + 4 * (volatile + sc_i * B * 4)
                d.lconst, d.bias, d.n_bias = L.lconst, L.bias, conv.out_channel
                d.noise_w, d.fir = L.noise_w, L.fir
                sc_i += 1
                if info.get("planes_out"):
                    L.pmax = pmax_ws.data_ptr() + 4 * pl_i * pm_words
                    pl_i += 1
            woff += wm_sizes[idx]
        p.n_dec_layers = len(seq)
        wm_tab_dev = _upload(wm_tab).to(dev)
        p.wm_table, p.wm_n, p.wm_rows = wm_tab_dev.data_ptr(), len(seq), rows

        # ---- renderer
        packed, layer_bias = ren._derived_buffers()
        net = ren.network
        if n_chunks is None:
            n_chunks = hip.nerf_suggest_chunks(B, img_size, N_samples)
        R = img_size * img_size
        part = torch.empty(n_chunks, B, H + 8, R, device=dev)
        features = torch.empty(B, H, img_size, img_size, device=dev)
        n = p.nerf
        n.w_first, n.packed, n.w_view = dev_ptr(net.pts_linears[0].weight), dev_ptr(packed), dev_ptr(net.views_linears.weight)
        packed32 = ren.packed32()            # "fp32_exact": the render kernel's exact-fp32 instantiation reads this stream instead
        n.packed32 = dev_ptr(packed32, "packed32", True)
        n.film, n.layer_bias = dev_ptr(film), dev_ptr(layer_bias)
        n.w_sigma, n.w_rgb = dev_ptr(net.sigma_linear.weight), dev_ptr(net.rgb_linear.weight)
        n.b_sigma, n.b_rgb = dev_ptr(net.sigma_linear.bias), dev_ptr(net.rgb_linear.bias)
        n.sigmoid_beta = dev_ptr(ren.sigmoid_beta)
        n.raw_density = int(not ren.with_sdf)
        n.B, n.img_size, n.n_samples, n.hidden, n.depth = B, img_size, N_samples, H, D
        n.static_viewdirs, n.n_chunks = int(self.static), n_chunks
        n.part = part.data_ptr()
        p.features = features.data_ptr()
        # does this shape's render launch combine its chunk partials itself?  (then the mask can leave it as two planar maps)
        probe = _lib.NerfParams.from_buffer_copy(n)
        probe.o_features = probe.o_thumb = probe.o_xyz = probe.o_mask = features.data_ptr()
        self.mask_planar = bool(_lib.load().cips3d_nerf_fuses_finish(C.byref(probe)))

        # ---- decoder workspace
        act = torch.empty(2, max_act, device=dev)
        y_lo = torch.empty(2 if any(li.get("chained") for li in layer_info) else 1, max_lo, device=dev)
        skip = torch.empty(2, max_skip, device=dev)
        p.act[0], p.act[1] = act[0].data_ptr(), act[1].data_ptr()
        p.y_lo = y_lo[0].data_ptr()
        p.y_lo2 = y_lo[1].data_ptr() if y_lo.shape[0] > 1 else None
        p.skip[0], p.skip[1] = skip[0].data_ptr(), skip[1].data_ptr()
        # ToRGB layers at the input resolution fold into the epilogue of the conv that feeds them (forward.hip): one slot
        # of [B,3,S,S] per row block per layer
        n_fold = sum(1 for i, li in enumerate(layer_info) if li["kind"] == 2 and i > 0 and layer_info[i - 1]["kind"] == 0
                     and li["H"] == img_size)
        slots = 16 * n_fold
        rgb_part = torch.empty(max(1, slots) * B * 3 * img_size * img_size, device=dev)
        p.rgb_part, p.rgb_part_slots = rgb_part.data_ptr(), slots
        self._keep.append(rgb_part)

        self.plan = p
        self.key = self.weights_key(G)
        self._keep += [packed32, lat, film, film_tab, mod_tab, s_buf, wm_buf, wm_tab_dev, packed, layer_bias, part, features, act,
                       y_lo, skip]
        self.noise_total = sum(s * s for s in self.noise_sizes)
        self.film, self.s_buf, self.range_ws = film, s_buf, range_ws

    def style_phase(self, z_r, z_d, mode=-1, trunc_psi=1.0, mean_r=None, mean_d=None, rng=None):
        """cips3d_style_phase alone (tests, tools): the mapping networks and every style head of this plan for (z_r, z_d);
        mode 0 = as launches, 1 = one launch, -1 = the forward's choice.  rng = (seed, base, normal, uniform): the draw that
        rides on it.  Results land in styles_r / styles_d / film / s_buf of the plan."""
        lib = _lib.load()
        io = ForwardIO()
        io.z_r, io.z_d = dev_ptr(z_r, "z_r"), dev_ptr(z_d, "z_d")
        io.mean_r = dev_ptr(mean_r, "mean_r", allow_none=True)
        io.mean_d = dev_ptr(mean_d, "mean_d", allow_none=True)
        io.trunc_psi = float(trunc_psi)
        if rng is not None:
            seed, base, normal, uniform = rng
            io.rng_seed, io.rng_base = seed, base
            if normal is not None:
                io.rng_normal, io.rng_n_normal = normal.data_ptr(), normal.numel()
            if uniform is not None:
                io.rng_uniform, io.rng_n_uniform = uniform.data_ptr(), uniform.numel()
        _lib.check(lib.cips3d_style_phase(C.byref(self.plan), C.byref(io), int(mode), _lib.stream_ptr()), "cips3d_style_phase")

    def __deepcopy__(self, memo):
        """A copied generator owns new parameter storage: the copy of a plan is a tombstone whose key never matches,
        so `Generator._forward_plan` rebuilds it (raw pointers must not be cloned)."""
        dead = object.__new__(ForwardPlan)
        dead.key = None
        return dead

    @staticmethod
    def weights_key(G):
        """Plans hold raw pointers: they die with any re-allocation of a parameter (.to(), load of new storage)
        and with a change of the NeRF weights the packed copy was made from."""
        ren = G.renderer
        return (G.style[0].weight.data_ptr(), G.decoder.conv1.conv.weight.data_ptr(), ren._weights_key(),
                bool(getattr(G.decoder, "bf16", False)), bool(getattr(G.decoder, "bf16_storage", False)),
                bool(getattr(G.decoder, "split", False)), bool(ren.with_sdf), bool(getattr(ren, "exact_fp32", False)))

    def _noise_bound(self, noise_bufs, fresh_noise):
        """Upper bound of |noise| over the call (cips3d_forward_io.noise_bound; the bound constants of a ranged plan): known
        for fresh draws; caller-supplied maps are measured on the device once per (storage, version)."""
        if not getattr(self, "ranged", False):
            return 0.0
        if fresh_noise:
            return hip.NOISE_BOUND_RNG if hip.FAST_RNG else hip.NOISE_BOUND_TORCH
        cache = self.__dict__.setdefault("_nb_cache", {})
        bound = hip.NOISE_BOUND_TORCH if any(nb is None for nb in noise_bufs) else 0.0
        todo = []
        for nb in noise_bufs:
            if nb is None:
                continue
            # maps under optimisation (leaf tensors that require gradients: `optim_noise_bufs`) change between calls through
            # whatever the optimiser does to their storage -- raw-pointer kernels included: measured every call, never cached
            key = None if nb.requires_grad else (nb.data_ptr(), nb._version, nb.numel())
            if key is not None and key in cache:
                bound = max(bound, cache[key])
            else:
                todo.append((key, hip.absmax(nb.detach().float().contiguous(), B=1)))
        for key, am in todo:        # (one synchronisation for all of them)
            v = float(hip.amax_value(am)[0])
            if key is not None:
                if len(cache) > 256:
                    cache.clear()
                cache[key] = v
            bound = max(bound, v)
        return bound

    def run(self, z_r, z_d, cam_poses, focals, near, far, perturb_u, noise_bufs, trunc_psi, mean_r, mean_d, return_sdf,
            events=None, fresh_perturb=False, marks=None, styles_resident=False, style_stamp=None, rgb_out=None,
            style_refs=None, views_in_flight=1):
        """fresh_perturb: draw the per-ray jitter here (perturb_u must be None) -- together with the decoder's fresh noise in
        one cips3d_rng_fill launch when both are fresh.
        styles_resident: a frame of a sequence (cips3d_forward_io.styles_resident): the style phase and the modulate table are
        skipped and the plan's tables are used as the last FULL run left them.  `style_stamp` identifies what those tables were
        computed from (latents / styles, truncation, means): a full run records it (with the call's bound of |noise|, which the
        range constants carry), a resident run must present the same one or it raises.  `style_refs`: the tensors the stamp names;
        a full run keeps them alive until the next one, so that the addresses in the stamp cannot be recycled by the allocator.
        rgb_out: a contiguous tensor [B, 3, R, R] the image is written into instead of a fresh one -- float32, or uint8 on a plan
        that is `u8_capable` (the last up-sampling stage then stores cips3d_rgb_to_uint8 of the image directly)."""
        lib = _lib.load()
        B, S, dev = self.B, self.img_size, self.device
        fresh_noise = noise_bufs is None or all(nb is None for nb in noise_bufs)
        keep = None
        if fresh_perturb:
            if perturb_u is not None:
                raise RuntimeError("fresh_perturb with an explicit perturb_u")
            if not hip.FAST_RNG:
                perturb_u = torch.rand(B, S * S, device=dev)
        io = ForwardIO()
        io.noise_bound = self._noise_bound(noise_bufs, fresh_noise)
        stamp = None if style_stamp is None else (style_stamp, float(io.noise_bound))
        if styles_resident:
            # (the FiLM / modulation tables belong to the modules and are shared by every plan of this batch size: nothing may
            # have rewritten them since -- hip.STYLE_EPOCH)
            if stamp is None or (stamp, hip.STYLE_EPOCHS[self.lane]) != getattr(self, "_resident_stamp", None):
                raise RuntimeError("styles_resident=True, but this plan's style tables were not computed from these latents / "
                                   "truncation / noise bound by the last full forward (run one frame without styles_resident first)")
            io.styles_resident = 1
        else:
            hip.STYLE_EPOCHS[self.lane] += 1
            self._resident_stamp = (stamp, hip.STYLE_EPOCHS[self.lane])
            self._resident_refs = style_refs
        if hip.FAST_RNG and (fresh_noise or fresh_perturb):
            # the draw is made by the forward call itself (cips3d_forward_io.rng_*: spread over the mapping launches)
            n_n, n_u = (B * self.noise_total if fresh_noise else 0), (B * S * S if fresh_perturb else 0)
            io.rng_seed, io.rng_base = hip.rng_reserve(n_n, n_u, dev)
            if n_n:
                keep = torch.empty(n_n, device=dev)
                io.rng_normal, io.rng_n_normal = keep.data_ptr(), n_n
            if n_u:
                perturb_u = torch.empty(B, S * S, device=dev)
                io.rng_uniform, io.rng_n_uniform = perturb_u.data_ptr(), n_u
        io.z_r = dev_ptr(z_r, "zs[0]", allow_none=True)
        io.z_d = dev_ptr(z_d, "zs[1]", allow_none=True)
        io.mean_r = dev_ptr(mean_r, "style_render_mean", allow_none=True)
        io.mean_d = dev_ptr(mean_d, "style_decoder_mean", allow_none=True)
        io.trunc_psi = float(trunc_psi)
        io.cam_poses, io.focals = dev_ptr(cam_poses, "cam_poses"), dev_ptr(focals, "focals")
        io.near_, io.far_ = dev_ptr(near, "near"), dev_ptr(far, "far")
        io.perturb_u = dev_ptr(perturb_u, "perturb_u", allow_none=True)
        R = S * S
        sdf = torch.empty(B, R, self.N, device=dev) if return_sdf else None
        io.sdf = dev_ptr(sdf, "sdf", allow_none=True)
        if fresh_noise:
            # fresh N(0,1) noise for every layer and sample: one generator launch for the whole decoder
            if keep is None:
                keep = torch.randn(B * self.noise_total, device=dev)
            off = 0
            for i, s in enumerate(self.noise_sizes):
                io.noise[i] = keep.data_ptr() + 4 * off
                io.noise_bstride[i] = s * s
                off += B * s * s
        else:
            if len(noise_bufs) != len(self.noise_sizes):
                raise RuntimeError(f"expected {len(self.noise_sizes)} noise buffers, got {len(noise_bufs)}")
            keep = []                 # fresh per-layer draws must outlive the enqueue below (only raw pointers go into io)
            for i, (nb, s) in enumerate(zip(noise_bufs, self.noise_sizes)):
                if nb is None:
                    nb = torch.randn(B, 1, s, s, device=dev)
                    keep.append(nb)
                if tuple(nb.shape[-2:]) != (s, s) or nb.shape[0] not in (1, B):
                    raise RuntimeError(f"noise buffer {i} has shape {tuple(nb.shape)}, expected (1|{B},1,{s},{s})")
                io.noise[i] = dev_ptr(nb, f"noise_bufs[{i}]")
                io.noise_bstride[i] = s * s if (nb.shape[0] == B and B > 1) else 0
        if rgb_out is None:
            rgb = torch.empty(B, 3, self.out_res, self.out_res, device=dev)
        else:
            rgb = rgb_out
            if (tuple(rgb.shape) != (B, 3, self.out_res, self.out_res) or not rgb.is_contiguous() or rgb.device != dev or
                    rgb.dtype not in (torch.float32, torch.uint8)):
                raise RuntimeError(f"rgb_out must be a contiguous float32 / uint8 tensor of shape {(B, 3, self.out_res, self.out_res)} on {dev}")
            if rgb.dtype == torch.uint8:
                if not self.u8_capable:
                    raise RuntimeError("this decoder does not end in a fused stage: no uint8 output (render float32, then hip.rgb_to_uint8)")
                io.rgb_is_u8 = 1
        thumb = torch.empty(B, 3, S, S, device=dev)
        xyz = torch.empty(B, 3, S, S, device=dev)
        # `mask` is returned as [2, B, S, S]: the background-weight map of every view, then the depth map of every view (the two
        # tensors Generator.forward hands out).  A render launch that fuses its finish writes that layout itself.
        mask = torch.empty((2, B, S, S) if self.mask_planar else (B, 2, S, S), device=dev)
        io.mask_planar = int(self.mask_planar)
        io.views_in_flight = int(views_in_flight)      # (a scheduling hint: pipeline.ViewPipeline; results do not depend on it)
        io.rgb, io.thumb, io.xyz, io.mask = rgb.data_ptr(), thumb.data_ptr(), xyz.data_ptr(), mask.data_ptr()
        if events is not None:
            io.ev_nerf_start, io.ev_nerf_stop = events
        if marks is not None:        # decoder timeline (bench.py): (handle array, info array, count) of hip.DecoderMarks
            io.ev_marks, io.ev_info, io.ev_count, io.n_ev_marks = marks
        _lib.check(lib.cips3d_generator_forward(C.byref(self.plan), C.byref(io), _lib.stream_ptr()),
                   "cips3d_generator_forward")
        if sdf is not None:
            sdf = sdf.view(B, S, S, self.N, 1)
        if not self.mask_planar:
            mask = mask.transpose(0, 1).contiguous()
        return rgb, thumb, xyz, mask, sdf

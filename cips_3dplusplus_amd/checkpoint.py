"""On-disk formats either side of the path.

* Model checkpoint directory as written by the reference trainer (`save_models`,
  /root/reference/exp/cips3d/scripts/train_v10.py:496-523): `config_command.yaml` — one top-level key
  (the launch command) whose value holds `G_cfg` / `G_kwargs` — and one `<name>.pth` per model holding its
  `state_dict()` (`G_ema.pth` is what every inference caller loads: exp/tests/test_cips3dpp.py:705-707,
  render_video_web_v10.py:1718-1722).
* Inversion result file written by `StyleGAN2Projector_Flip.project_wplus`
  (/root/reference/exp/cips3d/models/projector_v10.py:1044-1055,1262-1263) and read back by the demo pages
  (render_video_web_v10.py:2556-2574): a dict with `azim`, `elev`, `w_render_opt`, `w_decoder_opt`,
  `render_state_dict`, `decoder_state_dict`, `noise_bufs`, `padding`.

The yaml is read with PyYAML (`tl2`'s `TLCfgNode` is a yacs-style wrapper over the same mapping); tensors are
read with `torch.load(map_location="cpu")` and moved to the device by the module they are loaded into.
"""
import os

import torch
import yaml

_NON_CTOR_KEYS = ("register_modules", "name")


def load_config_command(path):
    """`list(TLCfgNode.load_yaml_file(path).values())[0]` (test_cips3dpp.py:705): the first command block."""
    with open(path) as f:
        doc = yaml.safe_load(f)
    if not isinstance(doc, dict) or not doc:
        raise ValueError(f"{path}: expected a mapping with the launch command as its single top-level key")
    cfg = next(iter(doc.values()))
    if "G_cfg" not in cfg:
        raise KeyError(f"{path}: no G_cfg under '{next(iter(doc))}'")
    return cfg


def generator_ctor_cfg(G_cfg):
    """G_cfg minus the registry keys `build_model` consumes (train_cips3d_ffhq_v10.yaml:90-93)."""
    return {k: v for k, v in G_cfg.items() if k not in _NON_CTOR_KEYS}


def _torch_load(path):
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except TypeError:       # older torch without weights_only
        return torch.load(path, map_location="cpu")


def load_generator(ckpt_dir, device="cuda", model="G_ema", strict=True):
    """build_model(loaded_cfg.G_cfg) + Checkpointer(G).load_state_dict_from_file(f"{ckpt_dir}/G_ema.pth").
    Returns (generator on `device`, the loaded command config)."""
    from .generator import Generator
    cfg = load_config_command(os.path.join(ckpt_dir, "config_command.yaml"))
    G = Generator(**generator_ctor_cfg(cfg["G_cfg"])).eval().requires_grad_(False)     # an inference / inversion handle, as build_generator
    sd = _torch_load(os.path.join(ckpt_dir, f"{model}.pth"))
    missing, unexpected = G.load_state_dict(sd, strict=strict)
    if not strict and (missing or unexpected):
        import warnings
        warnings.warn(f"load_generator: missing {list(missing)[:4]}… unexpected {list(unexpected)[:4]}…")
    return G.to(device), cfg


def save_generator(G, ckpt_dir, G_cfg, G_kwargs=None, command="train_cips3d_ffhq_v10", model="G_ema",
                   name="exp.cips3d.models.model_v3.Generator"):
    """Write the same two files the reference trainer writes for `model` (the other models of its dict — G, D,
    D_render, optimiser state — are training state and out of scope)."""
    os.makedirs(ckpt_dir, exist_ok=True)
    block = {"G_cfg": {"register_modules": [name.rsplit(".", 1)[0]], "name": name, **G_cfg}}
    if G_kwargs is not None:
        block["G_kwargs"] = G_kwargs
    with open(os.path.join(ckpt_dir, "config_command.yaml"), "w") as f:
        yaml.safe_dump({command: block}, f, sort_keys=False)
    torch.save({k: v.detach().cpu() for k, v in G.state_dict().items()}, os.path.join(ckpt_dir, f"{model}.pth"))
    return ckpt_dir


# ------------------------------------------------------------------------------------------ inversion results
INVERSION_KEYS = ("azim", "elev", "w_render_opt", "w_decoder_opt", "render_state_dict", "decoder_state_dict",
                  "noise_bufs", "padding")


def save_inversion(path, azim, elev, w_render_opt, w_decoder_opt, G, noise_bufs=None, padding=None):
    """projector_v10.py:1044-1055 + :1262-1263."""
    cpu = lambda t: t.detach().cpu()
    obj = {
        "azim": cpu(azim), "elev": cpu(elev), "w_render_opt": cpu(w_render_opt), "w_decoder_opt": cpu(w_decoder_opt),
        "render_state_dict": {k: cpu(v) for k, v in G.renderer.state_dict().items()},
        "decoder_state_dict": {k: cpu(v) for k, v in G.decoder.state_dict().items()},
        "noise_bufs": None if noise_bufs is None else [cpu(b) for b in noise_bufs],
        "padding": padding,
    }
    torch.save(obj, path)
    return path


def load_inversion(path, w_idx=0):
    """`__load_proj_w` (render_video_web_v10.py:2556-2574): returns
    (azim, elev, w_render_opt [1,D+1,W], w_decoder_opt [1,n_latent,512], decoder_state_dict, noise_bufs,
    render_state_dict).  `w_render_opt` is always row 0 — image and flip share the NeRF style — while
    `w_decoder_opt`, `azim`, `elev` are indexed by `w_idx` (0 = image, 1 = flipped image)."""
    kw = _torch_load(path)
    azim = float(kw["azim"][w_idx])
    elev = float(kw["elev"][w_idx])
    w_render = kw["w_render_opt"].detach()[[0]]
    w_decoder = kw["w_decoder_opt"].detach()[[w_idx]]
    noise = kw.get("noise_bufs")
    if noise is not None:
        noise = [b.detach().requires_grad_(False) for b in noise]
    return azim, elev, w_render, w_decoder, kw["decoder_state_dict"], noise, kw.get("render_state_dict")


def apply_inversion(G, decoder_state_dict=None, render_state_dict=None):
    """Checkpointer(G.decoder).load_state_dict(decoder_state_dict) / Checkpointer(G.renderer)… of the demo pages
    (render_video_web_v10.py:2038-2044): sub-modules are reloaded on their own, so the `decoder.*` /
    `renderer.*` key names are part of the contract."""
    if decoder_state_dict is not None:
        G.decoder.load_state_dict(decoder_state_dict, strict=True)
    if render_state_dict is not None:
        G.renderer.load_state_dict(render_state_dict, strict=True)
    return G


def stage_keys(state_dict, stages=(0, 1, 2, 3, 4, 5, 6, 7), conv_in=False):
    """Keys of the decoder stages a stylisation page blends (render_video_web_v10.py:56-71): stage i owns
    `convs.{2i}.`, `convs.{2i+1}.`, `to_rgbs.{i}.`; `conv_in` adds `conv1` / `to_rgb1`."""
    prefixes = (["conv1", "to_rgb1"] if conv_in else []) + \
               [p for i in stages for p in (f"convs.{2 * i}.", f"convs.{2 * i + 1}.", f"to_rgbs.{i}.")]
    prefixes = tuple(prefixes)
    return [k for k in state_dict if prefixes and k.startswith(prefixes)]


def blend_decoder_stages(inverted_sd, decoder, decay, stages=(), conv_in=False):
    """In place on `inverted_sd`: `w <- decay * w + (1 - decay) * decoder.state_dict()[k]` for the selected
    stages — the `ema_accumulate(interp_state_dict, g_ema.decoder, truncation_content)` step that precedes
    loading the inverted decoder (render_video_web_v10.py:170-177; tl2's helper is StyleGAN2's `accumulate`)."""
    src = decoder.state_dict()
    for k in stage_keys(inverted_sd, stages, conv_in):
        if inverted_sd[k].is_floating_point():
            inverted_sd[k] = inverted_sd[k].float() * float(decay) + src[k].detach().cpu().float() * (1.0 - float(decay))
    return inverted_sd

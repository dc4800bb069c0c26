"""Generator / camera / NeRF configuration presets of the released v10 recipes.

Values restate /root/reference/exp/cips3d/configs/train_cips3d_ffhq_v10.yaml:90-140 and
train_cips3d_compcars_v10.yaml:96-108 (`G_cfg`, `G_kwargs.cam_cfg`, `G_kwargs.nerf_cfg`);
the per-release overrides (NeRF depth, upsample_list) are the ones the release scripts pass
(bash/train_cips3d_ffhq_v10/train_r1024_r64_ks1.sh:22-23).
"""
import copy

_UPSAMPLE = {64: [], 128: [128], 256: [128, 256], 512: [128, 256, 512], 1024: [128, 256, 512, 1024]}


def ffhq_G_cfg(resolution=1024, N_layers_renderer=2, kernel_size=1):
    return {
        "enable_decoder": True,
        "freeze_renderer": False,
        "renderer_detach": True,
        "predict_rgb_residual": False,
        "scale_factor": 1,
        "renderer_cfg": {"N_layers_renderer": N_layers_renderer, "input_dim": 3, "hidden_dim": 256,
                         "view_dim": 3, "with_sdf": True, "output_features": True},
        "mapping_renderer_cfg": {"z_dim": 256, "style_dim": 256, "N_layers": 3},
        "decoder_cfg": {"size_start": 4, "size_end": 1024, "in_channel": 256, "channel_multiplier": 2,
                        "project_noise": False, "upsample_list": list(_UPSAMPLE[resolution]),
                        "kernel_size": kernel_size},
        "mapping_decoder_cfg": {"style_dim": 512, "lr_mul_mapping": 0.01, "N_layers": 5},
    }


def tiny_G_cfg(hidden=32, N_layers_renderer=2, kernel_size=1):
    """Smallest generator that still walks every code path (plain/up conv, blur, skip
    upsample, noise, toRGB): SURVEY.md Appendix D, with hidden_dim a multiple of 32."""
    return {
        "enable_decoder": True, "freeze_renderer": False, "renderer_detach": True,
        "predict_rgb_residual": False, "scale_factor": 1,
        "renderer_cfg": {"N_layers_renderer": N_layers_renderer, "input_dim": 3, "hidden_dim": hidden,
                         "view_dim": 3, "with_sdf": True, "output_features": True},
        "mapping_renderer_cfg": {"z_dim": hidden, "style_dim": hidden, "N_layers": 3},
        "decoder_cfg": {"size_start": 256, "size_end": 1024, "in_channel": hidden, "channel_multiplier": 2,
                        "project_noise": False, "upsample_list": [512, 1024], "kernel_size": kernel_size},
        "mapping_decoder_cfg": {"style_dim": 32, "lr_mul_mapping": 0.01, "N_layers": 5},
    }


FFHQ_CAM_CFG = {"img_size": 64, "uniform": False, "azim_range": 0.3, "elev_range": 0.15,
                "fov_ang": 6, "dist_radius": 0.12}
COMPCARS_CAM_CFG = {"img_size": 64, "uniform": True, "azim_range": [-3.14, 3.14],
                    "elev_range": [0.0, 0.1674], "fov_ang": 15, "dist_radius": 0.3}
TRAIN_NERF_CFG = {"N_samples": 24, "perturb": True, "static_viewdirs": False}
DEMO_NERF_CFG = {"N_samples": 128, "perturb": False, "static_viewdirs": True}


def clone(cfg):
    return copy.deepcopy(cfg)

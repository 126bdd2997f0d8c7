"""GPU: `python -m mikudance_amd.inference_video` -- the drop-in for the reference's scripts/inference_video.py -- end to end on a
synthetic `pretrained_weights/` tree (no checkpoints exist offline): YAML config -> AutoencoderKL / UNet2DConditionModel donor /
from_unet / from_pretrained_2d / CLIPVisionModelWithProjection loaders -> DDIMScheduler(**noise_scheduler_kwargs) ->
load_state_dict of the stage-2 files -> read_frames / depth resize / camera_to_scene_motion -> pipe(...) -> save_videos_grid.
Every component is this package's (HIP kernels underneath).  The UNets have the real SD-1.5 width because `from_unet` builds the
default geometry whatever the donor is (quirk 9); VAE and CLIP geometries come from their config.json and are reduced."""
import json
import math
import os

import numpy as np
import pytest
import torch
from PIL import Image
from safetensors.torch import save_file

pytestmark = pytest.mark.gpu

import mikudance_amd as M  # noqa: E402
from mikudance_amd import io_utils as U  # noqa: E402
from mikudance_amd.selftest import MM_KWARGS, SCHED_KWARGS  # noqa: E402
from mikudance_amd.synth import _is_norm_weight, positional_encoding_table  # noqa: E402


def cheap_state_dict(module_ctor, seed):
    """fp16 weights for a whole model in seconds: uniform(-a, a), a = sqrt(3 / fan_in) (unit-variance propagation), norm weights 1,
    biases 0, analytic positional encodings."""
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in module_ctor().state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in shapes.items():
        if k.endswith("pos_encoder.pe"):
            out[k] = positional_encoding_table(shp[2], shp[1]).half()
        elif _is_norm_weight(k):
            out[k] = torch.ones(shp, dtype=torch.float16)
        elif k.endswith(".bias") or k.endswith("mix_factor"):
            out[k] = torch.zeros(shp, dtype=torch.float16)
        else:
            fan_in = max(1, int(np.prod(shp[1:])))
            out[k] = torch.empty(shp, dtype=torch.float16).uniform_(-math.sqrt(3.0 / fan_in), math.sqrt(3.0 / fan_in), generator=g)
    return out


def test_inference_video_script_end_to_end(tmp_path, golden_dir):
    root = tmp_path / "pretrained_weights"
    sd15, vae_d, enc_d = root / "stable-diffusion-v1-5" / "unet", root / "sd-vae-ft-mse", root / "image_encoder"
    for d in (sd15, vae_d, enc_d, tmp_path / "configs", tmp_path / "inputs"):
        os.makedirs(d)
    unet_cfg = {"_class_name": "UNet2DConditionModel", "act_fn": "silu", "attention_head_dim": 8, "block_out_channels": [320, 640, 1280, 1280],
                "center_input_sample": False, "cross_attention_dim": 768, "downsample_padding": 1, "flip_sin_to_cos": True, "freq_shift": 0,
                "in_channels": 4, "layers_per_block": 2, "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": 32,
                "out_channels": 4, "sample_size": 64,
                "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
                "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"]}
    json.dump(unet_cfg, open(sd15 / "config.json", "w"))
    den_sd = cheap_state_dict(lambda: M.UNet3DConditionModel(sample_size=64, cross_attention_dim=768, **MM_KWARGS), 1)
    save_file({k: v for k, v in den_sd.items() if "motion_modules" not in k}, str(sd15 / "diffusion_pytorch_model.safetensors"))
    torch.save({k: v for k, v in den_sd.items() if "motion_modules" in k}, root / "motion_module.pth")
    torch.save({k: v for k, v in den_sd.items() if k.startswith("conv_in.")}, root / "denoising_unet.pth")        # strict=False
    del den_sd
    torch.save(cheap_state_dict(lambda: M.UNet2DConditionModel(cross_attention_dim=768), 2), root / "reference_unet.pth")
    vae_cfg = {"_class_name": "AutoencoderKL", "in_channels": 3, "out_channels": 3, "block_out_channels": [64, 64, 128, 128], "latent_channels": 4,
               "layers_per_block": 2, "norm_num_groups": 32, "act_fn": "silu", "scaling_factor": 0.18215, "sample_size": 256}
    json.dump(vae_cfg, open(vae_d / "config.json", "w"))
    save_file(cheap_state_dict(lambda: M.AutoencoderKL(**vae_cfg), 3), str(vae_d / "diffusion_pytorch_model.safetensors"))
    clip_cfg = {"hidden_size": 128, "intermediate_size": 256, "num_hidden_layers": 2, "num_attention_heads": 4, "image_size": 224, "patch_size": 56,
                "projection_dim": 768, "hidden_act": "quick_gelu", "model_type": "clip_vision_model"}
    json.dump(clip_cfg, open(enc_d / "config.json", "w"))
    save_file(cheap_state_dict(lambda: M.CLIPVisionModelWithProjection(clip_cfg), 4), str(enc_d / "model.safetensors"))

    rng = np.random.default_rng(0)
    F_, W, H = 2, 64, 64
    img = lambda: Image.fromarray(np.kron(rng.integers(0, 255, (10, 9, 3), dtype=np.uint8), np.ones((8, 8, 1), dtype=np.uint8)))
    img().save(tmp_path / "inputs" / "img-char.jpg")
    img().save(tmp_path / "inputs" / "skel-img-char.jpg")
    U.save_videos_from_pil([img() for _ in range(F_)], str(tmp_path / "inputs" / "pose-demo.mp4"), fps=12)
    U.save_videos_from_pil([img() for _ in range(F_)], str(tmp_path / "inputs" / "hand-demo.gif"), fps=12)
    z = np.load(os.path.join(golden_dir, "g2_scene_motion.npz"))
    np.save(tmp_path / "inputs" / "w2c.npy", z["w2c"][:F_])
    np.save(tmp_path / "inputs" / "c2w.npy", z["c2w"][:F_])
    np.save(tmp_path / "inputs" / "depth.npy", np.kron(z["depth"], np.ones((1, 4, 4))))
    import yaml
    yaml.safe_dump({"unet_additional_kwargs": MM_KWARGS, "noise_scheduler_kwargs": SCHED_KWARGS, "sampler": "DDIM"},
                   open(tmp_path / "configs" / "mikudance_config.yaml", "w"))
    yaml.safe_dump({"pretrained_base_model_path": str(root / "stable-diffusion-v1-5"), "pretrained_vae_path": str(vae_d),
                    "pretrained_temporal_vae_path": str(root / "vae_temporal_decoder"), "image_encoder_path": str(enc_d),
                    "denoising_unet_path": str(root / "denoising_unet.pth"), "reference_unet_path": str(root / "reference_unet.pth"),
                    "motion_module_path": str(root / "motion_module.pth"), "inference_config": str(tmp_path / "configs" / "mikudance_config.yaml"),
                    "weight_dtype": "fp16", "ref_image_path": str(tmp_path / "inputs" / "img-char.jpg"),
                    "ref_skel_path": str(tmp_path / "inputs" / "skel-img-char.jpg"), "ref_depth_path": str(tmp_path / "inputs" / "depth.npy"),
                    "tgt_pose_path": str(tmp_path / "inputs" / "pose-demo.mp4"), "tgt_face_path": "None",
                    "tgt_hand_path": str(tmp_path / "inputs" / "hand-demo.gif"), "tgt_w2c_path": str(tmp_path / "inputs" / "w2c.npy"),
                    "tgt_c2w_path": str(tmp_path / "inputs" / "c2w.npy")}, open(tmp_path / "configs" / "inference_video.yaml", "w"))

    from mikudance_amd import inference_video
    out = inference_video.main(["--config", str(tmp_path / "configs" / "inference_video.yaml"), "-W", str(W), "-H", str(H), "--steps", "2",
                                "--seed", "7", "--output_dir", str(tmp_path / "output")])
    assert os.path.basename(out).startswith(f"pose-demo_img-char_{H}x{W}_3_") and out.endswith(".mp4")
    frames = U.read_frames(out)
    assert len(frames) == F_ and frames[0].size == (3 * (W + 2) + 2, H + 4) and U.get_fps(out) == 12
    a = np.asarray(frames[0], dtype=np.float32)
    assert np.isfinite(a).all() and a[:, 2 * (W + 2):].std() > 0                     # third grid cell = the generated frame
    # --video_decoder: the temporal-decoder VAE (config.pretrained_temporal_vae_path), frames decoded in chunks with num_frames
    tv = root / "vae_temporal_decoder"
    os.makedirs(tv)
    tcfg = {"_class_name": "AutoencoderKLTemporalDecoder", "in_channels": 3, "out_channels": 3, "block_out_channels": [64, 64, 128, 128],
            "latent_channels": 4, "layers_per_block": 2, "sample_size": 768, "scaling_factor": 0.18215, "force_upcast": True}
    json.dump(tcfg, open(tv / "config.json", "w"))
    save_file(cheap_state_dict(lambda: M.AutoencoderKLTemporalDecoder(**tcfg), 5), str(tv / "diffusion_pytorch_model.safetensors"))
    out2 = inference_video.main(["--config", str(tmp_path / "configs" / "inference_video.yaml"), "-W", str(W), "-H", str(H), "--steps", "1",
                                 "--seed", "7", "--video_decoder", "--output_dir", str(tmp_path / "output_t")])
    f2 = U.read_frames(out2)
    assert len(f2) == F_ and f2[0].size == frames[0].size and np.isfinite(np.asarray(f2[0], dtype=np.float32)).all()
    # weight_dtype: fp32 (reference scripts/inference_video.py:66-69): accepted with a warning, fp32 at the module boundaries
    # (bit-identity with the fp16 run on equal inputs: tests/test_unets_gpu.py::test_fp32_typed_models_and_latents_match_fp16_run)
    cfg = yaml.safe_load(open(tmp_path / "configs" / "inference_video.yaml"))
    cfg["weight_dtype"] = "fp32"
    yaml.safe_dump(cfg, open(tmp_path / "configs" / "inference_video_fp32.yaml", "w"))
    with pytest.warns(UserWarning, match="fp32"):
        out3 = inference_video.main(["--config", str(tmp_path / "configs" / "inference_video_fp32.yaml"), "-W", str(W), "-H", str(H), "--steps", "2",
                                     "--seed", "7", "--output_dir", str(tmp_path / "output_fp32")])
    f3 = U.read_frames(out3)
    a3 = np.asarray(f3[0], dtype=np.float32)
    assert len(f3) == F_ and f3[0].size == frames[0].size and np.isfinite(a3).all() and a3[:, 2 * (W + 2):].std() > 0

"""mikudance_amd -- MI355X (gfx950) native denoising loop for MikuDance's image-to-video pipeline.

Public surface (mirrors the reference's Python API for the hot path):
    UNet3DConditionModel, UNet2DConditionModel (+ UNet2DConditionModelPlain donor), ReferenceAttentionControl,
    DDIMScheduler, MikuDanceVideoPipeline, Pose2VideoPipeline, get_context_scheduler, camera_to_scene_motion,
    AutoencoderKL, CLIPVisionModelWithProjection (the rows after the loop: VAE encode / decode and the CLIP image tower on
    the same kernels)
All compute goes through libmdance_hip.so (include/mdance_hip.h); importing the package needs neither a GPU nor the
library, calling any op does.
"""
from .clip_vision import CLIPVisionModelWithProjection, clip_preprocess  # noqa: F401
from .context import get_context_scheduler  # noqa: F401
from .mutual_mix_attention import ReferenceAttentionControl  # noqa: F401
from .pipeline_mikudance import MikuDanceVideoPipeline, MikuDanceVideoPipelineOutput  # noqa: F401
from .pipeline_stage2_vdo import Pose2VideoPipeline  # noqa: F401
from .scheduler import DDIMScheduler  # noqa: F401
from .unet_2d_mix import UNet2DConditionModel, UNet2DConditionModelPlain  # noqa: F401
from .unet_3d_mix import UNet3DConditionModel  # noqa: F401
from .vae import AutoencoderKL  # noqa: F401
from .vae_temporal import AutoencoderKLTemporalDecoder  # noqa: F401

__version__ = "0.1.0"

"""Pose2VideoPipeline -- API alias of MikuDanceVideoPipeline (reference src/pipelines/pipeline_stage2_vdo.py:36-712:
same loop; differs only in class name, context_frames=32 and the absence of the temporal-VAE switch)."""
from .pipeline_mikudance import MikuDanceVideoPipeline, MikuDanceVideoPipelineOutput


class Pose2VideoPipelineOutput(MikuDanceVideoPipelineOutput):
    pass


class Pose2VideoPipeline(MikuDanceVideoPipeline):
    default_context_frames = 32

    def __init__(self, vae, image_encoder, reference_unet, denoising_unet, scheduler, image_proj_model=None, tokenizer=None,
                 text_encoder=None):
        super().__init__(vae, image_encoder, reference_unet, denoising_unet, scheduler, image_proj_model, tokenizer,
                         text_encoder, video_decoder=False)

    def __call__(self, *args, **kwargs):
        out = super().__call__(*args, **kwargs)
        if isinstance(out, MikuDanceVideoPipelineOutput):
            return Pose2VideoPipelineOutput(videos=out.videos)
        return out

"""DDIMScheduler -- host-side mirror of diffusers==0.24.0 DDIMScheduler for the configuration MikuDance uses
(reference configs/inference/mikudance_config.yaml:24-33; constructed at scripts/inference_video.py:101-102):
linear betas rescaled to zero terminal SNR, v-prediction, trailing timestep spacing; eta = 0 (the script's default) or eta > 0.
The table is 1000 fp32 scalars on the host; the per-step arithmetic on the latents is the HIP kernel
md_cfg_ddim_step (fused with window averaging and classifier-free guidance)."""
from dataclasses import dataclass

import numpy as np
import torch


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """diffusers.utils.torch_utils.randn_tensor for one generator: the draw happens on the GENERATOR's device (a CPU generator ->
    CPU draw, then moved: what scripts/inference_video.py's torch.manual_seed generator gives) so that results do not depend
    on where the model lives."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    gdev = generator.device if generator is not None else device
    return torch.randn(tuple(shape), generator=generator, device=gdev, dtype=dtype).to(device)


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", trained_betas=None, clip_sample: bool = True, set_alpha_to_one: bool = True,
                 steps_offset: int = 0, prediction_type: str = "epsilon", thresholding: bool = False,
                 dynamic_thresholding_ratio: float = 0.995, clip_sample_range: float = 1.0, sample_max_value: float = 1.0,
                 timestep_spacing: str = "leading", rescale_betas_zero_snr: bool = False):
        if beta_schedule != "linear" or trained_betas is not None:
            raise NotImplementedError("only beta_schedule='linear' (MikuDance config) is implemented")
        if prediction_type != "v_prediction" or timestep_spacing != "trailing" or clip_sample or thresholding:
            raise NotImplementedError("only v_prediction / trailing / clip_sample=False (MikuDance config) is implemented")
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                           steps_offset=steps_offset, prediction_type=prediction_type, timestep_spacing=timestep_spacing,
                           rescale_betas_zero_snr=rescale_betas_zero_snr)
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        if rescale_betas_zero_snr:
            s = torch.cumprod(1.0 - betas, dim=0).sqrt()
            s0, sT = s[0].clone(), s[-1].clone()
            s = (s - sT) * (s0 / (s0 - sT))
            ab = s ** 2
            betas = 1 - torch.cat([ab[0:1], ab[1:] / ab[:-1]])
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_train_timesteps = num_train_timesteps
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps / num_inference_steps
        ts = np.round(np.arange(self.num_train_timesteps, 0, -ratio)).astype(np.int64) - 1
        self.timesteps = torch.from_numpy(ts)           # kept on the host: the loop reads them as Python ints

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step_coefficients(self, timestep):
        """(alpha_prod_t, alpha_prod_t_prev) as Python floats; prev_t = t - 1000 // N (integer division, literally)."""
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict: bool = True):
        """API-compatible step on GPU tensors of any shape (the fused pipeline calls md_cfg_ddim_step directly with
        window averaging and guidance folded in; this entry runs the same kernel without them)."""
        from . import ops
        if not sample.is_cuda or sample.numel() % 4:
            raise RuntimeError("DDIMScheduler.step: tensors must live on the GPU (no CPU path)")
        a_t, a_prev = self.step_coefficients(timestep)
        n = sample.numel() // 4
        lat = sample.detach().to(torch.float16).reshape(1, n, 4).contiguous().clone()
        v = model_output.detach().to(torch.float32).reshape(1, 1, n, 4).contiguous()
        one = torch.ones((1,), device=sample.device, dtype=torch.float32)
        z = None
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or"
                                 " `variance_noise` stays `None`.")
            if variance_noise is None:
                variance_noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            z = variance_noise.detach().to(device=sample.device, dtype=torch.float16).reshape(1, n, 4).contiguous()
        ops.cfg_ddim_step(lat, v, one, 1, n, 1.0, a_t, a_prev, halves=1, eta=float(eta), variance_noise=z)
        prev = lat.reshape(sample.shape).to(sample.dtype)
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev)

"""Seeded synthetic weights / inputs (there are no checkpoints and no network in the build or bench
environment).  Deterministic per KEY (order independent), so the reference modules
(in oracle/gen_golden.py), the CPU oracle and the HIP path can all be given bit-identical weights.
"""
import math
import zlib

import torch


def _is_norm_weight(key):
    parts = key.split(".")
    if parts[-1] != "weight":
        return False
    owner = parts[-2]
    if owner.startswith("norm") or owner in ("ff_norm", "conv_norm_out") or "norm" in owner:
        return True                                            # incl. CLIP's layer_norm1/2, pre_layrnorm, post_layernorm; VAE group_norm
    return len(parts) >= 3 and parts[-3] == "norms"          # motion module `norms.{0,1}.weight`


def positional_encoding_table(d_model, max_len=32):
    """The analytic sinusoid stored in `pos_encoder.pe` buffers (1, max_len, d_model)."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def synth_tensors(shapes, seed=1234, mode="fan_in"):
    """Generator form of synth_state_dict: yields (key, fp32 tensor) one at a time (every tensor is a pure function of its key, the
    seed and the mode), so a caller that casts / moves each tensor as it arrives never holds a second copy of the model."""
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        if key.endswith("pos_encoder.pe"):
            yield key, positional_encoding_table(shape[2], shape[1])
            continue
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        if _is_norm_weight(key):
            yield key, 1.0 + 0.02 * r
        elif key.endswith(".bias"):
            yield key, 0.02 * r
        elif mode == "fan_in":
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            yield key, r / math.sqrt(max(fan_in, 1))
        else:
            yield key, 0.02 * r


def synth_state_dict(shapes, seed=1234, mode="fan_in"):
    """shapes: dict key -> shape.  mode 'fan_in': weights N(0, 1/fan_in) (O(1) activations: sensitive parity
    tests); mode 'n002': N(0, 0.02^2) (SURVEY.md 8d bench weights).  Norm weights 1 + N(0, 0.02^2), biases
    N(0, 0.02^2), pos_encoder.pe analytic."""
    return dict(synth_tensors(shapes, seed=seed, mode=mode))


def synth_inputs(frames, h, w, ctx_len=257, ctx_dim=768, seed=100):
    """latents (1,4,F,h,w) from a CPU generator like the script's torch.manual_seed(seed) (quirk 11);
    ref_latents (1,F,22,h,w): 20 latent channels N(0,1)*0.18215-scale + 2 flow channels U(-0.03,0.03);
    embeds (2,L,D): row 0 zeros (uncond), row 1 N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    latents = torch.randn((1, 4, frames, h, w), generator=g)
    char = torch.randn((1, frames, 20, h, w), generator=g) * 0.18215 * 4.0
    flow = (torch.rand((1, frames, 2, h, w), generator=g) - 0.5) * 0.06
    embeds = torch.zeros((2, ctx_len, ctx_dim))
    embeds[1] = torch.randn((ctx_len, ctx_dim), generator=g)
    return latents, torch.cat([char, flow], dim=2), embeds

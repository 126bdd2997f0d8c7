"""TEST INFRASTRUCTURE ONLY -- stand-in for the un-vendored third-party `diffusers==0.24.0`.

The reference (/root/reference, requirements.txt:7) imports its leaf ops from diffusers, which is not
installed in this image and cannot be installed (no network).  This package exports exactly the
symbols the reference's hot-path modules import so that
`/root/reference/src/models/*.py` can be imported UNMODIFIED by `oracle/gen_golden.py` inside the
build container.  All reference-owned glue (UNet wiring, MAN placement, bank write/read/CFG masking,
temporal-attention PE quirk) then runs from the reference's own files; only the leaf ops below come
from this restatement of the public 0.24.0 semantics.  Nothing in the
reference's tests pins those leaf ops: PARITY UNPINNED at this boundary (stated in DESIGN.md).

Never imported by the product package `mikudance_amd`; never shipped as a dependency.
"""
import inspect
import math
import sys
import types
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

__version__ = "0.24.0-shim"


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], leaf, m)
    return m


# ------------------------------------------------------------------ utils
class BaseOutput(OrderedDict):
    """dataclass-backed ordered dict (attribute + index access), like diffusers.utils.BaseOutput."""

    def __init_subclass__(cls) -> None:
        super().__init_subclass__()

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                OrderedDict.__setitem__(self, f.name, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _Logger()

    @staticmethod
    def set_verbosity_info():
        pass

    @staticmethod
    def set_verbosity_error():
        pass


logging = _Logging()
USE_PEFT_BACKEND = False
SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
WEIGHTS_NAME = "diffusion_pytorch_model.bin"


def deprecate(*a, **k):
    return None


def scale_lora_layers(*a, **k):
    return None


def unscale_lora_layers(*a, **k):
    return None


def is_torch_version(op, ver):
    return True


def is_accelerate_available():
    return False


def is_xformers_available():
    return False


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """CPU generator + any device -> sample on CPU, then move (diffusers.utils.torch_utils)."""
    device = device or torch.device("cpu")
    gen_dev = generator.device.type if generator is not None else "cpu"
    rand_device = torch.device("cpu") if gen_dev == "cpu" else device
    return torch.randn(shape, generator=generator, device=rand_device, dtype=dtype).to(device)


def apply_freeu(resolution_idx, hidden_states, res_hidden_states, **kw):
    return hidden_states, res_hidden_states


# ------------------------------------------------------------------ config / model mixins
class _Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def register_to_config(init):
    def wrapped(self, *args, **kwargs):
        sig = inspect.signature(init)
        names = [p for p in sig.parameters if p != "self"]
        cfg = {n: sig.parameters[n].default for n in names}
        for n, a in zip(names, args):
            cfg[n] = a
        cfg.update({k: v for k, v in kwargs.items() if k in cfg})
        init(self, *args, **{k: v for k, v in kwargs.items() if k in sig.parameters})
        if not hasattr(self, "_internal_config"):
            self._internal_config = _Config()
        self._internal_config.update(cfg)

    wrapped.__wrapped__ = init
    return wrapped


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_config

    def register_to_config(self, **kw):
        if not hasattr(self, "_internal_config"):
            self._internal_config = _Config()
        self._internal_config.update(kw)

    @classmethod
    def load_config(cls, path, **kw):
        import json
        import os

        if os.path.isdir(path):
            path = os.path.join(path, cls.config_name)
        with open(path) as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config, **kwargs):
        init = getattr(cls.__init__, "__wrapped__", cls.__init__)
        names = set(inspect.signature(init).parameters) - {"self"}
        args = {k: v for k, v in dict(config).items() if k in names}
        args.update({k: v for k, v in kwargs.items() if k in names})
        return cls(**args)


class ModelMixin(nn.Module):
    _supports_gradient_checkpointing = False

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class UNet2DConditionLoadersMixin:
    pass


class DiffusionPipeline:
    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


# ------------------------------------------------------------------ leaf ops (0.24.0 semantics)
def get_activation(name):
    name = name.lower()
    if name in ("silu", "swish"):
        return nn.SiLU()
    if name == "gelu":
        return nn.GELU()
    if name == "relu":
        return nn.ReLU()
    if name == "mish":
        return nn.Mish()
    raise ValueError(name)


class LoRACompatibleConv(nn.Conv2d):
    def forward(self, x, scale=1.0):
        return super().forward(x)


class LoRACompatibleLinear(nn.Linear):
    def forward(self, x, scale=1.0):
        return super().forward(x)


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.downscale_freq_shift)
        emb = torch.exp(exponent)
        emb = timesteps[:, None].float() * emb[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None,
                 cond_proj_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.cond_proj = None
        self.act = get_activation(act_fn)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)
        self.post_act = None

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


class _NameOnly(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("name-only stub of a diffusers symbol that is off the hot path")


class AttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale=1.0, **kw):
        b = hidden_states.shape[0]
        q = attn.to_q(hidden_states)
        e = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        k = attn.to_k(e)
        v = attn.to_v(e)
        h = attn.heads
        d = k.shape[-1] // h
        q = q.view(b, -1, h, d).transpose(1, 2)
        k = k.view(b, -1, h, d).transpose(1, 2)
        v = v.view(b, -1, h, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, h * d).to(q.dtype)
        o = attn.to_out[0](o)
        o = attn.to_out[1](o)
        return o / attn.rescale_output_factor


AttnProcessor = AttnProcessor2_0


class AttnAddedKVProcessor(AttnProcessor2_0):
    pass


AttentionProcessor = AttnProcessor2_0
ADDED_KV_ATTENTION_PROCESSORS = ()
CROSS_ATTENTION_PROCESSORS = (AttnProcessor2_0,)


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None,
                 cross_attention_norm_num_groups=32, added_kv_proj_dim=None, norm_num_groups=None,
                 spatial_norm_dim=None, out_bias=True, scale_qk=True, only_cross_attention=False, eps=1e-5,
                 rescale_output_factor=1.0, residual_connection=False, _from_deprecated_attn_block=False,
                 processor=None):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.rescale_output_factor = rescale_output_factor
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = processor or AttnProcessor2_0()

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x, scale=1.0):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        assert activation_fn == "geglu"
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim)])

    def forward(self, x, scale=1.0):
        for m in self.net:
            x = m(x)
        return x


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, x, scale=1.0):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        super().__init__()
        assert use_conv
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x, output_size=None, scale=1.0):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish",
                 skip_time_act=False, time_embedding_norm="default", kernel=None, output_scale_factor=1.0,
                 use_in_shortcut=None, up=False, down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        super().__init__()
        assert time_embedding_norm == "default" and not up and not down
        out_channels = out_channels or in_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups_out or groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.nonlinearity = get_activation(non_linearity)
        use_in_shortcut = in_channels != out_channels if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1, bias=conv_shortcut_bias) if use_in_shortcut else None

    def forward(self, x, temb, scale=1.0):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.output_scale_factor


class VaeImageProcessor:
    def __init__(self, *a, **k):
        pass


# ------------------------------------------------------------------ module tree
_self = sys.modules[__name__]
m = _mod("diffusers.configuration_utils"); m.ConfigMixin = ConfigMixin; m.register_to_config = register_to_config
m = _mod("diffusers.loaders"); m.UNet2DConditionLoadersMixin = UNet2DConditionLoadersMixin
m = _mod("diffusers.models"); m.ModelMixin = ModelMixin
m = _mod("diffusers.models.modeling_utils"); m.ModelMixin = ModelMixin
m = _mod("diffusers.models.activations"); m.get_activation = get_activation
m = _mod("diffusers.models.attention"); m.Attention = Attention; m.FeedForward = FeedForward; m.AdaLayerNorm = _NameOnly
m.GEGLU = GEGLU
m = _mod("diffusers.models.attention_processor")
for _n in ("Attention", "AttnProcessor", "AttnProcessor2_0", "AttentionProcessor", "AttnAddedKVProcessor",
           "ADDED_KV_ATTENTION_PROCESSORS", "CROSS_ATTENTION_PROCESSORS"):
    setattr(m, _n, getattr(_self, _n))
m = _mod("diffusers.models.embeddings"); m.Timesteps = Timesteps; m.TimestepEmbedding = TimestepEmbedding
for _n in ("GaussianFourierProjection", "ImageHintTimeEmbedding", "ImageProjection", "ImageTimeEmbedding", "PositionNet",
           "TextImageProjection", "TextImageTimeEmbedding", "TextTimeEmbedding", "CaptionProjection",
           "SinusoidalPositionalEmbedding"):
    setattr(m, _n, type(_n, (_NameOnly,), {}))
m = _mod("diffusers.models.resnet"); m.ResnetBlock2D = ResnetBlock2D; m.Downsample2D = Downsample2D; m.Upsample2D = Upsample2D
m = _mod("diffusers.models.lora"); m.LoRACompatibleConv = LoRACompatibleConv; m.LoRACompatibleLinear = LoRACompatibleLinear
m = _mod("diffusers.models.normalization"); m.AdaLayerNormSingle = type("AdaLayerNormSingle", (_NameOnly,), {})
m = _mod("diffusers.models.dual_transformer_2d"); m.DualTransformer2DModel = type("DualTransformer2DModel", (_NameOnly,), {})
m = _mod("diffusers.utils")
for _n in ("USE_PEFT_BACKEND", "BaseOutput", "deprecate", "logging", "scale_lora_layers", "unscale_lora_layers",
           "is_torch_version", "is_accelerate_available", "SAFETENSORS_WEIGHTS_NAME", "WEIGHTS_NAME"):
    setattr(m, _n, getattr(_self, _n))
m = _mod("diffusers.utils.import_utils"); m.is_xformers_available = is_xformers_available
m = _mod("diffusers.utils.torch_utils"); m.apply_freeu = apply_freeu; m.randn_tensor = randn_tensor
m = _mod("diffusers.image_processor"); m.VaeImageProcessor = VaeImageProcessor

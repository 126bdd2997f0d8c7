"""CLIP vision tower + projection on the MI355X kernels.

The reference pipeline calls a third-party transformers `CLIPVisionModelWithProjection` once per clip
(src/pipelines/pipeline_mikudance.py:406-416; built at scripts/inference_video.py:96-98):

    clip_image = CLIPImageProcessor().preprocess(ref_image.resize((224, 224)), return_tensors="pt").pixel_values
    emb   = image_encoder(clip_image).last_hidden_state                  # (1, 257, 1024): ALL tokens, before post_layernorm
    emb   = image_encoder.vision_model.post_layernorm(emb)
    embeds = image_encoder.visual_projection(emb)                         # (1, 257, 768) -> cross-attention context

This module keeps that surface -- `from_pretrained(dir)` (config.json + model.safetensors / pytorch_model.bin with the
transformers key layout), `__call__(pixel_values).last_hidden_state`, `.vision_model.post_layernorm(x)`,
`.visual_projection(x)`, `.dtype`, `.to()` -- so the pipeline code is unchanged, and runs on the same C-ABI kernels as the
UNets: the 14x14 stride-14 patch embedding is a GEMM over unfolded patches (K = 588 zero-padded to 640) whose epilogue adds
the position embeddings; every Linear is md_gemm_f16 (+bias, +quick-GELU `x * sigmoid(1.702 x)`, +residual); attention is
md_attention_fwd_f16 (16 heads, d = 64, L = 257, V produced transposed by the GEMM); LayerNorm is md_layernorm_f16.

Parity: pinned to transformers' own implementation (tests/golden/g11_clip.safetensors <- oracle/gen_golden.py g11, random
seeded weights, ViT-L/14 geometry and a reduced one)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from . import ops, packing
from .blocks import Affine, Linear, _Packed

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(image, size=224):
    """transformers CLIPImageProcessor().preprocess(img) defaults: RGB -> resize shortest edge to 224 (bicubic) -> centre crop
    224 -> 1/255 -> normalise.  The pipeline hands over `ref_image.resize((224, 224))`, for which resize and crop are the
    identity; other sizes go through PIL's bicubic like the third-party processor.  Returns (1, 3, 224, 224) float32."""
    from PIL import Image
    img = image.convert("RGB")
    w, h = img.size
    if (w, h) != (size, size):
        short = min(w, h)
        nw, nh = (size, int(h * size / w)) if w == short else (int(w * size / h), size)
        img = img.resize((nw, nh), resample=Image.BICUBIC)
        left, top = (nw - size) // 2, (nh - size) // 2
        img = img.crop((left, top, left + size, top + size))
    arr = np.asarray(img).astype(np.float32) * np.float32(1.0 / 255.0)          # rescale (float32, like the processor)
    arr = (arr - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
    return torch.from_numpy(arr).permute(2, 0, 1)[None].contiguous()


class _ClipAttention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = Linear(dim, dim), Linear(dim, dim), Linear(dim, dim), Linear(dim, dim)


class _ClipMLP(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.fc1, self.fc2 = Linear(dim, inner), Linear(inner, dim)


class _ClipLayer(_Packed):
    """CLIPEncoderLayer: x + attn(LN1(x)); x + fc2(quick_gelu(fc1(LN2(x))))."""

    def __init__(self, dim, inner, heads, eps):
        super().__init__()
        self.dim, self.heads, self.eps = dim, heads, eps
        self.self_attn = _ClipAttention(dim)
        self.layer_norm1 = Affine(dim)
        self.mlp = _ClipMLP(dim, inner)
        self.layer_norm2 = Affine(dim)

    def _pack(self, dev):
        L, V = packing.linear_weight, packing.vec
        a = self.self_attn
        return dict(n1w=V(self.layer_norm1.weight, dev), n1b=V(self.layer_norm1.bias, dev),
                    n2w=V(self.layer_norm2.weight, dev), n2b=V(self.layer_norm2.bias, dev),
                    qk=torch.cat([L(a.q_proj.weight, dev), L(a.k_proj.weight, dev)], 0).contiguous(),
                    qkb=torch.cat([V(a.q_proj.bias, dev), V(a.k_proj.bias, dev)], 0).contiguous(),
                    v=L(a.v_proj.weight, dev), vb=V(a.v_proj.bias, dev), o=L(a.out_proj.weight, dev), ob=V(a.out_proj.bias, dev),
                    f1=L(self.mlp.fc1.weight, dev), f1b=V(self.mlp.fc1.bias, dev), f2=L(self.mlp.fc2.weight, dev),
                    f2b=V(self.mlp.fc2.bias, dev))

    def forward(self, x, L):
        """x: [L, C] tokens of ONE image."""
        pk = self.packed()
        C, H = self.dim, self.heads
        n = ops.layernorm(x, pk["n1w"], pk["n1b"], eps=self.eps)
        qk = ops.gemm(n, pk["qk"], bias=pk["qkb"])
        vt = ops.gemm(n, pk["v"], bias=pk["vb"], transpose_out=True, ldc_t=packing.pad_to(L, 8))   # V^T rows 16-byte aligned
        a = ops.attention(qk[:, :C], qk[:, C:], vt, 1, H, C // H, L, L)
        x = ops.gemm(a, pk["o"], bias=pk["ob"], residual=x)
        n = ops.layernorm(x, pk["n2w"], pk["n2b"], eps=self.eps)
        h = ops.gemm(n, pk["f1"], bias=pk["f1b"], act=ops.ACT_QUICKGELU)
        return ops.gemm(h, pk["f2"], bias=pk["f2b"], residual=x)


class _PatchConv(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dim, 3, patch, patch))        # bias=False in CLIP


class _Embedding(nn.Module):
    def __init__(self, n, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, dim))


class _ClipEmbeddings(nn.Module):
    def __init__(self, dim, patch, n_pos):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.empty(dim))
        self.patch_embedding = _PatchConv(dim, patch)
        self.position_embedding = _Embedding(n_pos, dim)


class _ClipEncoder(nn.Module):
    def __init__(self, dim, inner, heads, layers, eps):
        super().__init__()
        self.layers = nn.ModuleList([_ClipLayer(dim, inner, heads, eps) for _ in range(layers)])


class _LayerNormModule(_Packed):
    """A LayerNorm that is also callable on (B, L, C) tensors (the pipeline calls vision_model.post_layernorm itself)."""

    def __init__(self, dim, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.empty(dim))
        self.bias = nn.Parameter(torch.empty(dim))

    def _pack(self, dev):
        return dict(w=packing.vec(self.weight, dev), b=packing.vec(self.bias, dev))

    @torch.no_grad()
    def forward(self, x):
        pk = self.packed()
        y = ops.layernorm(x.reshape(-1, x.shape[-1]).to(torch.float16).contiguous(), pk["w"], pk["b"], eps=self.eps)
        return y.view(x.shape).to(x.dtype)


class _Projection(_Packed):
    """visual_projection: Linear(hidden, projection_dim, bias=False), callable on (B, L, C)."""

    def __init__(self, dim, out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out, dim))

    def _pack(self, dev):
        return dict(w=packing.linear_weight(self.weight, dev))

    @torch.no_grad()
    def forward(self, x):
        y = ops.gemm(x.reshape(-1, x.shape[-1]).to(torch.float16).contiguous(), self.packed()["w"])
        return y.view(x.shape[:-1] + (y.shape[-1],)).to(x.dtype)


class CLIPVisionTransformer(_Packed):
    def __init__(self, cfg):
        super().__init__()
        dim, patch = cfg.hidden_size, cfg.patch_size
        self.cfg = cfg
        self.grid = cfg.image_size // patch
        self.n_tokens = self.grid * self.grid + 1
        self.embeddings = _ClipEmbeddings(dim, patch, self.n_tokens)
        self.pre_layrnorm = _LayerNormModule(dim, cfg.layer_norm_eps)          # (sic) the published key name
        self.encoder = _ClipEncoder(dim, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers, cfg.layer_norm_eps)
        self.post_layernorm = _LayerNormModule(dim, cfg.layer_norm_eps)

    def _pack(self, dev):
        e = self.embeddings
        w = e.patch_embedding.weight.detach()
        k = w[0].numel()
        kp = packing.pad_to(k, 64)
        wp = torch.zeros((w.shape[0], kp), device=dev, dtype=torch.float16)
        wp[:, :k] = w.reshape(w.shape[0], k).to(device=dev, dtype=torch.float16)       # (c, ky, kx) order = F.unfold's
        pos = e.position_embedding.weight.detach().to(device=dev, dtype=torch.float32)
        cls_row = (e.class_embedding.detach().to(device=dev, dtype=torch.float32) + pos[0]).to(torch.float16)
        return dict(wp=wp, kp=kp, k=k, pos=pos[1:].to(torch.float16).contiguous(), cls=cls_row)

    def forward_tokens(self, pixel_values):
        """pixel_values (1, 3, S, S) -> last_hidden_state tokens [n_tokens, C] fp16 (after pre_layrnorm and all layers)."""
        pk = self.packed()
        cfg = self.cfg
        if pixel_values.shape[0] != 1:
            raise NotImplementedError("one reference image per call (the pipeline's batch_size is 1)")
        if tuple(pixel_values.shape[-2:]) != (cfg.image_size, cfg.image_size):
            raise ValueError(f"Input image size ({pixel_values.shape[-2]}*{pixel_values.shape[-1]}) doesn't match model "
                             f"({cfg.image_size}*{cfg.image_size}).")
        dev = pixel_values.device
        p = cfg.patch_size
        cols = torch.nn.functional.unfold(pixel_values.to(torch.float16), kernel_size=p, stride=p)[0].t()   # [grid*grid, 3*p*p]
        a = torch.zeros((cols.shape[0], pk["kp"]), device=dev, dtype=torch.float16)
        a[:, :pk["k"]] = cols
        x = torch.empty((self.n_tokens, cfg.hidden_size), device=dev, dtype=torch.float16)
        x[0] = pk["cls"]
        ops.gemm(a, pk["wp"], residual=pk["pos"], out=x[1:])                     # patch embedding + position embedding
        x = ops.layernorm(x, self.pre_layrnorm.packed()["w"], self.pre_layrnorm.packed()["b"], eps=cfg.layer_norm_eps)
        for layer in self.encoder.layers:
            x = layer(x, self.n_tokens)
        return x


class CLIPVisionModelWithProjection(nn.Module):
    """Drop-in for transformers.CLIPVisionModelWithProjection on the calls the MikuDance pipelines make."""

    DEFAULTS = dict(hidden_size=768, intermediate_size=3072, projection_dim=512, num_hidden_layers=12, num_attention_heads=12,
                    num_channels=3, image_size=224, patch_size=32, hidden_act="quick_gelu", layer_norm_eps=1e-5)

    def __init__(self, config=None, **kw):
        super().__init__()
        c = dict(self.DEFAULTS)
        if config is not None:
            given = dict(config) if hasattr(config, "keys") else (config.to_dict() if hasattr(config, "to_dict") else vars(config))
            if isinstance(given.get("vision_config"), dict):                       # a full CLIPConfig json: vision tower + projection_dim
                vc = dict(given["vision_config"])
                vc.setdefault("projection_dim", given.get("projection_dim", c["projection_dim"]))
                given = vc
            c.update({k: v for k, v in given.items() if k in c})
        c.update({k: v for k, v in kw.items() if k in c})
        if c["hidden_act"] != "quick_gelu" or c["num_channels"] != 3:
            raise NotImplementedError("CLIPVisionModelWithProjection (MI355X): only quick_gelu / RGB towers are implemented")
        if c["hidden_size"] // c["num_attention_heads"] not in (8, 16, 32, 40, 64, 80, 160):
            raise NotImplementedError("attention head dim must be one of 8 / 16 / 32 / 40 / 64 / 80 / 160 (md_attention_fwd_f16)")
        self.config = SimpleNamespace(**c)
        self.vision_model = CLIPVisionTransformer(self.config)
        self.visual_projection = _Projection(c["hidden_size"], c["projection_dim"])

    @property
    def dtype(self):
        return self.visual_projection.weight.dtype

    @property
    def device(self):
        return self.visual_projection.weight.device

    @torch.no_grad()
    def forward(self, pixel_values, **unused):
        x = self.vision_model.forward_tokens(pixel_values)
        last = x[None]
        pooled = self.vision_model.post_layernorm(last[:, 0])
        return SimpleNamespace(last_hidden_state=last.to(pixel_values.dtype if pixel_values.dtype.is_floating_point else torch.float16),
                               image_embeds=self.visual_projection(pooled), pooler_output=pooled)

    @torch.no_grad()
    def image_prompt_embeds(self, pixel_values):
        """The three calls of the pipeline fused: tokens -> post_layernorm -> visual_projection, (1, n_tokens, projection_dim)."""
        x = self.vision_model.forward_tokens(pixel_values)
        pl = self.vision_model.post_layernorm.packed()
        n = ops.layernorm(x, pl["w"], pl["b"], eps=self.config.layer_norm_eps)
        return ops.gemm(n, self.visual_projection.packed()["w"])[None]

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        path = os.path.join(str(path), subfolder) if subfolder else str(path)
        cfg_file = os.path.join(path, "config.json")
        if not os.path.isfile(cfg_file):
            raise OSError(f"{cfg_file} does not exist or is not a file")
        model = cls(json.load(open(cfg_file)))
        st, bn = os.path.join(path, "model.safetensors"), os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st, device="cpu")
        elif os.path.exists(bn):
            sd = torch.load(bn, map_location="cpu", weights_only=True)
        else:
            raise OSError(f"no weights file (model.safetensors / pytorch_model.bin) found in {path}")
        # checkpoints written by transformers < 4.31 carry the (non-parameter) position_ids buffer; text-tower / logit_scale
        # entries of a full CLIP checkpoint are not part of the vision tower
        sd = {k: v for k, v in sd.items() if (k.startswith("vision_model.") or k.startswith("visual_projection."))
              and not k.endswith("position_ids")}
        model.load_state_dict(sd, strict=True)
        return model

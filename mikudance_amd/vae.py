"""AutoencoderKL on the MI355X kernels (the row after the denoising loop).

The reference pipeline calls a third-party diffusers==0.24.0 `AutoencoderKL` (sd-vae-ft-mse) for 3F+2 encodes and F decodes
per clip (src/pipelines/pipeline_mikudance.py:115-130 decode_latents, :456-549 `self.vae.encode(...).latent_dist.mean`;
built at scripts/inference_video.py:72-79).  This module keeps that interface -- `from_pretrained`, `.encode(x).latent_dist`
(`.mean`, `.sample()`), `.decode(z).sample`, `.dtype`, `.device`, `.config.scaling_factor` -- and the diffusers state-dict
key layout (incl. the legacy `query/key/value/proj_attn` attention names of the published checkpoint), and runs on the
same C-ABI kernels as the UNets: NHWC 3x3 convs (the encoder's downsampler pads (0,1,0,1): `pad_lo = 0`), GroupNorm(+SiLU),
GEMMs.  The mid-block attention has ONE head of 512 channels -- too wide for the flash kernel's register tile -- so it runs
per frame as QK^T GEMM -> row softmax (md_softmax_rows_f16) -> PV GEMM.

Parity: the oracle's restatement (oracle/cpu_ref.py vae_*) follows the published diffusers semantics; diffusers is not in
/root/reference, so this row is PARITY UNPINNED."""
import json
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import ops, packing
from .blocks import GROUPS, Affine, Conv, Linear, _Packed, tokens

EPS = 1e-6


class VaeResnet(_Packed):
    """diffusers ResnetBlock2D with temb_channels=None: GN+SiLU -> conv -> GN+SiLU -> conv (+1x1 shortcut), eps 1e-6."""

    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.norm1 = Affine(cin)
        self.conv1 = Conv(cin, cout, 3)
        self.norm2 = Affine(cout)
        self.conv2 = Conv(cout, cout, 3)
        self.conv_shortcut = Conv(cin, cout, 1) if cin != cout else None

    def _pack(self, dev):
        pk = dict(n1w=packing.vec(self.norm1.weight, dev), n1b=packing.vec(self.norm1.bias, dev),
                  n2w=packing.vec(self.norm2.weight, dev), n2b=packing.vec(self.norm2.bias, dev),
                  c1=packing.conv3x3_weight(self.conv1.weight, dev), c1b=packing.vec(self.conv1.bias, dev),
                  c2=packing.conv3x3_weight(self.conv2.weight, dev), c2b=packing.vec(self.conv2.bias, dev))
        if self.conv_shortcut is not None:
            pk["sc"] = packing.conv1x1_weight(self.conv_shortcut.weight, dev)
            pk["scb"] = packing.vec(self.conv_shortcut.bias, dev)
        return pk

    def forward(self, x):
        pk = self.packed()
        h = ops.groupnorm(x, pk["n1w"], pk["n1b"], GROUPS, EPS, silu=True)
        h = ops.conv3x3(h, pk["c1"], self.cout, bias=pk["c1b"])
        h = ops.groupnorm(h, pk["n2w"], pk["n2b"], GROUPS, EPS, silu=True)
        sc = x if self.conv_shortcut is None else ops.gemm(tokens(x), pk["sc"], bias=pk["scb"]).view(x.shape[:-1] + (self.cout,))
        return ops.conv3x3(h, pk["c2"], self.cout, bias=pk["c2b"], residual=sc)


class VaeSampler(_Packed):
    """Downsample2D(padding=0): F.pad (0,1,0,1) + 3x3 stride-2 conv, or Upsample2D: nearest 2x + 3x3 conv (folded)."""

    def __init__(self, c, up):
        super().__init__()
        self.c, self.up = c, up
        self.conv = Conv(c, c, 3)

    def _pack(self, dev):
        return dict(w=packing.conv3x3_weight(self.conv.weight, dev), b=packing.vec(self.conv.bias, dev))

    def forward(self, x):
        pk = self.packed()
        if self.up:
            return ops.conv3x3(x, pk["w"], self.c, bias=pk["b"], upsample=True)
        return ops.conv3x3(x, pk["w"], self.c, bias=pk["b"], stride=2, pad_lo=0)


class VaeAttention(_Packed):
    """diffusers Attention(heads=1, dim_head=C, bias=True, norm_num_groups=32, residual_connection=True) on a feature map."""

    def __init__(self, c):
        super().__init__()
        self.c = c
        self.group_norm = Affine(c)
        self.to_q, self.to_k, self.to_v = Linear(c, c), Linear(c, c), Linear(c, c)
        self.to_out = nn.ModuleList([Linear(c, c)])

    def _pack(self, dev):
        L, V = packing.linear_weight, packing.vec
        return dict(nw=V(self.group_norm.weight, dev), nb=V(self.group_norm.bias, dev),
                    q=L(self.to_q.weight, dev), qb=V(self.to_q.bias, dev), k=L(self.to_k.weight, dev), kb=V(self.to_k.bias, dev),
                    v=L(self.to_v.weight, dev), vb=V(self.to_v.bias, dev),
                    o=L(self.to_out[0].weight, dev), ob=V(self.to_out[0].bias, dev))

    def forward(self, x):
        pk = self.packed()
        B, Hh, Ww, C = x.shape
        L = Hh * Ww
        if L % 8:
            raise ValueError(f"AutoencoderKL mid attention: {Hh}x{Ww} tokens must be a multiple of 8")
        Lp = packing.pad_to(L, 64)                                       # K of the PV GEMM
        n = tokens(ops.groupnorm(x, pk["nw"], pk["nb"], GROUPS, EPS))
        q = ops.gemm(n, pk["q"], bias=pk["qb"])
        k = ops.gemm(n, pk["k"], bias=pk["kb"])
        a = torch.empty((B * L, C), device=x.device, dtype=torch.float16)
        s = torch.zeros((L, Lp), device=x.device, dtype=torch.float16)    # score matrix of one frame (padding columns stay 0)
        vt = torch.zeros((C, Lp), device=x.device, dtype=torch.float16)
        for b in range(B):
            rows = slice(b * L, (b + 1) * L)
            ops.gemm(n[rows], pk["v"], bias=pk["vb"], transpose_out=True, out=vt)      # V^T [C][Lp], columns >= L stay zero
            ops.gemm(q[rows], k[rows], out=s[:, :L])
            ops.softmax_rows_(s[:, :L], scale=C ** -0.5)
            ops.gemm(s, vt, out=a[rows])
        return ops.gemm(a, pk["o"], bias=pk["ob"], residual=tokens(x)).view(x.shape)


class VaeMid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(c, c), VaeResnet(c, c)])
        self.attentions = nn.ModuleList([VaeAttention(c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, down):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin, cout), VaeResnet(cout, cout)])
        if down:
            self.downsamplers = nn.ModuleList([VaeSampler(cout, up=False)])
        self.down = down

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.downsamplers[0](x) if self.down else x


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, up):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin if i == 0 else cout, cout) for i in range(3)])
        if up:
            self.upsamplers = nn.ModuleList([VaeSampler(cout, up=True)])
        self.up = up

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.upsamplers[0](x) if self.up else x


class Encoder(_Packed):
    def __init__(self, in_channels, latent_channels, chans):
        super().__init__()
        self.in_channels, self.zc = in_channels, 2 * latent_channels
        self.conv_in = Conv(in_channels, chans[0], 3)
        self.down_blocks = nn.ModuleList([_DownBlock(chans[max(i - 1, 0)], c, i < len(chans) - 1) for i, c in enumerate(chans)])
        self.mid_block = VaeMid(chans[-1])
        self.conv_norm_out = Affine(chans[-1])
        self.conv_out = Conv(chans[-1], self.zc, 3)
        self.c0 = chans[0]

    def _pack(self, dev):
        return dict(ci=packing.conv3x3_weight(self.conv_in.weight, dev), cib=packing.vec(self.conv_in.bias, dev),
                    nw=packing.vec(self.conv_norm_out.weight, dev), nb=packing.vec(self.conv_norm_out.bias, dev),
                    co=packing.conv3x3_weight(self.conv_out.weight, dev), cob=packing.vec(self.conv_out.bias, dev))

    def forward(self, x64):
        """x64: (B, H, W, 64) fp16, image channels first, zero padded.  Returns (B, H/8, W/8, 64) with 2*latent valid channels."""
        pk = self.packed()
        h = ops.conv3x3(x64, pk["ci"], self.c0, bias=pk["cib"])
        for blk in self.down_blocks:
            h = blk(h)
        h = self.mid_block(h)
        h = ops.groupnorm(h, pk["nw"], pk["nb"], GROUPS, EPS, silu=True)
        out = torch.zeros(h.shape[:3] + (64,), device=h.device, dtype=torch.float16)
        ops.conv3x3(h, pk["co"], self.zc, bias=pk["cob"], out=out[..., :self.zc])
        return out


class Decoder(_Packed):
    def __init__(self, latent_channels, out_channels, chans):
        super().__init__()
        self.out_channels = out_channels
        rev = list(reversed(chans))
        self.conv_in = Conv(latent_channels, rev[0], 3)
        self.mid_block = VaeMid(rev[0])
        self.up_blocks = nn.ModuleList([_UpBlock(rev[max(i - 1, 0)], c, i < len(rev) - 1) for i, c in enumerate(rev)])
        self.conv_norm_out = Affine(rev[-1])
        self.conv_out = Conv(rev[-1], out_channels, 3)
        self.c0 = rev[0]

    def _pack(self, dev):
        return dict(ci=packing.conv3x3_weight(self.conv_in.weight, dev), cib=packing.vec(self.conv_in.bias, dev),
                    nw=packing.vec(self.conv_norm_out.weight, dev), nb=packing.vec(self.conv_norm_out.bias, dev),
                    co=packing.conv3x3_weight(self.conv_out.weight, dev), cob=packing.vec(self.conv_out.bias, dev))

    def forward(self, z64):
        """z64: (B, h, w, 64) fp16 (post_quant_conv output, zero padded).  Returns (B, 8h, 8w, out_channels) fp16."""
        pk = self.packed()
        h = ops.conv3x3(z64, pk["ci"], self.c0, bias=pk["cib"])
        h = self.mid_block(h)
        for blk in self.up_blocks:
            h = blk(h)
        h = ops.groupnorm(h, pk["nw"], pk["nb"], GROUPS, EPS, silu=True)
        return ops.conv3x3(h, pk["co"], self.out_channels, bias=pk["cob"])


class DiagonalGaussian:
    """diffusers DiagonalGaussianDistribution: moments (B, 2z, h, w) -> mean | logvar (clamped to [-30, 20])."""

    def __init__(self, moments):
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device="cpu" if generator is not None and generator.device.type == "cpu"
                            else self.mean.device, dtype=torch.float32).to(self.mean.device, self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL(_Packed):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), latent_channels=4, layers_per_block=2,
                 norm_num_groups=32, act_fn="silu", scaling_factor=0.18215, sample_size=512, down_block_types=None, up_block_types=None,
                 force_upcast=True, **unused):
        super().__init__()
        if layers_per_block != 2 or norm_num_groups != GROUPS or act_fn != "silu":
            raise ValueError("AutoencoderKL: only the sd-vae-ft-mse family (2 layers per block, 32 groups, SiLU) is built")
        chans = tuple(block_out_channels)
        if any(c % 64 for c in chans):
            raise ValueError(f"AutoencoderKL (MI355X): block_out_channels {chans} must be multiples of 64 (the conv / GEMM kernels tile "
                             "the channel dimension by 64; sd-vae-ft-mse is (128, 256, 512, 512))")
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, block_out_channels=chans,
                                      latent_channels=latent_channels, layers_per_block=2, norm_num_groups=GROUPS, act_fn="silu",
                                      scaling_factor=scaling_factor, sample_size=sample_size)
        self.zc = latent_channels
        self.encoder = Encoder(in_channels, latent_channels, chans)
        self.decoder = Decoder(latent_channels, out_channels, chans)
        self.quant_conv = Conv(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = Conv(latent_channels, latent_channels, 1)

    dtype = property(lambda self: self.quant_conv.weight.dtype)
    device = property(lambda self: self.quant_conv.weight.device)

    def _pack(self, dev):
        def pad_k(w):                       # 1x1 conv weight (N, K, 1, 1) -> [N][64]
            out = torch.zeros((w.shape[0], 64), dtype=torch.float16, device=dev)
            out[:, :w.shape[1]] = w.detach().reshape(w.shape[0], w.shape[1]).to(dev, torch.float16)
            return out
        return dict(q=pad_k(self.quant_conv.weight), qb=packing.vec(self.quant_conv.bias, dev),
                    pq=pad_k(self.post_quant_conv.weight), pqb=packing.vec(self.post_quant_conv.bias, dev))

    # ---- reference-facing interface (NCHW tensors in / out, like diffusers)
    @torch.no_grad()
    def encode(self, x, return_dict=True):
        if x.dim() != 4 or x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError(f"AutoencoderKL.encode expects (B, C, H, W) with H, W multiples of 8, got {tuple(x.shape)}")
        pk = self.packed()
        B, C, H, W = x.shape
        st = x.stride()
        x64 = ops.pack_nhwc(x, B, 1, (st[0], 0, st[1], st[2], st[3]), 0, C, 64, H, W)
        h = self.encoder(x64)                                                            # (B, h, w, 64), 2z valid
        m64 = torch.zeros_like(h)
        ops.gemm(tokens(h), pk["q"], bias=pk["qb"], out=tokens(m64)[:, :2 * self.zc])
        moments = torch.empty((B, 2 * self.zc, H // 8, W // 8), device=x.device, dtype=x.dtype if x.dtype != torch.float64 else torch.float32)
        so = moments.stride()
        ops.unpack_nhwc(m64, moments, B, 1, (so[0], 0, so[1], so[2], so[3]), 2 * self.zc, H // 8, W // 8)
        out = SimpleNamespace(latent_dist=DiagonalGaussian(moments))
        return out if return_dict else (out.latent_dist,)

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        if z.dim() != 4 or z.shape[1] != self.zc:
            raise ValueError(f"AutoencoderKL.decode expects (B, {self.zc}, h, w), got {tuple(z.shape)}")
        pk = self.packed()
        B, C, h, w = z.shape
        st = z.stride()
        z64 = ops.pack_nhwc(z, B, 1, (st[0], 0, st[1], st[2], st[3]), 0, C, 64, h, w)
        p64 = torch.zeros_like(z64)
        ops.gemm(tokens(z64), pk["pq"], bias=pk["pqb"], out=tokens(p64)[:, :self.zc])
        y = self.decoder(p64)                                                            # (B, 8h, 8w, 3)
        img = torch.empty((B, y.shape[-1], 8 * h, 8 * w), device=z.device, dtype=z.dtype if z.dtype != torch.float64 else torch.float32)
        so = img.stride()
        ops.unpack_nhwc(y, img, B, 1, (so[0], 0, so[1], so[2], so[3]), y.shape[-1], 8 * h, 8 * w)
        out = SimpleNamespace(sample=img)
        return out if return_dict else (img,)

    def forward(self, sample, sample_posterior=False, generator=None):
        post = self.encode(sample).latent_dist
        return self.decode(post.sample(generator) if sample_posterior else post.mode())

    # ---- checkpoint loading (diffusers layout; the published sd-vae-ft-mse file uses the legacy attention names)
    LEGACY = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}

    @classmethod
    def convert_legacy_keys(cls, sd):
        out = {}
        for k, v in sd.items():
            for a, b in cls.LEGACY.items():
                if ".attentions." in k and a in k:
                    k = k.replace(a, b)
                    if v.dim() == 4:
                        v = v.reshape(v.shape[0], v.shape[1])
            out[k] = v
        return out

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kw):
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else str(pretrained_model_path)
        cfg_file = os.path.join(path, "config.json")
        if not os.path.isfile(cfg_file):
            raise RuntimeError(f"{cfg_file} does not exist or is not a file")
        cfg = {k: v for k, v in json.load(open(cfg_file)).items() if not k.startswith("_")}
        model = cls(**cfg)
        st = os.path.join(path, "diffusion_pytorch_model.safetensors")
        bn = os.path.join(path, "diffusion_pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st, device="cpu")
        elif os.path.exists(bn):
            sd = torch.load(bn, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no weights file found in {path}")
        model.load_state_dict(cls.convert_legacy_keys(sd), strict=True)
        return model

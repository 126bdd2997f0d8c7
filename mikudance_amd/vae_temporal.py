"""AutoencoderKLTemporalDecoder on the MI355X kernels (the `--video_decoder` switch).

The reference builds a third-party diffusers==0.24.0 `AutoencoderKLTemporalDecoder` (the Stable-Video-Diffusion VAE: the usual
encoder, a decoder whose every ResNet is followed by a temporal ResNet over the frame axis) at scripts/inference_video.py:72-75
and decodes latents in chunks of 16 frames with `vae.decode(chunk, num_frames=len(chunk)).sample`
(src/pipelines/pipeline_mikudance.py:132-150).  This module keeps that interface and the published state-dict layout:

    encoder.*  quant_conv.*                                   as AutoencoderKL (no post_quant_conv)
    decoder.conv_in, decoder.mid_block.resnets.{0,1}.{spatial_res_block, temporal_res_block}.{norm1,conv1,norm2,conv2},
    ....time_mixer.mix_factor, decoder.mid_block.attentions.0.{group_norm,to_q,to_k,to_v,to_out.0},
    decoder.up_blocks.i.resnets.{0,1,2}.(same) [+ spatial_res_block.conv_shortcut], decoder.up_blocks.i.upsamplers.0.conv,
    decoder.conv_norm_out, decoder.conv_out, decoder.time_conv_out        (temporal convs: Conv3d weights [Cout, Cin, 3, 1, 1])

Temporal pieces on the existing kernels, no new ones:
  * Conv3d (3,1,1), padding (1,0,0) over a clip = ONE implicit GEMM (K = 3 C): on the (clips, frames, h*w, C) view it is a 3 x 1
    filter over an image whose rows are the frames (md_conv_nhwc_f16, kw = 1; frames of a clip are contiguous) -- no im2col copy,
    one rounding like the reference's Conv3d;
  * GroupNorm of the temporal ResNet runs over (frames, h, w) per group = md_groupnorm_nhwc_f16 on the (clip, f*h*w, C) view;
  * AlphaBlender(merge_strategy="learned", switch_spatial_to_temporal_mix=True): out = (1 - s) x_spatial + s x_temporal with
    s = sigmoid(mix_factor) and x_temporal = x_spatial + h, i.e. x_spatial + s h: folded into the second temporal conv (weights
    and bias scaled by s when packing, x_spatial as its residual).

Parity: diffusers is neither in /root/reference nor in this image -- restated from the published implementation, PARITY UNPINNED
(checked against an independent functional restatement with F.conv3d in oracle/cpu_ref.py)."""
import json
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import ops, packing
from .blocks import GROUPS, Affine, Conv, _Packed, groupnorm_frames, tokens
from .vae import EPS, DiagonalGaussian, Encoder, VaeAttention, VaeResnet, VaeSampler


class _Conv3dT(nn.Module):
    """Parameter holder for nn.Conv3d(c_in, c_out, (3, 1, 1), padding=(1, 0, 0))."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 1, 1))
        self.bias = nn.Parameter(torch.empty(cout))


def _temporal_conv(x, w3, bias, frames, residual=None):
    """nn.Conv3d(C, Cout, (3, 1, 1), padding (1, 0, 0)) over clips of `frames` frames.  x: (B, H, W, C) fp16 with B = clips * frames,
    frames of a clip contiguous; w3: [Cout][3 taps (prev, this, next frame)][C].

    On the (clips, frames, H*W, C) view the temporal convolution IS a 3 x 1 filter over an image whose rows are the frames:
    md_conv_nhwc_f16 with kw = 1 -- one implicit GEMM with K = 3 C, the three taps summed in the fp32 accumulator and rounded
    ONCE like the reference's Conv3d (three accumulating token GEMMs, the form used until round 3, rounded the partial sums to
    fp16 twice more).  Clips beyond the kernel's image limits (2^24 pixels, 2^32 elements) fall back to that form."""
    B, H, W, C = x.shape
    hw = H * W
    cout = w3.shape[0]
    if frames * hw < (1 << 24) and frames * hw * C < (1 << 32):           # both limits of md_conv_nhwc_f16's tap arithmetic
        out = ops.conv3x3(x.view(B // frames, frames, hw, C), w3, cout, bias=bias, kw=1,
                          residual=tokens(residual) if residual is not None else None)
        return out.view(B, H, W, cout)
    taps = [w3.view(cout, 3, C)[:, t].contiguous() for t in range(3)]
    xt = tokens(x)
    out = ops.gemm(xt, taps[1], bias=bias, residual=tokens(residual) if residual is not None else None)
    if frames > 1:
        for c0 in range(0, B, frames):                                   # per clip: frame t gets x_{t-1} and x_{t+1}
            lo, hi = c0 * hw, (c0 + frames) * hw
            o_late, o_early = out[lo + hw:hi], out[lo:hi - hw]
            ops.gemm(xt[lo:hi - hw], taps[0], residual=o_late, out=o_late)
            ops.gemm(xt[lo + hw:hi], taps[2], residual=o_early, out=o_early)
    return out.view(B, H, W, -1)


class TemporalResnet(_Packed):
    """diffusers TemporalResnetBlock(in == out, temb_channels=None, eps): GN(f,h,w)+SiLU -> Conv3d(3,1,1) -> GN+SiLU -> Conv3d + x."""

    def __init__(self, c, eps):
        super().__init__()
        self.c, self.eps = c, eps
        self.norm1, self.conv1 = Affine(c), _Conv3dT(c, c)
        self.norm2, self.conv2 = Affine(c), _Conv3dT(c, c)

    def _pack3(self, conv, dev, scale=None):
        w = conv.weight.detach().to(dev, torch.float32)[:, :, :, 0, 0]                  # [Cout][Cin][3]
        b = conv.bias.detach().to(dev, torch.float32)
        if scale is not None:
            w, b = w * scale, b * scale
        # [Cout][tap][Cin]: K runs over (frame tap, channel), the layout of a 3 x 1 filter for md_conv_nhwc_f16(kw = 1)
        return w.permute(0, 2, 1).reshape(w.shape[0], -1).to(torch.float16).contiguous(), b.to(torch.float16).contiguous()

    def _pack(self, dev):
        V = packing.vec
        pk = dict(n1w=V(self.norm1.weight, dev), n1b=V(self.norm1.bias, dev), n2w=V(self.norm2.weight, dev), n2b=V(self.norm2.bias, dev))
        pk["c1"], pk["c1b"] = self._pack3(self.conv1, dev)
        return pk


class SpatioTemporalResBlock(_Packed):
    def __init__(self, cin, cout, eps=1e-6, temporal_eps=1e-5):
        super().__init__()
        self.spatial_res_block = VaeResnet(cin, cout)
        self.temporal_res_block = TemporalResnet(cout, temporal_eps)
        self.time_mixer = nn.Module()
        self.time_mixer.mix_factor = nn.Parameter(torch.empty(1))

    def _pack(self, dev):
        # x_spatial + sigmoid(mix) * temporal_residual: the blend rides on the second temporal conv
        s = torch.sigmoid(self.time_mixer.mix_factor.detach().to(dev, torch.float32))
        w3, b = self.temporal_res_block._pack3(self.temporal_res_block.conv2, dev, scale=s)
        return dict(c2=w3, c2b=b)

    def forward(self, x, frames):
        pk, t = self.packed(), self.temporal_res_block
        tp = t.packed()
        xs = self.spatial_res_block(x)
        h = groupnorm_frames(xs, tp["n1w"], tp["n1b"], t.eps, True, frames)
        h = _temporal_conv(h, tp["c1"], tp["c1b"], frames)
        h = groupnorm_frames(h, tp["n2w"], tp["n2b"], t.eps, True, frames)
        return _temporal_conv(h, pk["c2"], pk["c2b"], frames, residual=xs)


class _MidT(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(c, c), SpatioTemporalResBlock(c, c)])
        self.attentions = nn.ModuleList([VaeAttention(c)])

    def forward(self, x, frames):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, frames)), frames)


class _UpT(nn.Module):
    def __init__(self, cin, cout, up):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(cin if i == 0 else cout, cout) for i in range(3)])
        if up:
            self.upsamplers = nn.ModuleList([VaeSampler(cout, up=True)])
        self.up = up

    def forward(self, x, frames):
        for r in self.resnets:
            x = r(x, frames)
        return self.upsamplers[0](x) if self.up else x


class TemporalDecoder(_Packed):
    def __init__(self, latent_channels, out_channels, chans):
        super().__init__()
        rev = list(reversed(chans))
        self.out_channels, self.c0 = out_channels, rev[0]
        self.conv_in = Conv(latent_channels, rev[0], 3)
        self.mid_block = _MidT(rev[0])
        self.up_blocks = nn.ModuleList([_UpT(rev[max(i - 1, 0)], c, i < len(rev) - 1) for i, c in enumerate(rev)])
        self.conv_norm_out = Affine(rev[-1])
        self.conv_out = Conv(rev[-1], out_channels, 3)
        self.time_conv_out = _Conv3dT(out_channels, out_channels)

    def _pack(self, dev):
        pk = dict(ci=packing.conv3x3_weight(self.conv_in.weight, dev), cib=packing.vec(self.conv_in.bias, dev),
                  nw=packing.vec(self.conv_norm_out.weight, dev), nb=packing.vec(self.conv_norm_out.bias, dev),
                  co=packing.conv3x3_weight(self.conv_out.weight, dev), cob=packing.vec(self.conv_out.bias, dev))
        # time_conv_out over `out_channels` (3) channels: one GEMM on [x_{t-1} | x_t | x_{t+1}] rows, K = 9 zero padded to 64
        w = self.time_conv_out.weight.detach().to(dev, torch.float32)[:, :, :, 0, 0]    # [Cout][Cin][3]
        oc = self.out_channels
        wt = torch.zeros((oc, 64), device=dev, dtype=torch.float16)
        for t in range(3):
            wt[:, t * oc:(t + 1) * oc] = w[:, :, t].to(torch.float16)
        pk["tw"], pk["tb"] = wt, packing.vec(self.time_conv_out.bias, dev)
        return pk

    def forward(self, z64, frames):
        pk = self.packed()
        h = ops.conv3x3(z64, pk["ci"], self.c0, bias=pk["cib"])
        h = self.mid_block(h, frames)
        for blk in self.up_blocks:
            h = blk(h, frames)
        h = ops.groupnorm(h, pk["nw"], pk["nb"], GROUPS, EPS, silu=True)
        y = ops.conv3x3(h, pk["co"], self.out_channels, bias=pk["cob"])                  # (B, H, W, 3)
        B, H, W, oc = y.shape
        hw = H * W
        a = torch.zeros((B * hw, 64), device=y.device, dtype=torch.float16)
        yt = tokens(y)
        a[:, oc:2 * oc] = yt
        for c0 in range(0, B, frames):
            lo, hi = c0 * hw, (c0 + frames) * hw
            if frames > 1:
                a[lo + hw:hi, 0:oc] = yt[lo:hi - hw]
                a[lo:hi - hw, 2 * oc:3 * oc] = yt[lo + hw:hi]
        out = torch.zeros((B * hw, 8), device=y.device, dtype=torch.float16)
        ops.gemm(a, pk["tw"], bias=pk["tb"], out=out[:, :oc])
        return out.view(B, H, W, 8)


class AutoencoderKLTemporalDecoder(_Packed):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), latent_channels=4, layers_per_block=2,
                 sample_size=768, scaling_factor=0.18215, force_upcast=True, down_block_types=None, **unused):
        super().__init__()
        chans = tuple(block_out_channels)
        if layers_per_block != 2 or any(c % 64 for c in chans):
            raise ValueError("AutoencoderKLTemporalDecoder (MI355X): 2 layers per block and channel counts that are multiples of 64")
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, block_out_channels=chans,
                                      latent_channels=latent_channels, layers_per_block=2, sample_size=sample_size,
                                      scaling_factor=scaling_factor, force_upcast=force_upcast)
        self.zc = latent_channels
        self.encoder = Encoder(in_channels, latent_channels, chans)
        self.decoder = TemporalDecoder(latent_channels, out_channels, chans)
        self.quant_conv = Conv(2 * latent_channels, 2 * latent_channels, 1)

    dtype = property(lambda self: self.quant_conv.weight.dtype)
    device = property(lambda self: self.quant_conv.weight.device)

    def _pack(self, dev):
        w = self.quant_conv.weight
        q = torch.zeros((w.shape[0], 64), dtype=torch.float16, device=dev)
        q[:, :w.shape[1]] = w.detach().reshape(w.shape[0], w.shape[1]).to(dev, torch.float16)
        return dict(q=q, qb=packing.vec(self.quant_conv.bias, dev))

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        if x.dim() != 4 or x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError(f"encode expects (B, C, H, W) with H, W multiples of 8, got {tuple(x.shape)}")
        pk = self.packed()
        B, C, H, W = x.shape
        st = x.stride()
        h = self.encoder(ops.pack_nhwc(x, B, 1, (st[0], 0, st[1], st[2], st[3]), 0, C, 64, H, W))
        m64 = torch.zeros_like(h)
        ops.gemm(tokens(h), pk["q"], bias=pk["qb"], out=tokens(m64)[:, :2 * self.zc])
        hl, wl = h.shape[1], h.shape[2]
        moments = torch.empty((B, 2 * self.zc, hl, wl), device=x.device, dtype=x.dtype if x.dtype != torch.float64 else torch.float32)
        so = moments.stride()
        ops.unpack_nhwc(m64, moments, B, 1, (so[0], 0, so[1], so[2], so[3]), 2 * self.zc, hl, wl)
        out = SimpleNamespace(latent_dist=DiagonalGaussian(moments))
        return out if return_dict else (out.latent_dist,)

    @torch.no_grad()
    def decode(self, z, num_frames=1, return_dict=True, image_only_indicator=None):
        """z: (clips * num_frames, latent, h, w).  `image_only_indicator` is accepted for signature compatibility; with the
        decoder's merge_strategy="learned" the blend does not depend on it (diffusers AlphaBlender.get_alpha)."""
        if z.dim() != 4 or z.shape[1] != self.zc or z.shape[0] % num_frames:
            raise ValueError(f"decode expects (clips * num_frames, {self.zc}, h, w), got {tuple(z.shape)} with num_frames={num_frames}")
        B, C, h, w = z.shape
        st = z.stride()
        z64 = ops.pack_nhwc(z, B, 1, (st[0], 0, st[1], st[2], st[3]), 0, C, 64, h, w)
        y = self.decoder(z64, num_frames)                                                # (B, 8h, 8w, 8): 3 valid channels
        oc = self.decoder.out_channels
        H, W = y.shape[1], y.shape[2]                                                    # 2^(levels-1) x the latent size
        img = torch.empty((B, oc, H, W), device=z.device, dtype=z.dtype if z.dtype != torch.float64 else torch.float32)
        so = img.stride()
        ops.unpack_nhwc(y, img, B, 1, (so[0], 0, so[1], so[2], so[3]), oc, H, W)
        out = SimpleNamespace(sample=img)
        return out if return_dict else (img,)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kw):
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else str(pretrained_model_path)
        cfg_file = os.path.join(path, "config.json")
        if not os.path.isfile(cfg_file):
            raise RuntimeError(f"{cfg_file} does not exist or is not a file")
        cfg = {k: v for k, v in json.load(open(cfg_file)).items() if not k.startswith("_")}
        model = cls(**cfg)
        st, bn = os.path.join(path, "diffusion_pytorch_model.safetensors"), os.path.join(path, "diffusion_pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st, device="cpu")
        elif os.path.exists(bn):
            sd = torch.load(bn, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no weights file found in {path}")
        model.load_state_dict(sd, strict=True)
        return model

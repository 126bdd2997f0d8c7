"""reference_unet: SD-1.5 2-D UNet with a 20-channel conv_in, 4 MAN blocks and no conv_out, run in WRITE mode to
harvest the LayerNorm'd hidden states in front of its 16 spatial self-attention layers.

API mirror of the reference `UNet2DConditionModel` of src/models/unet_2d_mix.py:88-1384 (`from_unet`, state-dict
keys, `forward(sample[22ch], timestep, encoder_hidden_states, return_dict)`), plus the plain donor UNet of
src/models/unet_2d_condition.py used only as a weight source (scripts/inference_video.py:81-85).
"""
import json
import os
from dataclasses import dataclass

import torch
from torch import nn

from . import ops
from .blocks import MANModule
from .unet_3d_mix import _Config, _Skips, _UNetBase


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor

    def __getitem__(self, i):
        return (self.sample,)[i]


_NO_MOTION = dict(down=[False] * 4, up=[False] * 4, mid=False)


class _UNet2DBase(_UNetBase):
    kind = "2d"
    conv_in_mult = 1
    with_man = False

    def __init__(self, sample_size=None, in_channels: int = 4, out_channels: int = 4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, norm_num_groups=32, norm_eps: float = 1e-5, cross_attention_dim=768, attention_head_dim=8,
                 act_fn="silu", use_linear_projection=False, **unused):
        super().__init__()
        if hasattr(sample_size, "items"):
            # quirk 9 (src/models/unet_2d_mix.py:902): `cls(unet.config)` passes the donor's config as `sample_size`;
            # every other argument keeps its constructor default -- reproduced literally.
            sample_size = dict(sample_size)
        cfg = dict(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                   block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                   norm_num_groups=norm_num_groups, norm_eps=norm_eps, cross_attention_dim=cross_attention_dim,
                   attention_head_dim=attention_head_dim, act_fn=act_fn, use_linear_projection=use_linear_projection)
        cfg.update(unused)
        self.config = _Config(cfg)
        if layers_per_block != 2 or norm_num_groups != 32 or attention_head_dim != 8 or in_channels != 4 or act_fn != "silu" \
                or use_linear_projection:
            raise NotImplementedError("UNet2DConditionModel (MI355X): only the SD-1.5 geometry is implemented")
        self.in_channels = in_channels
        self._block_out = tuple(block_out_channels)
        flags = dict(down=[False] * len(block_out_channels), up=[False] * len(block_out_channels), mid=False)
        self._build(in_channels * self.conv_in_mult, self._block_out, cross_attention_dim, norm_eps, flags, {}, with_out=False)

    def _register_extra(self):
        if self.with_man:
            self.man_blocks = nn.ModuleList([MANModule(c, 2) for c in self._block_out])


class UNet2DConditionModelPlain(_UNet2DBase):
    """Weight donor (reference src/models/unet_2d_condition.py, conv_out removed :645-654).  Parameters only."""

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kw):
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else str(pretrained_model_path)
        cfg_file = os.path.join(path, "config.json")
        if not os.path.isfile(cfg_file):
            raise RuntimeError(f"{cfg_file} does not exist or is not a file")
        cfg = json.load(open(cfg_file))
        import inspect
        names = set(inspect.signature(_UNet2DBase.__init__).parameters) - {"self", "unused"}
        model = cls(**{k: v for k, v in cfg.items() if k in names})
        st = os.path.join(path, "diffusion_pytorch_model.safetensors")
        bn = os.path.join(path, "diffusion_pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st, device="cpu")
        elif os.path.exists(bn):
            sd = torch.load(bn, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no weights file found in {path}")
        sd = {k: v for k, v in sd.items() if not k.startswith(("conv_out.", "conv_norm_out."))}
        model.load_state_dict(sd, strict=True)
        return model

    def forward(self, *a, **k):
        raise NotImplementedError("the plain SD-1.5 UNet is only a weight donor for UNet2DConditionModel.from_unet")


class UNet2DConditionModel(_UNet2DBase):
    conv_in_mult = 5      # conv_in takes in_channels*5 = 20 guidance channels (src/models/unet_2d_mix.py:321-326)
    with_man = True

    @classmethod
    def from_unet(cls, unet):
        """reference src/models/unet_2d_mix.py:896-919."""
        newnet = cls(unet.config)
        with torch.no_grad():
            w = torch.zeros_like(newnet.conv_in.weight)
            w[:, :4] = unet.conv_in.weight
            newnet.conv_in.weight = nn.Parameter(w)
            newnet.conv_in.bias = nn.Parameter(unet.conv_in.bias.detach().clone())
        newnet.time_embedding.load_state_dict(unet.time_embedding.state_dict())
        newnet.down_blocks.load_state_dict(unet.down_blocks.state_dict(), strict=False)
        newnet.mid_block.load_state_dict(unet.mid_block.state_dict(), strict=False)
        newnet.up_blocks.load_state_dict(unet.up_blocks.state_dict(), strict=False)
        # MAN parameters have no donor: initialise like nn.Conv2d would, they are overwritten by the checkpoint
        for p in newnet.man_blocks.parameters():
            nn.init.normal_(p, std=0.02)
        newnet._pk = None
        return newnet

    # ------------------------------------------------------------------------------------------ internal NHWC forward
    def forward_nhwc(self, x, motion_at, cross, timesteps=None):
        """x: (B, h, w, 64) fp16 (20 guidance channels, zero padded); motion_at(h', w') -> (B, h', w', 64) nearest
        resized scene-motion map; timesteps: None (t = 0, what the pipeline passes: pipeline_mikudance.py:649), one value, or
        one value per sample.  Returns the (unused) sample."""
        pk = self.packed()
        dev = x.device
        B, hh, ww, _ = x.shape
        force_size = self._needs_upsample_size(hh, ww, len(self.down_blocks))
        t = torch.zeros(1) if timesteps is None else torch.as_tensor(timesteps, dtype=torch.float32).reshape(-1).cpu()
        if t.numel() > 1 and bool((t == t[0]).all()):
            t = t[:1]
        assert t.numel() in (1, B), f"timestep: one value or one per sample ({B}), got {t.numel()}"
        trows = self._time_rows(pk, t, dev)                                  # one group (every frame shares t) or one per frame
        fpg = B // t.numel()                                                 # frames per time-embedding row
        blocks = self.transformer_blocks_in_order()
        # last bank writer in EXECUTION order (down -> mid -> up): the last attention of the last up block
        last_writer = self.up_blocks[-1].attentions[-1].transformer_blocks[0] \
            if blocks and all(b.ref_mode == "write" for b in blocks) else None
        # the sample after the LAST bank write is discarded by the pipeline: skip that dead tail (result preserving)
        c0 = self.conv_in.weight.shape[0]
        skips = _Skips(self._skip_plan())                                    # skips are born inside their concat buffers
        x = ops.conv3x3(x, pk["cin"], c0, bias=pk["cinb"], out=skips.slot(B, hh, ww, c0, dev))
        for i, blk in enumerate(self.down_blocks):
            for j, r in enumerate(blk.resnets):
                H_, W_ = x.shape[1:3]
                dst = skips.slot(B, H_, W_, r.cout, dev)
                x = r(x, self._temb(pk, trows, r), fpg * H_ * W_, out=None if blk.has_cross_attention else dst)
                if blk.has_cross_attention:
                    x = blk.attentions[j](x, cross, out=dst)
            if blk.downsamplers is not None:
                x = blk.downsamplers[0](x, out=skips.slot(B, (x.shape[1] + 1) // 2, (x.shape[2] + 1) // 2, blk.downsamplers[0].c, dev))
            # MAN after the skips were captured (quirk 10, src/models/unet_2d_mix.py:1272-1289): reads the skip slice, writes a fresh tensor
            x = self.man_blocks[i](x, motion_at(x.shape[1], x.shape[2]))
        mb = self.mid_block
        x = mb.resnets[0](x, self._temb(pk, trows, mb.resnets[0]), fpg * x.shape[1] * x.shape[2])
        x = mb.attentions[0](x, cross)
        x = mb.resnets[1](x, self._temb(pk, trows, mb.resnets[1]), fpg * x.shape[1] * x.shape[2], out=skips.hidden_slot())
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                x = skips.pop(x)                                             # [hidden | skip]: nothing is copied
                H_, W_ = x.shape[1:3]
                last = j == len(blk.resnets) - 1
                dst = skips.hidden_slot() if len(skips) and not (last and blk.upsamplers is not None) else None
                x = r(x, self._temb(pk, trows, r), fpg * H_ * W_, out=None if blk.has_cross_attention else dst)
                if blk.has_cross_attention:
                    tb = blk.attentions[j].transformer_blocks[0]
                    if self.skip_dead_tail and tb is last_writer:
                        tb.stop_after_bank = True
                        try:
                            blk.attentions[j](x, cross)
                        finally:
                            tb.stop_after_bank = False
                        return None
                    x = blk.attentions[j](x, cross, out=dst)
            if blk.upsamplers is not None:
                x = blk.upsamplers[0](x, skips.top_hw() if force_size else None, out=skips.hidden_slot())
        return x

    skip_dead_tail = False

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None, attention_mask=None,
                cross_attention_kwargs=None, added_cond_kwargs=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, down_intrablock_additional_residuals=None, encoder_attention_mask=None,
                return_dict: bool = True):
        """sample: (B, 22, h, w) = 20 character-guidance channels + 2 scene-motion channels (:1208-1210)."""
        B, c, hh, ww = sample.shape
        t = torch.as_tensor(timestep).reshape(-1).float().cpu()             # scalar or (B,) (reference :1058-1072 expands a scalar)
        nchar = c - 2
        st = sample.stride()
        x = ops.pack_nhwc(sample, B, 1, (st[0], 0, st[1], st[2], st[3]), 0, nchar, 64, hh, ww)

        def motion_at(h2, w2):
            return ops.pack_nhwc(sample, B, 1, (st[0], 0, st[1], st[2], st[3]), nchar, 2, 64, h2, w2, hin=hh, win=ww)

        ctx = encoder_hidden_states
        cross = self._cross(ctx, list(range(B)) if ctx.shape[0] == B else [0] * B, sample.device)
        y = self.forward_nhwc(x, motion_at, cross, timesteps=t)
        if y is None:
            out = None
        else:
            out = torch.empty((B, y.shape[-1], y.shape[1], y.shape[2]), device=sample.device, dtype=sample.dtype)
            so = out.stride()
            ops.unpack_nhwc(y, out, B, 1, (so[0], 0, so[1], so[2], so[3]), y.shape[-1], y.shape[1], y.shape[2])
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)

"""denoising_unet: SD-1.5 UNet inflated over frames + AnimateDiff motion modules, MI355X-native.

API mirror of the reference `UNet3DConditionModel` (src/models/unet_3d_mix.py:34-691): same constructor keywords,
`from_pretrained_2d`, `load_state_dict` key layout (tests/golden/g6_state_dict_keys.json), `.in_channels`,
`forward(sample, timestep, encoder_hidden_states, ..., return_dict)` -> `UNet3DConditionOutput(sample)`.
Internally frames are folded into the batch and everything runs NHWC fp16 on the HIP kernels.
"""
import json
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import torch
from torch import nn

from . import ops, packing
from .blocks import (Affine, Conv, ConvSampler, CrossContext, MotionModule, ResnetBlock, SpatialTransformer,
                     TimestepEmbedding, _Packed, groupnorm_frames, timestep_sinusoid, tokens)


@dataclass
class UNet3DConditionOutput:
    sample: torch.Tensor

    def __getitem__(self, i):
        return (self.sample,)[i]


class _Config(dict):
    __getattr__ = dict.__getitem__


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, ctx_dim, attn, downsample, motion, kind, eps, mm_kwargs):
        super().__init__()
        self.has_cross_attention = attn
        self.resnets = nn.ModuleList([ResnetBlock(cin if i == 0 else cout, cout, temb, eps) for i in range(2)])
        if attn:
            self.attentions = nn.ModuleList([SpatialTransformer(cout, ctx_dim, kind) for _ in range(2)])
        if kind == "3d":
            self.motion_modules = nn.ModuleList([MotionModule(cout, **mm_kwargs) if motion else None for _ in range(2)]) \
                if motion else [None, None]
        self.downsamplers = nn.ModuleList([ConvSampler(cout, up=False)]) if downsample else None


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, temb, ctx_dim, attn, upsample, motion, kind, eps, mm_kwargs):
        super().__init__()
        self.has_cross_attention = attn
        res = []
        for i in range(3):
            skip = cin if i == 2 else cout
            rin = prev if i == 0 else cout
            res.append(ResnetBlock(rin + skip, cout, temb, eps))
        self.resnets = nn.ModuleList(res)
        if attn:
            self.attentions = nn.ModuleList([SpatialTransformer(cout, ctx_dim, kind) for _ in range(3)])
        if kind == "3d":
            self.motion_modules = nn.ModuleList([MotionModule(cout, **mm_kwargs) for _ in range(3)]) if motion else [None] * 3
        self.upsamplers = nn.ModuleList([ConvSampler(cout, up=True)]) if upsample else None


class _MidBlock(nn.Module):
    def __init__(self, c, temb, ctx_dim, motion, kind, eps, mm_kwargs):
        super().__init__()
        self.has_cross_attention = True
        self.attentions = nn.ModuleList([SpatialTransformer(c, ctx_dim, kind)])
        self.resnets = nn.ModuleList([ResnetBlock(c, c, temb, eps), ResnetBlock(c, c, temb, eps)])
        if kind == "3d":
            self.motion_modules = nn.ModuleList([MotionModule(c, **mm_kwargs)]) if motion else [None]


_SIDE_STREAMS = {}


def _side_streams(dev):
    """Two side streams per device for the two-queue evaluation of a CFG batch (created once; HIP maps them onto two hardware queues)."""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device=key), torch.cuda.Stream(device=key))
    return _SIDE_STREAMS[key]


class _Skips:
    """Bookkeeping of the concat buffers during one forward (see _UNetBase._skip_plan)."""

    def __init__(self, plan):
        self.plan, self.k, self.stack = plan, 0, []

    def slot(self, B, H, W, cs, device):
        """Destination of the next skip tensor to be produced: the right-hand channel slice of a fresh concat buffer."""
        c1 = self.plan[self.k]
        self.k += 1
        buf = torch.empty((B, H, W, c1 + cs), device=device, dtype=torch.float16)
        self.stack.append((buf, c1))
        return buf[..., c1:]

    def hidden_slot(self):
        """Where the up path writes the hidden state that meets the top skip: the left-hand slice of that skip's buffer."""
        buf, c1 = self.stack[-1]
        return buf[..., :c1]

    def top_hw(self):
        return self.stack[-1][0].shape[1:3]

    def pop(self, x):
        """The concatenated [hidden | skip] tensor; `x` must be the hidden slice written through hidden_slot()."""
        buf, c1 = self.stack.pop()
        assert x.data_ptr() == buf.data_ptr() and x.shape[-1] == c1 and x.shape[:3] == buf.shape[:3], "hidden state was not produced in place"
        return buf

    def __len__(self):
        return len(self.stack)


class _UNetBase(_Packed):
    """Shared skeleton of the two UNets (block layout of SD-1.5: 3 cross-attn levels + 1 plain level)."""

    kind = "2d"

    def _build(self, conv_in_ch, block_out_channels, cross_attention_dim, norm_eps, motion_flags, mm_kwargs, with_out):
        c0 = block_out_channels[0]
        temb = c0 * 4
        n = len(block_out_channels)
        self.conv_in = Conv(conv_in_ch, c0, 3)
        self.time_embedding = TimestepEmbedding(c0, temb)
        # registration order down -> (man) -> up -> mid mirrors the reference so that a DFS over the module tree
        # visits transformer blocks in the same order (mutual_mix_attention.py:292-301 sorts stably on that order)
        self.down_blocks = nn.ModuleList()
        out_c = c0
        for i in range(n):
            in_c, out_c = out_c, block_out_channels[i]
            self.down_blocks.append(_DownBlock(in_c, out_c, temb, cross_attention_dim, attn=i < n - 1, downsample=i < n - 1,
                                               motion=motion_flags["down"][i], kind=self.kind, eps=norm_eps, mm_kwargs=mm_kwargs))
        self._register_extra()
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(block_out_channels))
        out_c = rev[0]
        for i in range(n):
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, n - 1)]
            self.up_blocks.append(_UpBlock(in_c, out_c, prev, temb, cross_attention_dim, attn=i > 0, upsample=i < n - 1,
                                           motion=motion_flags["up"][i], kind=self.kind, eps=norm_eps, mm_kwargs=mm_kwargs))
        self.mid_block = _MidBlock(block_out_channels[-1], temb, cross_attention_dim, motion_flags["mid"], self.kind, norm_eps,
                                   mm_kwargs)
        if with_out:
            self.conv_norm_out = Affine(c0)
            self.conv_out = Conv(c0, 4, 3)
        self.norm_eps = norm_eps
        self._cross_cache = {}

    def _register_extra(self):
        pass

    # -------- skip connections are born inside the concat buffer of the up-block resnet that consumes them
    def _skip_plan(self):
        """The reference concatenates [hidden, skip] along the channels in front of every up-block resnet
        (src/models/unet_3d_blocks.py:736,877; diffusers unet_2d_blocks likewise): a pure copy of 2 x (C1 + Cs) bytes per element.
        Here the k-th skip tensor (production order: conv_in, every down layer, every downsampler) is WRITTEN by its producer
        into the right-hand Cs channels of a (B, H, W, C1 + Cs) buffer, the up path later writes its C1 hidden channels into the
        left part, and the resnet reads the buffer as it is.  Down-path consumers of a skip tensor read the channel slice
        (pixel pitch C1 + Cs).  Returns C1 for every skip in production order."""
        ups = [r for blk in self.up_blocks for r in blk.resnets]
        cs = [self.conv_in.weight.shape[0]]
        for blk in self.down_blocks:
            cs += [r.cout for r in blk.resnets]
            if blk.downsamplers is not None:
                cs.append(blk.downsamplers[0].c)
        assert len(cs) == len(ups), (len(cs), len(ups))
        return [ups[len(ups) - 1 - k].cin - c for k, c in enumerate(cs)]


    # -------- packed top-level weights: conv_in (Cin zero-padded to 64), time embedding, all time_emb_proj fused
    def _resnets(self):
        out = []
        for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks):
            out += list(blk.resnets)
        return out

    def _pack(self, dev):
        pk = dict(cin=packing.conv3x3_weight(self.conv_in.weight, dev), cinb=packing.vec(self.conv_in.bias, dev),
                  t1=packing.linear_weight(self.time_embedding.linear_1.weight, dev), t1b=packing.vec(self.time_embedding.linear_1.bias, dev),
                  t2=packing.linear_weight(self.time_embedding.linear_2.weight, dev), t2b=packing.vec(self.time_embedding.linear_2.bias, dev))
        rs = self._resnets()
        pk["tp"] = torch.cat([packing.linear_weight(r.time_emb_proj.weight, dev) for r in rs], 0).contiguous()
        pk["tpb"] = torch.cat([packing.vec(r.time_emb_proj.bias, dev) for r in rs], 0).contiguous()
        offs, o = [], 0
        for r in rs:
            offs.append(o)
            o += r.cout
        pk["tp_off"] = {id(r): (a, r.cout) for r, a in zip(rs, offs)}
        if hasattr(self, "conv_out"):
            pk["ow"], pk["ob"] = packing.vec(self.conv_norm_out.weight, dev), packing.vec(self.conv_norm_out.bias, dev)
            pk["co"], pk["cob"] = packing.conv3x3_weight(self.conv_out.weight, dev), packing.vec(self.conv_out.bias, dev)
        return pk

    def _time_rows(self, pk, timesteps, dev):
        """silu(time_embedding(sinusoid(t))) for each group, then EVERY resnet's time_emb_proj in one GEMM.
        reference src/models/unet_3d_mix.py:467-488 + src/models/resnet.py:226."""
        c0 = self.time_embedding.linear_1.weight.shape[1]
        e = timestep_sinusoid(timesteps, c0).to(device=dev, dtype=torch.float16)
        e = ops.gemm(e, pk["t1"], bias=pk["t1b"], act=ops.ACT_SILU)
        e = ops.gemm(e, pk["t2"], bias=pk["t2b"], act=ops.ACT_SILU)      # = silu(emb); emb itself is never used raw
        return ops.gemm(e, pk["tp"], bias=pk["tpb"])                       # [groups, sum(Cout)]

    def _temb(self, pk, trows, r):
        a, n = pk["tp_off"][id(r)]
        return trows[:, a:a + n]

    def _cross(self, ctx, index_list, dev):
        """ctx: (nkv, L, D) any float dtype; index_list: per-frame context batch.

        The padded fp16 copy (and, per block, its K/V projections) is step-invariant and cached -- ONE entry.  Identity of
        the source is never inferred from its address alone: the entry keeps a strong reference to the caller's tensor, so
        its storage cannot be freed and handed to a different tensor while the entry lives, and `_version` (shared by all
        views of a storage) catches in-place edits.  Blocks key their K/V cache on the CrossContext OBJECT (`is`)."""
        index_list = tuple(index_list)
        hit = self._cross_cache.get("entry")
        if hit is not None and hit.src.data_ptr() == ctx.data_ptr() and hit.src_version == ctx._version \
                and hit.src.shape == ctx.shape and hit.src.stride() == ctx.stride() and hit.src.dtype == ctx.dtype \
                and hit.src.device == ctx.device and hit.index_list == index_list:
            return hit
        self._cross_cache.clear()
        nkv, L, D = ctx.shape
        lpad = packing.pad_to(L, 8)
        buf = torch.zeros((nkv, lpad, D), device=dev, dtype=torch.float16)
        buf[:, :L] = ctx.to(device=dev, dtype=torch.float16)
        zero_ctx = (buf == 0).flatten(1).all(1).tolist()              # one host sync per new context, not per step
        zero_frames = 0
        while zero_frames < len(index_list) and zero_ctx[index_list[zero_frames]]:
            zero_frames += 1
        hit = CrossContext(buf.view(nkv * lpad, D), torch.tensor(index_list, dtype=torch.int32, device=dev), L, lpad,
                           zero_frames=zero_frames)
        hit.src, hit.src_version, hit.index_list = ctx, ctx._version, index_list
        self._cross_cache["entry"] = hit
        return hit

    def clear_context_cache(self):
        """Drop the cached cross-attention context and every block's K/V projections of it (called by the pipeline at the
        start and end of each denoise(): nothing outlives a clip)."""
        self._cross_cache.clear()
        for tb in self.transformer_blocks_in_order():
            tb._kv_cache = None

    def transformer_blocks_in_order(self):
        """DFS order of the reference's torch_dfs over (down, up, mid) -- see _build."""
        out = []
        for blk in list(self.down_blocks) + list(self.up_blocks) + [self.mid_block]:
            if blk.has_cross_attention:
                out += [a.transformer_blocks[0] for a in blk.attentions]
        return out

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    @staticmethod
    def _needs_upsample_size(h, w, levels):
        """The reference's `forward_upsample_size` (src/models/unet_3d_mix.py:447-455, src/models/unet_2d_mix.py:1016-1027):
        a latent that is not a multiple of 2**(levels-1) is still legal (any W, H % 8 == 0, scripts/inference_video.py:108);
        each upsampler then resizes to the spatial size of the skip it is about to meet instead of exactly 2x."""
        m = 2 ** (levels - 1)
        return bool(h % m or w % m)


class UNet3DConditionModel(_UNetBase):
    kind = "3d"

    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 center_input_sample: bool = False, flip_sin_to_cos: bool = True, freq_shift: int = 0,
                 down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                 mid_block_type="UNetMidBlock3DCrossAttn",
                 up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                 only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block: int = 2,
                 downsample_padding: int = 1, mid_block_scale_factor: float = 1, act_fn: str = "silu",
                 norm_num_groups: int = 32, norm_eps: float = 1e-5, cross_attention_dim: int = 1280,
                 attention_head_dim=8, dual_cross_attention=False, use_linear_projection=False, class_embed_type=None,
                 num_class_embeds=None, upcast_attention=False, resnet_time_scale_shift="default",
                 use_inflated_groupnorm=False, use_motion_module=False, motion_module_resolutions=(1, 2, 4, 8),
                 motion_module_mid_block=False, motion_module_decoder_only=False, motion_module_type=None,
                 motion_module_kwargs=None, unet_use_cross_frame_attention=None, unet_use_temporal_attention=None,
                 mode=None, task_type="action", **unused):
        super().__init__()
        cfg = {k: v for k, v in locals().items() if k not in ("self", "unused", "__class__")}
        self.config = _Config(cfg)
        unsupported = [n for n, bad in (("layers_per_block", layers_per_block != 2), ("norm_num_groups", norm_num_groups != 32),
                                        ("attention_head_dim", attention_head_dim != 8), ("in_channels", in_channels != 4),
                                        ("out_channels", out_channels != 4), ("act_fn", act_fn != "silu"),
                                        ("use_linear_projection", use_linear_projection), ("dual_cross_attention", dual_cross_attention),
                                        ("class_embed_type", class_embed_type is not None), ("num_class_embeds", num_class_embeds is not None),
                                        ("unet_use_temporal_attention", bool(unet_use_temporal_attention)),
                                        ("unet_use_cross_frame_attention", bool(unet_use_cross_frame_attention)),
                                        ("center_input_sample", center_input_sample),
                                        ("resnet_time_scale_shift", resnet_time_scale_shift != "default")) if bad]
        if unsupported:
            raise NotImplementedError(f"UNet3DConditionModel (MI355X): unsupported configuration values for {unsupported}; "
                                      "the hot path implements the SD-1.5 / MikuDance geometry")
        mmk = dict(motion_module_kwargs or {})
        if use_motion_module:
            if motion_module_type != "Vanilla" or mmk.get("num_attention_heads", 8) != 8 or mmk.get("num_transformer_block", 1) != 1 \
                    or tuple(mmk.get("attention_block_types", ("Temporal_Self", "Temporal_Self"))) != ("Temporal_Self", "Temporal_Self") \
                    or mmk.get("temporal_attention_dim_div", 1) != 1 or not mmk.get("temporal_position_encoding", False):
                raise NotImplementedError("only the MikuDance motion-module configuration (configs/inference/mikudance_config.yaml) is supported")
        mm_kwargs = dict(max_len=mmk.get("temporal_position_encoding_max_len", 24))
        n = len(block_out_channels)
        flags = dict(down=[bool(use_motion_module and (2 ** i in motion_module_resolutions) and not motion_module_decoder_only)
                           for i in range(n)],
                     up=[bool(use_motion_module and (2 ** (3 - i) in motion_module_resolutions)) for i in range(n)],
                     mid=bool(use_motion_module and motion_module_mid_block))
        self.in_channels = in_channels
        self.sample_size = sample_size
        self.mode = mode
        self.temporal_position_encoding_max_len = mm_kwargs["max_len"] if use_motion_module else None
        self.use_inflated_groupnorm = bool(use_inflated_groupnorm)
        self._build(in_channels, tuple(block_out_channels), cross_attention_dim, norm_eps, flags, mm_kwargs, with_out=True)

    # ------------------------------------------------------------------------------------------ internal NHWC forward
    def forward_nhwc(self, x, nb, f, timesteps, cross, halves_identical=False, two_queues=False):
        """x: (nb*f, h, w, 64) fp16 (4 latent channels, zero padded); returns pred tokens [(nb*f*h*w), 4].
        halves_identical: the caller GUARANTEES that the two clip-halves of x hold the same latents (classifier-free guidance: the loop of
        src/pipelines/pipeline_mikudance.py:626-633 feeds `torch.cat([latents] * 2)`) -- with equal timesteps, conv_in and the first resnet
        (no attention in front of them: nothing has seen the context or the bank yet) then produce the same tensor for both halves, so
        they run on ONE half and the result is copied (per-image arithmetic: bit-identical, tests/test_unets_gpu.py).
        two_queues (with halves_identical): behind those shared layers the unconditional and the conditional half are independent all the way
        to the output (per-image GroupNorm, per-row attention and LayerNorm, per-clip-half temporal attention: src/models/unet_3d_mix.py:
        418-598, src/models/mutual_mix_attention.py:173-201), and they are evaluated as TWO KERNEL QUEUES (_forward_two_queues)."""
        pk = self.packed()
        dev = x.device
        _, hh, ww, _ = x.shape
        force_size = self._needs_upsample_size(hh, ww, len(self.down_blocks))
        trows = self._time_rows(pk, timesteps, dev)                         # [nb, sumC]
        gf = 1 if self.use_inflated_groupnorm else f                        # plain nn.GroupNorm on 5-D: stats across frames
        B = x.shape[0]
        c0 = self.conv_in.weight.shape[0]
        tt = torch.as_tensor(timesteps).reshape(-1)
        share = bool(halves_identical and nb == 2 and float(tt[0]) == float(tt[-1]))
        b0 = self.down_blocks[0]
        if share and two_queues and x.is_cuda and (b0.has_cross_attention or b0.motion_modules[0] is not None):
            return self._forward_two_queues(x, f, trows, cross, pk, gf, force_size)
        skips = _Skips(self._skip_plan())
        first = None
        if share:
            slot = skips.slot(B, hh, ww, c0, dev)
            ops.conv3x3(x[:f], pk["cin"], c0, bias=pk["cinb"], out=slot[:f])
            slot[f:].copy_(slot[:f])
            x = slot
            # the first resnet on one half (time rows of group 0 = those of group 1), then both halves from the copy
            r = b0.resnets[0]
            if b0.has_cross_attention or b0.motion_modules[0] is not None:
                first = r(x[:f], self._temb(pk, trows, r), f * hh * ww, gf).repeat(2, 1, 1, 1)
            else:                                                        # the resnet is its layer's last operator: it writes the skip in place
                dst = skips.slot(B, hh, ww, r.cout, dev)
                r(x[:f], self._temb(pk, trows, r), f * hh * ww, gf, out=dst[:f])
                dst[f:].copy_(dst[:f])
                first = dst
        else:
            x = ops.conv3x3(x, pk["cin"], c0, bias=pk["cinb"], out=skips.slot(B, hh, ww, c0, dev))
        return tokens(self._body(x, skips, nb, f, trows, cross, pk, gf, force_size, first))

    def _body(self, x, skips, nb, f, trows, cross, pk, gf, force_size, first=None, out=None):
        """Everything behind conv_in.  x: conv_in's output (nb*f, h, w, C0), already in its skip slot; `first`: the first resnet's output when
        the caller has evaluated it (shared between the clip-halves); `out`: where conv_out writes, (nb*f, h, w, 4)."""
        dev = x.device
        B = x.shape[0]
        for bi, blk in enumerate(self.down_blocks):
            for j, r in enumerate(blk.resnets):
                H_, W_ = x.shape[1:3]
                mm = blk.motion_modules[j]
                last_op = not (blk.has_cross_attention or mm is not None)
                if first is not None and bi == 0 and j == 0:
                    x = first                                            # (written into its skip slot by the caller if it is the layer's last operator)
                else:
                    dst = skips.slot(B, H_, W_, r.cout, dev)             # the layer's LAST operator writes the skip in place
                    x = r(x, self._temb(pk, trows, r), f * H_ * W_, gf, out=None if not last_op else dst)
                if not last_op and first is not None and bi == 0 and j == 0:
                    dst = skips.slot(B, H_, W_, r.cout, dev)
                if blk.has_cross_attention:
                    x = blk.attentions[j](x, cross, out=None if mm is not None else dst)
                if mm is not None:
                    x = mm(x, nb, f, out=dst)
            if blk.downsamplers is not None:
                x = blk.downsamplers[0](x, out=skips.slot(B, (x.shape[1] + 1) // 2, (x.shape[2] + 1) // 2, blk.downsamplers[0].c, dev))
        mb = self.mid_block
        x = mb.resnets[0](x, self._temb(pk, trows, mb.resnets[0]), f * x.shape[1] * x.shape[2], gf)
        x = mb.attentions[0](x, cross)
        if mb.motion_modules[0] is not None:
            x = mb.motion_modules[0](x, nb, f)
        x = mb.resnets[1](x, self._temb(pk, trows, mb.resnets[1]), f * x.shape[1] * x.shape[2], gf, out=skips.hidden_slot())
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                x = skips.pop(x)                                         # [hidden | skip]: nothing is copied
                H_, W_ = x.shape[1:3]
                mm = blk.motion_modules[j]
                last = j == len(blk.resnets) - 1
                # the layer's last operator writes the next resnet's hidden slice -- unless an upsampler comes first (it does then)
                dst = skips.hidden_slot() if len(skips) and not (last and blk.upsamplers is not None) else None
                x = r(x, self._temb(pk, trows, r), f * H_ * W_, gf, out=None if (blk.has_cross_attention or mm is not None) else dst)
                if blk.has_cross_attention:
                    x = blk.attentions[j](x, cross, out=None if mm is not None else dst)
                if mm is not None:
                    x = mm(x, nb, f, out=dst)
            if blk.upsamplers is not None:
                x = blk.upsamplers[0](x, skips.top_hw() if force_size else None, out=skips.hidden_slot())
        x = groupnorm_frames(x, pk["ow"], pk["ob"], self.norm_eps, True, gf)
        return ops.conv3x3(x, pk["co"], 4, bias=pk["cob"], out=out)

    def _forward_two_queues(self, x, f, trows, cross, pk, gf, force_size):
        """The two clip-halves of a classifier-free-guidance batch as two kernel queues (stage B, without CU masks).

        One queue of B = 2f kernels leaves the chip idle wherever a launch cannot fill it -- partial last rounds of the persistent GEMM / conv
        grids, the 12 x 12 level, the tails of every launch -- and runs the HBM-class and the MFMA-class kernels strictly one after the other.
        Two queues of B = f kernels fill each other's gaps (profiles/r06_ab_two_queues.log: two half-batch queues finish 8 % sooner than the
        same launches back to back).  Layout of one call:
          calling stream : time rows, conv_in and the first resnet on ONE half (shared: halves_identical); the cross-attention K / V of every
                           block if this context is new (they are read by both queues); fork event
          queue 0        : the unconditional half -- blocks.CHAIN = 0: no bank, zero-context rows (the to_out bias as a row term)
          queue 1        : the conditional half   -- blocks.CHAIN = 1: every row reads the bank and the CLIP tokens
          calling stream : waits for both; the prediction of both halves sits in ONE buffer allocated before the fork.
        One Python thread enqueues queue 0, then queue 1 (the host runs several steps ahead of the GPU either way).  Memory: everything a queue
        allocates comes from that stream's pool of the caching allocator; the few tensors that cross streams (first-resnet output, time rows, the
        prediction buffer, conv_in's skip) are allocated on the calling stream BEFORE the fork and referenced until AFTER the join, and the
        calling stream allocates nothing in between.  Per-layer tables that are built lazily on first use (folded LayerNorms, positional row
        tables per (halves, frames)) would be built by queue 0 and read by queue 1: the FIRST two-queue call per latent shape therefore
        runs its queues one after the other (queue 1 waits for queue 0's end), every later one side by side.
        Results: each half goes through the same kernels with B = f instead of B = 2f; where the tile dispatch picks another flavour for the
        smaller problem the fp32 summation order differs (bits may differ from the one-queue evaluation, error against fp32 is the same:
        tests/test_two_queues_gpu.py); run to run the two-queue evaluation is bitwise reproducible (no atomics, no order dependence)."""
        from . import blocks
        dev = x.device
        _, hh, ww, _ = x.shape
        c0 = self.conv_in.weight.shape[0]
        main = torch.cuda.current_stream(dev)
        qs = _side_streams(dev)
        plan = self._skip_plan()
        sk = [_Skips(plan), _Skips(plan)]
        slot0 = sk[0].slot(f, hh, ww, c0, dev)
        ops.conv3x3(x[:f], pk["cin"], c0, bias=pk["cinb"], out=slot0)
        r = self.down_blocks[0].resnets[0]
        first = r(slot0, self._temb(pk, trows[0:1], r), f * hh * ww, gf)
        for tb in self.transformer_blocks_in_order():
            tb.context_kv(cross)
        pred = torch.empty((2 * f, hh, ww, 4), device=dev, dtype=torch.float16)
        primed = pk.setdefault("_two_queue_shapes", set())
        serial = (f, hh, ww) not in primed or getattr(self, "serialize_queues", False)     # serialize_queues: bench.py's per-launch timing pass
        fork = torch.cuda.Event()
        fork.record(main)
        done = []
        for c in (0, 1):
            with torch.cuda.stream(qs[c]):
                qs[c].wait_event(fork)
                if c == 1:
                    if serial:
                        qs[1].wait_event(done[0])
                    slot1 = sk[1].slot(f, hh, ww, c0, dev)
                    slot1.copy_(slot0)
                blocks.CHAIN = c
                try:
                    self._body(slot0 if c == 0 else slot1, sk[c], 1, f, trows[c:c + 1], cross.rows(c * f, (c + 1) * f), pk, gf, force_size,
                               first=first, out=pred[c * f:(c + 1) * f])
                finally:
                    blocks.CHAIN = None
                ev = torch.cuda.Event()
                ev.record(qs[c])
                done.append(ev)
        for ev in done:
            main.wait_event(ev)
        primed.add((f, hh, ww))
        del first, slot0                                                  # (referenced up to here: see above)
        return tokens(pred)

    # ------------------------------------------------------------------------------------------ reference-compatible forward
    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, attention_mask=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None, return_dict: bool = True,
                self_attention_additional_feats=None):
        if sample.dim() != 5:
            raise AssertionError(f"Expected hidden_states to have ndim=5, but got ndim={sample.dim()}.")
        if attention_mask is not None or down_block_additional_residuals is not None or mid_block_additional_residual is not None:
            raise NotImplementedError("attention_mask / ControlNet residuals are not on the MikuDance hot path")
        b, c, f, hh, ww = sample.shape
        st = sample.stride()
        x = ops.pack_nhwc(sample, b * f, f, (st[0], st[2], st[1], st[3], st[4]), 0, c, 64, hh, ww)
        t = torch.as_tensor(timestep).reshape(-1).cpu()
        t = t.expand(b) if t.numel() == 1 else t
        ctx = encoder_hidden_states
        if ctx.shape[0] == b * f:
            index = list(range(b * f))
        else:
            index = [i // f for i in range(b * f)]                            # 'b n c -> (b f) n c' (transformer_3d.py:122-125)
        cross = self._cross(ctx, index, sample.device)
        pred = self.forward_nhwc(x, b, f, t, cross)
        out = torch.empty((b, 4, f, hh, ww), device=sample.device, dtype=sample.dtype if sample.dtype != torch.float64 else torch.float32)
        so = out.stride()
        ops.unpack_nhwc(pred.view(b * f, hh, ww, 4), out, b * f, f, (so[0], so[2], so[1], so[3], so[4]), 4, hh, ww)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)

    # ------------------------------------------------------------------------------------------ loaders
    @classmethod
    def from_config(cls, config, **kwargs):
        import inspect
        names = set(inspect.signature(cls.__init__).parameters) - {"self", "unused"}
        args = {k: v for k, v in dict(config).items() if k in names}
        args.update({k: v for k, v in kwargs.items() if k in names})
        return cls(**args)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, motion_module_path, subfolder=None, unet_additional_kwargs=None,
                           mm_zero_proj_out=False):
        """Mirror of reference src/models/unet_3d_mix.py:600-691: SD-1.5 `unet/` directory (config.json +
        diffusion_pytorch_model.{safetensors,bin}) + motion-module state dict merged in, strict=False."""
        pretrained_model_path = Path(pretrained_model_path)
        motion_module_path = Path(motion_module_path)
        if subfolder is not None:
            pretrained_model_path = pretrained_model_path.joinpath(subfolder)
        config_file = pretrained_model_path / "config.json"
        if not (config_file.exists() and config_file.is_file()):
            raise RuntimeError(f"{config_file} does not exist or is not a file")
        unet_config = json.load(open(config_file))
        for k in ("down_block_types", "up_block_types", "mid_block_type", "_class_name"):
            unet_config.pop(k, None)
        kw = dict(unet_additional_kwargs or {})
        kw = {k: (dict(v) if hasattr(v, "items") else v) for k, v in kw.items()}
        model = cls.from_config(unet_config, **kw)
        st = pretrained_model_path / "diffusion_pytorch_model.safetensors"
        bn = pretrained_model_path / "diffusion_pytorch_model.bin"
        if st.exists():
            from safetensors.torch import load_file
            state_dict = load_file(str(st), device="cpu")
        elif bn.exists():
            state_dict = torch.load(bn, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no weights file found in {pretrained_model_path}")
        if motion_module_path.exists() and motion_module_path.is_file():
            suf = motion_module_path.suffix.lower()
            if suf in (".pth", ".pt", ".ckpt"):
                motion_state_dict = torch.load(motion_module_path, map_location="cpu", weights_only=True)
            elif suf == ".safetensors":
                from safetensors.torch import load_file
                motion_state_dict = load_file(str(motion_module_path), device="cpu")
            else:
                raise RuntimeError(f"unknown file format for motion module weights: {motion_module_path.suffix}")
            if mm_zero_proj_out:
                motion_state_dict = {k: v for k, v in motion_state_dict.items() if "proj_out" not in k}
            state_dict.update(motion_state_dict)
        model.load_state_dict(state_dict, strict=False)
        return model

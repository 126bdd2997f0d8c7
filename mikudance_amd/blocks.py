"""Building blocks shared by the two UNets.  Every module owns parameters under the REFERENCE's state-dict key
names (so the reference's .pth / SD-1.5 checkpoints load unchanged) and runs on NHWC fp16 activations through the
HIP kernels in mikudance_amd.ops.  Packed (kernel-layout) weights are built lazily and dropped whenever parameters
are reloaded or moved.

Activation convention: a (B, H, W, C) fp16 contiguous tensor; its 2-D view [B*H*W, C] is the token matrix, so the
reference's NCHW<->token shuffles (src/models/transformer_3d.py:121,134-136,185-189,201;
src/models/motion_module.py:159,166-168,182-189,404-406,437) do not exist here.
"""
import torch
from torch import nn

from . import ops, packing

GROUPS = 32
HEADS = 8


class _Packed(nn.Module):
    """nn.Module with a lazily built cache of kernel-layout weights."""

    def __init__(self):
        super().__init__()
        self._pk = None

    def _apply(self, fn, *a, **k):
        self._pk = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._pk = None
        return super()._load_from_state_dict(*a, **k)

    def packed(self):
        if self._pk is None:
            with torch.no_grad():
                self._pk = self._pack(next(self.parameters()).device)
        return self._pk


class Linear(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None


class Conv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))


class Affine(nn.Module):
    """weight/bias of a GroupNorm or LayerNorm."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c))
        self.bias = nn.Parameter(torch.empty(c))


def tokens(x):
    """[pixels, C] VIEW of an NHWC tensor -- also of a channel slice of a wider one (row pitch = the wider tensor's channel count).
    Used for operands and for `out=` destinations alike, so it must never be a copy: flatten() silently copies when the outer
    dimensions cannot be merged, and a kernel would then write into a temporary."""
    t = x.flatten(0, -2)
    if t.data_ptr() != x.data_ptr() or t.stride(-1) != 1:
        raise ops._lib.MdanceHipError(f"tokens(): shape {tuple(x.shape)} strides {tuple(x.stride())} has no [pixels, C] view")
    return t


def groupnorm_frames(x, w, b, eps, silu, gn_frames=1):
    """GroupNorm(32) of (B,H,W,C) frames; gn_frames consecutive frames share their statistics (frames of a clip-half are
    contiguous, so the cross-frame form is the same kernel on the (B/gn_frames, gn_frames*H*W, C) view)."""
    if gn_frames <= 1:
        return ops.groupnorm(x, w, b, GROUPS, eps, silu=silu)
    B, H, W, C = x.shape
    assert B % gn_frames == 0
    return ops.groupnorm(x.view(B // gn_frames, gn_frames * H * W, C), w, b, GROUPS, eps, silu=silu).view(x.shape)


# ------------------------------------------------------------------------------------------------ ResnetBlock
class ResnetBlock(_Packed):
    """ResnetBlock3D / diffusers ResnetBlock2D (reference src/models/resnet.py:123-247):
    GN+SiLU -> conv3x3 (+time_emb_proj(silu(temb)) fused as a row broadcast) -> GN+SiLU -> conv3x3 (+shortcut fused)."""

    def __init__(self, cin, cout, temb_ch, eps=1e-5):
        super().__init__()
        self.cin, self.cout, self.eps = cin, cout, eps
        self.norm1 = Affine(cin)
        self.conv1 = Conv(cin, cout, 3)
        self.time_emb_proj = Linear(temb_ch, cout)
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

    def forward(self, x, temb, rows_per_group, gn_frames=1, out=None):
        """x (B,H,W,cin); temb [groups, cout] = time_emb_proj(silu(emb)) rows (already projected, see TimeEmbedding).
        gn_frames > 1: GroupNorm statistics run over that many consecutive frames (the reference's plain nn.GroupNorm on
        the 5-D tensor when use_inflated_groupnorm=False, src/models/resnet.py:156-191); 1 = per frame (InflatedGroupNorm).
        x may be a channel slice of a wider NHWC tensor and `out` (B,H,W,cout) another one: see _UNetBase._skip_plan."""
        pk = self.packed()
        h = groupnorm_frames(x, pk["n1w"], pk["n1b"], self.eps, True, gn_frames)
        h = ops.conv3x3(h, pk["c1"], self.cout, bias=pk["c1b"], rowadd=temb, rows_per_group=rows_per_group)
        h = groupnorm_frames(h, pk["n2w"], pk["n2b"], self.eps, True, gn_frames)
        if self.conv_shortcut is not None:
            sc = ops.gemm(tokens(x), pk["sc"], bias=pk["scb"]).view(x.shape[:-1] + (self.cout,))
        else:
            sc = x
        return ops.conv3x3(h, pk["c2"], self.cout, bias=pk["c2b"], residual=sc, out=out)


class ConvSampler(_Packed):
    """Downsample (3x3 stride 2 pad 1) or Upsample (nearest 2x folded into the 3x3 conv's addressing).
    reference src/models/resnet.py:31-120; key `conv.{weight,bias}`."""

    def __init__(self, c, up):
        super().__init__()
        self.c, self.up = c, up
        self.conv = Conv(c, c, 3)

    def _pack(self, dev):
        return dict(w=packing.conv3x3_weight(self.conv.weight, dev), b=packing.vec(self.conv.bias, dev))

    def forward(self, x, size=None, out=None):
        """size=(h, w): forced nearest-resize target (the reference's `upsample_size` for latents that are not a multiple of
        2**levels, src/models/unet_3d_mix.py:447-455,564-586); the exact 2x case stays folded into the conv's addressing."""
        pk = self.packed()
        if self.up:
            B, H, W, C = x.shape
            if size is not None and tuple(size) != (2 * H, 2 * W):
                x = ops.pack_nhwc(x, B, 1, (x.stride(0), 0, 1, x.stride(1), x.stride(2)), 0, C, C, size[0], size[1], hin=H, win=W)
                return ops.conv3x3(x, pk["w"], self.c, bias=pk["b"], out=out)
            return ops.conv3x3(x, pk["w"], self.c, bias=pk["b"], upsample=True, out=out)
        return ops.conv3x3(x, pk["w"], self.c, bias=pk["b"], stride=2, out=out)


# ------------------------------------------------------------------------------------------------ attention / FF
class Attention(nn.Module):
    """Parameter holder for diffusers Attention (bias-free q/k/v, biased to_out.0)."""

    def __init__(self, dim, kv_dim=None):
        super().__init__()
        self.to_q = Linear(dim, dim, bias=False)
        self.to_k = Linear(kv_dim or dim, dim, bias=False)
        self.to_v = Linear(kv_dim or dim, dim, bias=False)
        self.to_out = nn.ModuleList([Linear(dim, dim)])


class _GEGLU(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = Linear(dim, dim * 8)


class FeedForward(nn.Module):
    """diffusers FeedForward(geglu): keys net.0.proj.*, net.2.*"""

    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([_GEGLU(dim), nn.Identity(), Linear(dim * 4, dim)])


def _pack_ff(ff, dev, pk, prefix="ff"):
    w, b = packing.geglu_weight(ff.net[0].proj.weight, ff.net[0].proj.bias, dev)
    pk[prefix + "1"], pk[prefix + "1b"] = w, b
    pk[prefix + "2"], pk[prefix + "2b"] = packing.linear_weight(ff.net[2].weight, dev), packing.vec(ff.net[2].bias, dev)


# Normalisations that never reach HBM (tests switch this off to compare against the literal operator sequence).  Where the
# W-stationary streaming GEMM serves the consumer with a plain epilogue (K = 320 on >= 32768 tokens: the 96 x 96 level), a LayerNorm is
# folded into its Linear (md_gemm_ln_f16: norm2 -> to_q, the motion module's norms -> q|k|v) and a SiLU-free GroupNorm is applied to the
# rows inside the GEMM (md_gemm_affine_f16: norm -> proj_in); everywhere else -- the other levels, the GEGLU consumers of norm3 / ff_norm
# -- the literal pair of operators runs (the plan queries say no: measured, profiles/r05_ab_fused_norms*.log).  norm1 stays a kernel of
# its own: its output IS the bank / K-V operand.
FUSE_NORMS = __import__("os").environ.get("MD_FUSE_NORMS", "1") != "0"      # MD_FUSE_NORMS=0: A/B runs against the literal operator pairs


def ln_linear(pk, h, norm, lin, bias=None, rowadd=None, rows_per_group=0, act=ops.ACT_NONE, eps=1e-5):
    """LayerNorm(pk[norm + 'w'], pk[norm + 'b']) -> Linear(pk[lin], pk[bias]) [+ row term] [GEGLU] on the token matrix h."""
    M, K = h.shape
    w = pk[lin]
    if FUSE_NORMS and ops.gemm_ln_plan(M, w.shape[0], K, act, rowadd is not None, a=h):
        fk = lin + ":ln"
        if fk not in pk:                  # folded once per layer, on first use (only the layers the fused kernel serves pay for it)
            pk[fk] = packing.ln_fold(w, None if bias is None else pk[bias], pk[norm + "w"], pk[norm + "b"])
        wf, sc = pk[fk]
        return ops.gemm_ln(h, wf, sc, eps=eps, rowadd=rowadd, rows_per_group=rows_per_group, act=act)
    n = ops.layernorm(h, pk[norm + "w"], pk[norm + "b"], eps=eps)
    return ops.gemm(n, w, bias=None if bias is None else pk[bias], rowadd=rowadd, rows_per_group=rows_per_group, act=act)


def gn_linear(x, gamma, beta, eps, w, bias):
    """GroupNorm(32, eps) (no SiLU) -> 1 x 1 conv / Linear on the tokens of x (B, H, W, C); returns [B*H*W, N]."""
    B, C = x.shape[0], x.shape[-1]
    M = x.numel() // C
    if FUSE_NORMS and ops.gemm_affine_plan(M, w.shape[0], C, M // B, x=x):
        return ops.gemm_affine(x, ops.groupnorm_table(x, gamma, beta, GROUPS, eps), w, bias=bias)
    return ops.gemm(tokens(ops.groupnorm(x, gamma, beta, GROUPS, eps)), w, bias=bias)


def _run_ff(pk, h, norm, prefix="ff"):
    """h + FeedForward(LayerNorm(h)) (GEGLU)."""
    g = ln_linear(pk, h, norm, prefix + "1", bias=prefix + "1b", act=ops.ACT_GEGLU)
    return ops.gemm(g, pk[prefix + "2"], bias=pk[prefix + "2b"], residual=h)


class CrossContext:
    """Per-forward cross-attention context: `ctx` [(nkv*Lpad), Dctx] zero-padded rows, `index` int32[B] mapping each
    frame to its context batch, Lk valid tokens, Lpad stride.  K/V projections of the context are cached per block
    (step-invariant; SURVEY.md 2.2 'cross k/v')."""

    def __init__(self, ctx, index, lk, lpad, zero_frames=0):
        self.ctx, self.index, self.lk, self.lpad = ctx, index, lk, lpad
        # leading frames whose context is all zeros (the CFG unconditional half: reference
        # src/pipelines/pipeline_mikudance.py:418-423 builds it with zeros_like).  For those rows K = V = 0 (to_k / to_v
        # have no bias), so the cross-attention output is exactly the to_out bias: see TransformerBlock.forward.
        self.zero_frames = zero_frames
        self.root = self                  # the context whose K / V projections the blocks cache (slices share their root's)
        self._slices = {}

    def rows(self, lo, hi):
        """The context as frames [lo, hi) of the batch see it (one CFG clip-half evaluated on its own queue): same padded rows, same cached
        K / V projections (`root`), the frames' own index entries and what is left of the leading zero-context frames."""
        s = self._slices.get((lo, hi))
        if s is None:
            s = CrossContext(self.ctx, self.index[lo:hi].contiguous(), self.lk, self.lpad, zero_frames=max(0, min(self.zero_frames, hi) - lo))
            s.root = self.root
            self._slices[(lo, hi)] = s
        return s


ZERO_CONTEXT_SKIP = True     # tests switch this off to compare against the literal evaluation

# Which clip-half of a classifier-free-guidance batch the CURRENT call evaluates when the two halves run as two kernel queues
# (UNet3DConditionModel.forward_nhwc, two_queues): None = the whole batch (rows [0, M/2) unconditional, the rest conditional);
# 0 = unconditional rows only (the bank is ignored: reference src/models/mutual_mix_attention.py:181-201);
# 1 = conditional rows only (every row reads the bank).  Set and reset by the UNet around each half's launches: the host enqueues
# the two halves one after the other (one Python thread), only the GPU runs them side by side.
CHAIN = None


class TransformerBlock(_Packed):
    """BasicTransformerBlock (2-D, reference src/models/attention.py:12-295) and TemporalBasicTransformerBlock (3-D,
    :298-484) share parameters; behaviour is selected by the reference-attention mode set through
    ReferenceAttentionControl (reference src/models/mutual_mix_attention.py:93-280):
      None    : plain self-attn -> cross-attn -> FF
      "write" : same, and bank = [norm1(x)]
      "read"  : K/V source = norm1(x) + bank on the conditional rows, plain on the unconditional rows (CFG)."""

    def __init__(self, dim, ctx_dim, heads=HEADS, kind="2d"):
        super().__init__()
        self.dim, self.heads, self.kind = dim, heads, kind
        self.norm1 = Affine(dim)
        self.attn1 = Attention(dim)
        self.norm2 = Affine(dim)
        self.attn2 = Attention(dim, ctx_dim)
        self.norm3 = Affine(dim)
        self.ff = FeedForward(dim)
        self.bank = []
        self.ref_mode = None
        self.ref_cfg = False
        self.stop_after_bank = False      # last writer block: everything after the bank write is dead code
        self._kv_cache = None             # (CrossContext object, K, V^T, {row tables}); matched with `is`, never by address

    def _pack(self, dev):
        L, V = packing.linear_weight, packing.vec
        a1, a2 = self.attn1, self.attn2
        pk = {f"n{i}w": V(n.weight, dev) for i, n in ((1, self.norm1), (2, self.norm2), (3, self.norm3))}
        pk.update({f"n{i}b": V(n.bias, dev) for i, n in ((1, self.norm1), (2, self.norm2), (3, self.norm3))})
        pk["q1"], pk["k1"], pk["v1"] = L(a1.to_q.weight, dev), L(a1.to_k.weight, dev), L(a1.to_v.weight, dev)
        pk["qk1"] = torch.cat([pk["q1"], pk["k1"]], 0).contiguous()
        pk["o1"], pk["o1b"] = L(a1.to_out[0].weight, dev), V(a1.to_out[0].bias, dev)
        pk["q2"], pk["k2"], pk["v2"] = L(a2.to_q.weight, dev), L(a2.to_k.weight, dev), L(a2.to_v.weight, dev)
        pk["o2"], pk["o2b"] = L(a2.to_out[0].weight, dev), V(a2.to_out[0].bias, dev)
        _pack_ff(self.ff, dev, pk)
        self._kv_cache = None
        return pk

    def context_kv(self, cross):
        """(K, V^T, {row tables}) of the cross-attention context, projected once per context (keyed on the context's root OBJECT: the row
        slices of a context that the two-queue evaluation hands to the clip-halves share their root's projections)."""
        if self._kv_cache is None or self._kv_cache[0] is not cross.root:
            pk = self.packed()
            self._kv_cache = (cross.root, ops.gemm(cross.ctx, pk["k2"]), ops.gemm(cross.ctx, pk["v2"], transpose_out=True), {})
        return self._kv_cache[1:]

    def forward(self, h, B, L, cross):
        """h: [B*L, C] tokens.  Returns tokens."""
        pk = self.packed()
        C, H = self.dim, self.heads
        D = C // H
        if self.ref_mode == "read" and len(self.bank) == 1 and not (self.ref_cfg and CHAIN == 0):
            bank = self.bank[0]
            brows = bank.shape[0] * bank.shape[1] if bank.dim() == 3 else bank.shape[0]
            M = h.shape[0]
            if self.ref_cfg and CHAIN == 1:
                # the conditional half on its own: every row reads the bank (the conditional frames' part of a literal 2f-frame bank)
                begin, b2 = 0, bank.reshape(-1, C)
                if brows == 2 * M:
                    b2 = b2[M:]
                assert b2.shape[0] == M, (b2.shape, M)
            elif self.ref_cfg:
                # unconditional rows (first half) ignore the bank (mutual_mix_attention.py:181-201)
                begin = M // 2
                b2 = bank.reshape(-1, C)
                if brows == M:
                    b2 = b2[begin:]
                assert b2.shape[0] == M - begin, (b2.shape, M)
            else:
                begin, b2 = 0, bank.reshape(-1, C)
                assert b2.shape[0] == M
            n, kv = ops.layernorm(h, pk["n1w"], pk["n1b"], add=b2.contiguous(), add_mode=1, add_row_begin=begin)
            q = ops.gemm(n, pk["q1"])
            k = ops.gemm(kv, pk["k1"])
            vt = ops.gemm(kv, pk["v1"], transpose_out=True)
        else:
            n = ops.layernorm(h, pk["n1w"], pk["n1b"])
            if self.ref_mode == "write":
                self.bank.append(n.view(B, L, C))
                if self.stop_after_bank:
                    return h
            qk = ops.gemm(n, pk["qk1"])
            q, k = qk[:, :C], qk[:, C:]
            vt = ops.gemm(n, pk["v1"], transpose_out=True)
        a = ops.attention(q, k, vt, B, H, D, L, L)
        kv2 = self.context_kv(cross)
        zf = min(cross.zero_frames, B) if ZERO_CONTEXT_SKIP else 0
        if zf:
            # frames with an all-zero context: cross-attention == to_out bias.  It rides on the attn1 out-projection as a
            # per-frame row-broadcast term, and those rows skip norm2 / to_q / attention / to_out altogether.
            tab = kv2[2].get((B, zf))
            if tab is None:
                tab = torch.zeros((B, C), device=h.device, dtype=torch.float16)
                tab[:zf] = pk["o2b"]
                kv2[2][(B, zf)] = tab
            h = ops.gemm(a, pk["o1"], bias=pk["o1b"], residual=h, rowadd=tab, rows_per_group=L)
        else:
            h = ops.gemm(a, pk["o1"], bias=pk["o1b"], residual=h)
        # cross attention to the CLIP tokens
        if zf < B:
            hs = h[zf * L:]
            q2 = ln_linear(pk, hs, "n2", "q2")
            a2 = ops.attention(q2, kv2[0], kv2[1], B - zf, H, D, L, cross.lk, kv_stride=cross.lpad, kv_index=cross.index[zf:])
            ops.gemm(a2, pk["o2"], bias=pk["o2b"], residual=hs, out=hs)    # in place: each element is read and written by one thread
        return _run_ff(pk, h, "n3")


class SpatialTransformer(_Packed):
    """Transformer2DModel / Transformer3DModel: GroupNorm(eps 1e-6) -> 1x1 conv -> block -> 1x1 conv -> + residual
    (reference src/models/transformer_2d.py:296-392, src/models/transformer_3d.py:106-205)."""

    def __init__(self, c, ctx_dim, kind):
        super().__init__()
        self.c = c
        self.norm = Affine(c)
        self.proj_in = Conv(c, c, 1)
        self.transformer_blocks = nn.ModuleList([TransformerBlock(c, ctx_dim, kind=kind)])
        self.proj_out = Conv(c, c, 1)

    def _pack(self, dev):
        return dict(nw=packing.vec(self.norm.weight, dev), nb=packing.vec(self.norm.bias, dev),
                    pi=packing.conv1x1_weight(self.proj_in.weight, dev), pib=packing.vec(self.proj_in.bias, dev),
                    po=packing.conv1x1_weight(self.proj_out.weight, dev), pob=packing.vec(self.proj_out.bias, dev))

    def forward(self, x, cross, out=None):
        pk = self.packed()
        B, Hh, Ww, C = x.shape
        blk = self.transformer_blocks[0]
        h = gn_linear(x, pk["nw"], pk["nb"], 1e-6, pk["pi"], pk["pib"])
        h = blk(h, B, Hh * Ww, cross)
        if blk.ref_mode == "write" and blk.stop_after_bank:
            return x
        if out is not None:
            ops.gemm(h, pk["po"], bias=pk["pob"], residual=tokens(x), out=tokens(out))
            return out
        return ops.gemm(h, pk["po"], bias=pk["pob"], residual=tokens(x)).view(x.shape)


# ------------------------------------------------------------------------------------------------ motion module
class _PositionalEncoding(nn.Module):
    def __init__(self, dim, max_len):
        super().__init__()
        from .synth import positional_encoding_table
        self.register_buffer("pe", positional_encoding_table(dim, max_len))


class _TemporalAttention(Attention):
    def __init__(self, dim, max_len):
        super().__init__(dim)
        self.pos_encoder = _PositionalEncoding(dim, max_len)


class _TemporalBlock(nn.Module):
    def __init__(self, dim, max_len):
        super().__init__()
        self.attention_blocks = nn.ModuleList([_TemporalAttention(dim, max_len), _TemporalAttention(dim, max_len)])
        self.norms = nn.ModuleList([Affine(dim), Affine(dim)])
        self.ff = FeedForward(dim)
        self.ff_norm = Affine(dim)


class _TemporalTransformer(nn.Module):
    def __init__(self, dim, max_len):
        super().__init__()
        self.norm = Affine(dim)
        self.proj_in = Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList([_TemporalBlock(dim, max_len)])
        self.proj_out = Linear(dim, dim)
        # zero_module(proj_out) at construction (reference src/models/motion_module.py:73-76): what `mm_zero_proj_out=True`
        # relies on when it drops the proj_out keys of the motion checkpoint.  Every other parameter is left unallocated-
        # uninitialised (torch.empty): the loaders overwrite all of them.
        nn.init.zeros_(self.proj_out.weight)
        nn.init.zeros_(self.proj_out.bias)


class MotionModule(_Packed):
    """VanillaTemporalModule (reference src/models/motion_module.py:45-272, 364-439): GroupNorm(1e-6) -> Linear ->
    2 x [LayerNorm -> attention over frames (PE on the query input only) + res] -> LayerNorm -> GEGLU FF + res ->
    Linear -> + residual."""

    def __init__(self, dim, max_len=32, heads=HEADS):
        super().__init__()
        self.dim, self.heads, self.max_len = dim, heads, max_len
        self.temporal_transformer = _TemporalTransformer(dim, max_len)

    def _pack(self, dev):
        L, V = packing.linear_weight, packing.vec
        tt = self.temporal_transformer
        tb = tt.transformer_blocks[0]
        pk = dict(nw=V(tt.norm.weight, dev), nb=V(tt.norm.bias, dev), pi=L(tt.proj_in.weight, dev), pib=V(tt.proj_in.bias, dev),
                  po=L(tt.proj_out.weight, dev), pob=V(tt.proj_out.bias, dev),
                  fnw=V(tb.ff_norm.weight, dev), fnb=V(tb.ff_norm.bias, dev))
        for i, (ab, nm) in enumerate(zip(tb.attention_blocks, tb.norms)):
            pk[f"n{i}w"], pk[f"n{i}b"] = V(nm.weight, dev), V(nm.bias, dev)
            # q = to_q(n + pe[frame]) (PE on the query input only, motion_module.py:416-417) = to_q(n) + pe[frame] @ Wq^T: the
            # second term is a per-frame constant row, so q, k and v come from ONE GEMM over n (N = 3C) whose epilogue adds
            # the [pe @ Wq^T | 0 | 0] row of the token's frame (row-broadcast term, as for the time embedding)
            pk[f"qkv{i}"] = torch.cat([L(ab.to_q.weight, dev), L(ab.to_k.weight, dev), L(ab.to_v.weight, dev)], 0).contiguous()
            pk[f"o{i}"], pk[f"o{i}b"] = L(ab.to_out[0].weight, dev), V(ab.to_out[0].bias, dev)
            pe = ab.pos_encoder.pe[0].detach().to(dev, torch.float32)
            tab = torch.zeros((pe.shape[0], 3 * self.dim), device=dev, dtype=torch.float32)
            tab[:, :self.dim] = pe @ ab.to_q.weight.detach().to(dev, torch.float32).t()
            pk[f"peq{i}"] = tab.to(torch.float16)
        _pack_ff(tb.ff, dev, pk)
        return pk

    def forward(self, x, nb, f, out=None):
        """x: (nb*f, H, W, C), frames of one clip-half contiguous."""
        pk = self.packed()
        if f > self.max_len:
            raise ValueError(f"window of {f} frames exceeds the positional-encoding table ({self.max_len})")
        _, Hh, Ww, C = x.shape
        HW, H = Hh * Ww, self.heads
        h = gn_linear(x, pk["nw"], pk["nb"], 1e-6, pk["pi"], pk["pib"])
        for i in range(2):
            tab = pk.get(("peq", i, nb, f))
            if tab is None:
                tab = pk[("peq", i, nb, f)] = pk[f"peq{i}"][:f].repeat(nb, 1).contiguous()       # one row per (clip-half, frame)
            qkv = ln_linear(pk, h, f"n{i}", f"qkv{i}", rowadd=tab, rows_per_group=HW)
            a = ops.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], nb, f, HW, H, C // H)
            h = ops.gemm(a, pk[f"o{i}"], bias=pk[f"o{i}b"], residual=h)
        h = _run_ff(pk, h, "fn")
        if out is not None:
            ops.gemm(h, pk["po"], bias=pk["pob"], residual=tokens(x), out=tokens(out))
            return out
        return ops.gemm(h, pk["po"], bias=pk["pob"], residual=tokens(x)).view(x.shape)


# ------------------------------------------------------------------------------------------------ MAN
class MANModule(_Packed):
    """Motion-Adaptive Normalization (reference src/models/man_module.py:7-33): InstanceNorm(x)*(1+gamma)+beta with
    gamma/beta = 3x3 convs over relu(3x3 conv(nearest-resized 2-channel scene-motion map))."""

    def __init__(self, norm_dim, m_dim=2):
        super().__init__()
        self.norm_dim, self.m_dim = norm_dim, m_dim
        self.mlp_shared = nn.Sequential(Conv(m_dim, 128, 3), nn.ReLU())
        self.mlp_gamma = Conv(128, norm_dim, 3)
        self.mlp_beta = Conv(128, norm_dim, 3)

    def _pack(self, dev):
        gb = torch.cat([self.mlp_gamma.weight, self.mlp_beta.weight], 0)
        return dict(sh=packing.conv3x3_weight(self.mlp_shared[0].weight, dev), shb=packing.vec(self.mlp_shared[0].bias, dev),
                    gb=packing.conv3x3_weight(gb, dev),
                    gbb=packing.vec(torch.cat([self.mlp_gamma.bias, self.mlp_beta.bias], 0), dev))

    def forward(self, x, motion_nhwc):
        """motion_nhwc: (B, h', w', 64) fp16 -- the 2 flow channels nearest-resized to x's resolution, zero padded."""
        pk = self.packed()
        a = ops.conv3x3(motion_nhwc, pk["sh"], 128, bias=pk["shb"], act=ops.ACT_RELU)
        gb = ops.conv3x3(a, pk["gb"], 2 * self.norm_dim, bias=pk["gbb"])
        return ops.instnorm_spade(x, gb)


# ------------------------------------------------------------------------------------------------ time embedding
class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = Linear(cin, dim)
        self.linear_2 = Linear(dim, dim)


def timestep_sinusoid(t, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) -- fp32 on the host (a few hundred values)."""
    import math
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    arg = torch.as_tensor(t, dtype=torch.float32).reshape(-1, 1) * freqs[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)

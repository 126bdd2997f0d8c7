"""ReferenceAttentionControl -- API mirror of reference src/models/mutual_mix_attention.py:19-378.

The reference monkey-patches `forward` of every (Temporal)BasicTransformerBlock; here the blocks implement the
write / read behaviour natively (mikudance_amd.blocks.TransformerBlock) and this class only sets their mode,
pairs reader and writer blocks in the reference's order, and moves the banks.
"""
import torch

from .blocks import TransformerBlock


def torch_dfs(model: torch.nn.Module):
    result = [model]
    for child in model.children():
        result += torch_dfs(child)
    return result


class ReferenceAttentionControl:
    def __init__(self, unet, mode="write", do_classifier_free_guidance=False, attention_auto_machine_weight=float("inf"),
                 gn_auto_machine_weight=1.0, style_fidelity=1.0, reference_attn=True, reference_adain=False,
                 fusion_blocks="midup", batch_size=1) -> None:
        assert mode in ["read", "write"]
        assert fusion_blocks in ["midup", "full"]
        if reference_adain:
            raise NotImplementedError("reference_adain is never enabled by the MikuDance pipelines")
        self.unet = unet
        self.mode = mode
        self.reference_attn = reference_attn
        self.fusion_blocks = fusion_blocks
        self.do_classifier_free_guidance = do_classifier_free_guidance
        if reference_attn:
            for i, m in enumerate(self._blocks(unet)):
                m.ref_mode = mode
                m.ref_cfg = bool(do_classifier_free_guidance)
                m.bank = []

    def _blocks(self, unet):
        """Transformer blocks in the reference's pairing order: DFS over the module tree (mid+up only for 'midup'),
        then a STABLE sort by descending width (:292-301)."""
        if self.fusion_blocks == "midup":
            mods = torch_dfs(unet.mid_block) + torch_dfs(unet.up_blocks)
        else:
            mods = torch_dfs(unet)
        blocks = [m for m in mods if isinstance(m, TransformerBlock)]
        return sorted(blocks, key=lambda x: -x.dim)

    def update(self, writer, dtype=torch.float16):
        """reader.bank <- writer.bank cast to fp16 UNCONDITIONALLY (quirk 3, :317-354).  Banks are already fp16 device
        tensors here, so no copy is made."""
        if self.reference_attn:
            for r, w in zip(self._blocks(self.unet), writer._blocks(writer.unet)):
                r.bank = [v if v.dtype == dtype else v.to(dtype) for v in w.bank]

    def clear(self):
        if self.reference_attn:
            for r in self._blocks(self.unet):
                r.bank = []

"""GPU: the CLIP vision tower + projection on the HIP kernels against transformers' OWN
CLIPVisionModelWithProjection (tests/golden/g11_clip.safetensors <- oracle/gen_golden.py g11): the exact call sequence of
src/pipelines/pipeline_mikudance.py:406-416.  fp16 path vs fp32 golden: relative L2 <= 2e-2, cosine >= 0.999."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file, save_file

pytestmark = pytest.mark.gpu

import mikudance_amd as M  # noqa: E402
from mikudance_amd.selftest import cosine, rel_l2  # noqa: E402
from mikudance_amd.synth import synth_state_dict  # noqa: E402


def _build(cfg, seed):
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in M.CLIPVisionModelWithProjection(cfg).state_dict().items()}
    sd = synth_state_dict(shapes, seed=seed)
    m = M.CLIPVisionModelWithProjection(cfg)
    m.load_state_dict(sd, strict=True)
    return m.to(device="cuda", dtype=torch.float16), sd


@pytest.fixture(scope="module")
def gold(golden_dir):
    return load_file(os.path.join(golden_dir, "g11_clip.safetensors")), json.load(open(os.path.join(golden_dir, "g11_meta.json")))


def _pixels(meta):
    s = meta["config"]["image_size"]
    return torch.randn(1, 3, s, s, generator=torch.Generator().manual_seed(meta["seed_pixels"]))


def test_clip_small_pipeline_call_sequence_vs_transformers(gold):
    g, meta = gold
    m = meta["small"]
    enc, sd = _build(m["config"], m["seed_weights"])
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - m["checksum"]) < 1e-6 * m["checksum"]
    px = _pixels(m).cuda().half()
    last = enc(px).last_hidden_state                                       # the three calls the pipeline makes
    emb = enc.visual_projection(enc.vision_model.post_layernorm(last))
    assert tuple(last.shape) == (1, 17, 128) and tuple(emb.shape) == (1, 17, 64) and emb.dtype == torch.float16
    assert rel_l2(last.float(), g["g11.small.last_hidden_state"]) < 2e-2
    assert rel_l2(emb.float(), g["g11.small.embeds"]) < 2e-2 and cosine(emb.float(), g["g11.small.embeds"]) > 0.999
    fused = enc.image_prompt_embeds(px)
    assert rel_l2(fused.float(), g["g11.small.embeds"]) < 2e-2
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 3, 42, 42, device="cuda", dtype=torch.float16))


def test_clip_vit_l14_vs_transformers(gold):
    """The real geometry: ViT-L/14 at 224x224 -> (1, 257, 768), 24 layers, 16 heads of 64."""
    g, meta = gold
    m = meta["l14"]
    enc, sd = _build(m["config"], m["seed_weights"])
    assert len(sd) == m["keys"]
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - m["checksum"]) < 1e-6 * m["checksum"]
    emb = enc.image_prompt_embeds(_pixels(m).cuda().half())
    want = g["g11.l14.embeds"].float()
    assert tuple(emb.shape) == (1, 257, 768)
    r, c = rel_l2(emb.float(), want), cosine(emb.float(), want)
    assert r < 2e-2 and c > 0.999, (r, c)


def test_clip_from_pretrained_and_pipeline_clip_embeds(gold, tmp_path):
    """from_pretrained on a transformers-style directory (config.json + model.safetensors, incl. the legacy position_ids buffer),
    then MikuDanceVideoPipeline.clip_embeds(PIL image) == the tower on clip_preprocess(image)."""
    import numpy as np
    from PIL import Image
    g, meta = gold
    m = meta["small"]
    cfg = dict(m["config"], image_size=224, patch_size=56, model_type="clip_vision_model", architectures=["CLIPVisionModelWithProjection"])
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in M.CLIPVisionModelWithProjection(cfg).state_dict().items()}
    sd = synth_state_dict(shapes, seed=3)
    sd["vision_model.embeddings.position_ids"] = torch.arange(17)[None]
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    save_file(sd, str(tmp_path / "model.safetensors"))
    enc = M.CLIPVisionModelWithProjection.from_pretrained(tmp_path).to("cuda", dtype=torch.float16)
    img = Image.fromarray(np.random.default_rng(1).integers(0, 255, (300, 200, 3), dtype=np.uint8))
    pipe = M.MikuDanceVideoPipeline(None, enc, None, None, None)
    got = pipe.clip_embeds(img)
    from oracle import cpu_ref as O
    want = O.clip_image_prompt_embeds({k: v for k, v in sd.items() if v.dtype.is_floating_point}, M.clip_preprocess(img.resize((224, 224))), 4, 56)
    assert tuple(got.shape) == (1, 17, 64)
    assert rel_l2(got.float(), want) < 2e-2 and cosine(got.float(), want) > 0.999
    with pytest.raises(OSError):
        M.CLIPVisionModelWithProjection.from_pretrained(tmp_path / "nope")

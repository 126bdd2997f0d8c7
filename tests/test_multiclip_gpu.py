"""GPU: more than one clip per process (the `log_validation` pattern of the reference, scripts/train_stage2.py:197-266, and
dp.shard with several clips per rank).  The cross-attention context (and its K/V projections) is cached per clip; a second
clip whose CLIP tokens have the same shape -- and, thanks to the caching allocator, very likely the same ADDRESS as the
freed tokens of the first -- must not see the first clip's context.  Each clip is checked against the CPU oracle."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu

from mikudance_amd import DDIMScheduler, MikuDanceVideoPipeline  # noqa: E402
from mikudance_amd.selftest import SCHED_KWARGS, build_models, cosine, rel_l2  # noqa: E402
from mikudance_amd.synth import synth_inputs  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402


@pytest.fixture(scope="module")
def models():
    return build_models()


def test_two_clips_same_shapes_each_match_the_oracle(models):
    ref, den, ref_sd, den_sd = models
    pipe = MikuDanceVideoPipeline(None, None, ref, den, DDIMScheduler(**SCHED_KWARGS))
    latA, rlA, embA = synth_inputs(4, 16, 16, ctx_len=5, ctx_dim=64, seed=100)
    latB, rlB, embB = synth_inputs(4, 16, 16, ctx_len=5, ctx_dim=64, seed=200)
    addrs = []
    outs = []
    for lat, rl, emb in ((latA, rlA, embA), (latB, rlB, embB), (latA, rlA, embA)):
        e = torch.cat([torch.zeros_like(emb[1:]), emb[1:]], 0).cuda().half()        # fresh `torch.cat` result, _version 0
        addrs.append(e.data_ptr())
        outs.append(pipe.denoise(lat.cuda().half(), rl.cuda().half(), e, 2, 3.5).float().cpu())
        del e
        gc.collect()
    with torch.no_grad():
        wantA = O.denoise_loop(ref_sd, den_sd, latA, rlA, embA, 2, guidance_scale=3.5, reduced=True)
        wantB = O.denoise_loop(ref_sd, den_sd, latB, rlB, embB, 2, guidance_scale=3.5, reduced=True)
    for got, want in zip(outs, (wantA, wantB, wantA)):
        assert rel_l2(got, want) < 3e-2 and cosine(got, want) > 0.999, (rel_l2(got, want), addrs)
    assert torch.equal(outs[0], outs[2])
    # and the context matters: clip B denoised with clip A's tokens is measurably different from B's own result
    wrong = pipe.denoise(latB.cuda().half(), rlB.cuda().half(), embA.cuda().half(), 2, 3.5).float().cpu()
    assert rel_l2(wrong, wantB) > 2 * rel_l2(outs[1], wantB)


def test_same_address_different_content_is_not_a_cache_hit(models):
    """Directly: a context tensor overwritten in place (same address, same shape; `_version` moves) and a new tensor that
    reuses a freed address must both be re-projected."""
    ref, den, _, _ = models
    idx = [0] * 2 + [1] * 2
    a = torch.randn(2, 5, 64, device="cuda").half()
    h1 = den._cross(a, idx, a.device)
    assert den._cross(a, idx, a.device) is h1 and den._cross(a[:2], idx, a.device) is h1      # same storage, same version
    a.mul_(2.0)
    h2 = den._cross(a, idx, a.device)
    assert h2 is not h1 and torch.equal(h2.ctx.view(2, 8, 64)[:, :5], a)
    seen = 0
    for trial in range(8):
        den.clear_context_cache()
        x = torch.randn(2, 5, 64, device="cuda").half()
        p = x.data_ptr()
        hx = den._cross(x, idx, x.device)
        del x                                       # the cache entry still owns the storage: it cannot be handed out again
        y = torch.randn(2, 5, 64, device="cuda").half()
        seen += int(y.data_ptr() == p)
        hy = den._cross(y, idx, y.device)
        assert hy is not hx and torch.equal(hy.ctx.view(2, 8, 64)[:, :5], y)
    assert seen == 0


class _ImageDependentCLIP(torch.nn.Module):
    """Duck-typed CLIP tower whose tokens depend on the picture (seeded by its mean)."""

    def __init__(self, tokens=5, dim=64):
        super().__init__()
        self.tokens, self.dim = tokens, dim
        self.vision_model = type("V", (), {"post_layernorm": torch.nn.Identity()})()
        self.visual_projection = torch.nn.Identity()
        self.p = torch.nn.Parameter(torch.zeros(1))

    dtype = property(lambda self: self.p.dtype)

    def forward(self, pixel_values):
        g = torch.Generator().manual_seed(int(abs(float(pixel_values.float().mean())) * 1e6) % 100003)
        h = torch.randn(1, self.tokens, self.dim, generator=g).to(pixel_values.device, pixel_values.dtype)
        return type("O", (), {"last_hidden_state": h})


def test_call_twice_like_log_validation(models):
    """pipe(refA ...) then pipe(refB ...) in one process == pipe(refB ...) in a clean state, bit for bit."""
    import numpy as np
    from PIL import Image
    from test_unets_gpu import _FakeVAE
    ref, den, _, _ = models
    H = W = 128
    F_ = 3
    rng = np.random.default_rng(0)
    img = lambda: Image.fromarray(rng.integers(0, 255, (160, 144, 3), dtype=np.uint8))
    flow = rng.uniform(-0.03, 0.03, (F_, 2, H // 8, W // 8))
    A = (img(), img(), [img() for _ in range(F_)], [img() for _ in range(F_)], [img() for _ in range(F_)], flow)
    B = (Image.fromarray(rng.integers(0, 64, (160, 144, 3), dtype=np.uint8)),) + A[1:]
    pipe = MikuDanceVideoPipeline(vae=_FakeVAE(), image_encoder=_ImageDependentCLIP(), reference_unet=ref, denoising_unet=den,
                                  scheduler=DDIMScheduler(**SCHED_KWARGS)).to("cuda", dtype=torch.float16)
    seen = {}
    orig = pipe.denoise

    def spy(latents, ref_latents, embeds, *a, **k):
        out = orig(latents, ref_latents, embeds, *a, **k)
        seen.setdefault("lat", []).append(out.float().cpu())
        return out

    pipe.denoise = spy
    pipe(*A, W, H, F_, 2, 3.5, generator=torch.manual_seed(1))
    pipe(*B, W, H, F_, 2, 3.5, generator=torch.manual_seed(2))
    den.clear_context_cache(); ref.clear_context_cache()
    torch.cuda.empty_cache()
    pipe(*B, W, H, F_, 2, 3.5, generator=torch.manual_seed(2))
    a, b, b_clean = seen["lat"]
    assert torch.equal(b, b_clean)
    assert not torch.equal(a, b)

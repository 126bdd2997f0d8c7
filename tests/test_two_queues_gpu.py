"""The two clip-halves of a classifier-free-guidance batch as two kernel queues (UNet3DConditionModel._forward_two_queues).

Behind conv_in and the first resnet the unconditional and the conditional half of the denoising UNet never meet again until the guidance formula
(reference src/models/unet_3d_mix.py:418-598: per-image GroupNorm, per-row LayerNorm / attention; src/models/mutual_mix_attention.py:173-201: the
bank is read by the conditional rows only; src/models/motion_module.py:245-268: temporal attention per clip-half), so they are evaluated on two
streams.  What must hold: (1) the result is the one-queue result up to the summation order of the kernels that the smaller batch selects, and as
close to the CPU oracle; (2) queues side by side == queues one after the other, BITWISE (same kernels, same order per queue: any difference is a
race between the queues); (3) run-to-run bitwise reproducible; (4) the per-half context slices and the literal (2f-frame) banks are handled."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(golden_dir):
    """The reduced-width pair of tests/test_unets_gpu.py (seeds and inputs of the G4 / G5 goldens)."""
    import json
    from safetensors.torch import load_file
    from mikudance_amd.selftest import build_models
    assert torch.cuda.is_available()
    meta = json.load(open(os.path.join(golden_dir, "g4_g5_meta.json")))
    ref, den, ref_sd, den_sd = build_models(seed_den=meta["seed_den"], seed_ref=meta["seed_ref"])
    return meta, ref, den, ref_sd, den_sd, load_file(os.path.join(golden_dir, "g4_g5_unets.safetensors"))


def _pipe(ref, den):
    from mikudance_amd import DDIMScheduler, MikuDanceVideoPipeline
    from mikudance_amd.selftest import SCHED_KWARGS
    pipe = MikuDanceVideoPipeline(None, None, ref, den, DDIMScheduler(**SCHED_KWARGS))
    pipe.two_queues = True                 # opt-in (default: one queue of B = 2f kernels)
    return pipe


def _primed(den):
    return den.packed().setdefault("_two_queue_shapes", set())


def test_two_queues_against_one_queue_and_the_oracle(small):
    """Reduced width, 4 frames, 3 DDIM steps with guidance: two queues vs one queue vs the fp32 CPU oracle on the same inputs."""
    from mikudance_amd.selftest import cosine, rel_l2
    from oracle import cpu_ref as O                                            # checker only
    meta, ref, den, ref_sd, den_sd, t = small
    lat, rl, emb = t["in.latents"][:, :, :4], t["in.ref_latents"][:, :4], t["in.embeds"]
    pipe = _pipe(ref, den)
    assert pipe.two_queues and pipe.share_first_layers
    args = (lat.cuda().half(), rl.cuda().half(), emb.cuda().half(), 3, 3.5)
    two = pipe.denoise(*args)
    assert (4, lat.shape[-2], lat.shape[-1]) in _primed(den)                   # ... and it did take the two-queue path
    pipe.two_queues = False
    one = pipe.denoise(*args)
    with torch.no_grad():
        want = O.denoise_loop(ref_sd, den_sd, lat.float(), rl.float(), emb.float(), 3, guidance_scale=3.5, reduced=True)
    e2, e1 = rel_l2(two.float(), want), rel_l2(one.float(), want)
    print(f"PARITY_MEASURE two_queues rel_l2 {e2:.3e} one_queue {e1:.3e} two-vs-one {rel_l2(two.float(), one.float()):.3e}")
    assert e2 < 3e-2 and cosine(two.float(), want) > 0.999                     # SURVEY's bound
    assert e2 < 1.5 * e1 + 1e-3                                                # ... and no worse than the one-queue evaluation
    assert rel_l2(two.float(), one.float()) < 1e-2


def test_queues_side_by_side_equal_queues_one_after_the_other_bitwise(small, full):
    """The race screen.  den.serialize_queues makes queue 1 wait for the end of queue 0: same kernels, same arguments, same order inside each
    queue -- only the overlap is gone.  Any bit of difference is a tensor that crossed the queues without an event.  Reduced width (several
    steps, repeated) and the benchmark's own width at 48 x 48 latents, 4 frames; the first call of a shape is serial by construction, so each
    side is run twice."""
    from mikudance_amd.synth import synth_inputs
    meta, ref, den, ref_sd, den_sd, t = small
    cases = [((ref, den), (t["in.latents"][:, :, :4].cuda().half(), t["in.ref_latents"][:, :4].cuda().half(), t["in.embeds"].cuda().half(), 4, 3.5), 3),
             (full[:2], tuple(x.half().cuda() for x in synth_inputs(4, 48, 48, ctx_len=257, ctx_dim=768, seed=5)) + (2, 3.5), 2)]
    for (r_, d_), args, reps in cases:
        pipe = _pipe(r_, d_)
        try:
            d_.serialize_queues = True
            serial = pipe.denoise(*args)
        finally:
            d_.serialize_queues = False
        for _ in range(reps):
            assert torch.equal(pipe.denoise(*args), serial)


def test_literal_evaluation_and_non_zero_unconditional_context(small):
    """reference_reuse = False keeps 2f-frame banks (the conditional queue takes their second half) and re-runs the reference UNet every step; an
    unconditional context that is NOT all zeros (a negative prompt) makes queue 0 run real cross-attention on its own row slice of the context."""
    from mikudance_amd.selftest import rel_l2
    meta, ref, den, ref_sd, den_sd, t = small
    lat, rl = t["in.latents"][:, :, :4].cuda().half(), t["in.ref_latents"][:, :4].cuda().half()
    emb = t["in.embeds"].cuda().half()
    emb_nz = emb.clone()
    emb_nz[0] = emb[1].flip(0) * 0.5
    for e, reuse in ((emb, False), (emb_nz, True), (emb_nz, False)):
        outs = []
        for tq in (True, False):
            pipe = _pipe(ref, den)
            pipe.reference_reuse, pipe.two_queues = reuse, tq
            outs.append(pipe.denoise(lat, rl, e, 2, 3.5).float())
        assert rel_l2(outs[0], outs[1]) < 1e-2, (reuse, rel_l2(outs[0], outs[1]))
    assert rel_l2(outs[0], _pipe(ref, den).denoise(lat, rl, emb, 2, 3.5).float()) > 1e-3      # the non-zero context did change the result


def test_context_rows_share_their_roots_projections():
    from mikudance_amd import blocks
    ctx = torch.zeros((2 * 8, 64), dtype=torch.float16)
    index = torch.tensor([0] * 3 + [1] * 3, dtype=torch.int32)
    c = blocks.CrossContext(ctx, index, 5, 8, zero_frames=3)
    u, k = c.rows(0, 3), c.rows(3, 6)
    assert u.root is c and k.root is c and c.rows(0, 3) is u
    assert u.zero_frames == 3 and k.zero_frames == 0 and u.index.tolist() == [0, 0, 0] and k.index.tolist() == [1, 1, 1]
    assert blocks.CrossContext(ctx, index, 5, 8, zero_frames=4).rows(3, 6).zero_frames == 1

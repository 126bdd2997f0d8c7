"""MikuDanceVideoPipeline -- API mirror of reference src/pipelines/pipeline_mikudance.py:36-704 whose denoising
loop (:573-686) runs on the MI355X-native UNets / HIP kernels of this package.

`__call__` keeps the reference signature and call order (CLIP embed -> VAE-encode every condition image -> 22-channel
guidance tensor -> loop -> VAE decode).  The VAE and the CLIP vision tower are whatever modules the caller passes
(`vae.encode(x).latent_dist.mean`, `vae.decode(z).sample`, `image_encoder(...)`) exactly as in the reference --
mikudance_amd.AutoencoderKL / mikudance_amd.CLIPVisionModelWithProjection run them on the same HIP kernels.  `denoise()` is the hot path itself on
tensors and is what bench.py and the parity tests drive.

Result-preserving reductions (proven identical on the CPU oracle in tests/test_oracle.py):
  * the reference UNet's inputs and t == 0 are step-invariant -> it is evaluated ONCE per window, not once per step;
  * only the conditional half of its banks is ever consumed -> it is evaluated on the f conditional frames only,
    each with the context row the reference's `[u,c,u,c,...]` interleaving would have given it;
  * the sample after its last bank write is discarded -> that tail is skipped;
  * unconditional rows ignore the bank -> one attention pass with a per-row K/V source replaces two.
`reference_reuse=False` restores the literal per-step evaluation (same results, for A/B timing).
"""
import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from . import ops
from .context import get_context_scheduler
from .mutual_mix_attention import ReferenceAttentionControl


# ---- latent interpolation helpers: API mirror of reference src/pipelines/utils.py (a module-level method switch that the
# caller sets; nothing in the reference sets it, so interpolation_factor >= 2 without it fails there exactly like here)
tensor_interpolation = None


def get_tensor_interpolation_method():
    return tensor_interpolation


def set_tensor_interpolation_method(is_slerp):
    global tensor_interpolation
    tensor_interpolation = slerp if is_slerp else linear


def linear(v1, v2, t):
    return (1.0 - t) * v1 + t * v2


def slerp(v0, v1, t, DOT_THRESHOLD=0.9995):
    """Spherical interpolation of two latent frames treated as single vectors (src/pipelines/utils.py:20-31); falls back to
    the straight line when they are nearly parallel."""
    u0 = v0 / v0.norm()
    u1 = v1 / v1.norm()
    dot = (u0 * u1).sum()
    if dot.abs() > DOT_THRESHOLD:
        return (1.0 - t) * v0 + t * v1
    omega = dot.acos()
    return (((1.0 - t) * omega).sin() * v0 + (t * omega).sin() * v1) / omega.sin()


@dataclass
class MikuDanceVideoPipelineOutput:
    videos: Union[torch.Tensor, np.ndarray]


def _pil_to_tensor(img, height, width, normalize):
    """VaeImageProcessor(do_convert_rgb=True[, do_normalize]).preprocess for one PIL image -> (1,3,H,W) fp32."""
    from PIL import Image
    img = img.convert("RGB").resize((width, height), resample=Image.LANCZOS)
    arr = torch.from_numpy(np.asarray(img).astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    return arr * 2.0 - 1.0 if normalize else arr


class MikuDanceVideoPipeline:
    _optional_components = []
    default_context_frames = 30

    def __init__(self, vae, image_encoder, reference_unet, denoising_unet, scheduler, image_proj_model=None, tokenizer=None,
                 text_encoder=None, video_decoder=False):
        self.vae, self.image_encoder = vae, image_encoder
        self.reference_unet, self.denoising_unet, self.scheduler = reference_unet, denoising_unet, scheduler
        self.image_proj_model, self.tokenizer, self.text_encoder = image_proj_model, tokenizer, text_encoder
        self.video_decoder = video_decoder
        self.decode_chunk_size = 16                                          # reference :81
        self.vae_scale_factor = 8
        self.vae_batch = 8                                                   # images per VAE call (the reference: 1)
        self.reference_reuse = True
        self.share_first_layers = True                                       # denoising UNet: conv_in + first resnet once for both CFG halves
        # ... and, behind them, OPTIONALLY the unconditional and the conditional half as two kernel queues (UNet3DConditionModel._forward_two_queues;
        # needs share_first_layers; pipe.two_queues = True or MD_TWO_QUEUES=1).  Off by default: on MI355X two queues of B = f kernels finish 7 % sooner
        # than the same launches back to back, but one queue of B = 2f kernels is already that much more efficient -- +0.2-0.35 % end to end
        # (profiles/r06_ab_two_queue_halves.log): measured, validated (tests/test_two_queues_gpu.py), not worth a second evaluation order by default
        self.two_queues = os.environ.get("MD_TWO_QUEUES", "0") == "1"
        self._device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")

    # ------------------------------------------------------------------------------------------ plumbing
    def to(self, device=None, dtype=None):
        for m in (self.vae, self.image_encoder, self.reference_unet, self.denoising_unet):
            if m is not None and hasattr(m, "to"):
                m.to(device=device, dtype=dtype) if dtype is not None else m.to(device)
        if device is not None:
            self._device = torch.device(device)
        return self

    @property
    def _execution_device(self):
        return self._device

    def progress_bar(self, iterable=None, total=None):
        from tqdm import tqdm
        return tqdm(iterable, total=total, disable=getattr(self, "_progress_disabled", True))

    def prepare_latents(self, batch_size, num_channels_latents, width, height, video_length, dtype, device, generator,
                        latents=None):
        """reference :173-207 -- noise is drawn from the caller's generator on ITS device (CPU for the script's
        torch.manual_seed generator, quirk 11), then moved."""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            gdev = generator.device if generator is not None and not isinstance(generator, list) else torch.device("cpu")
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    # ------------------------------------------------------------------------------------------ the hot path
    @torch.no_grad()
    def denoise(self, latents, ref_latents, image_prompt_embeds, num_inference_steps, guidance_scale, context_schedule="uniform",
                context_frames=None, context_stride=1, context_overlap=8, callback=None, callback_steps=1, eta=0.0, generator=None,
                window_parallel=None):
        """The loop of reference src/pipelines/pipeline_mikudance.py:573-686.

        latents             (1, 4, F, h, w)  initial noise (any float dtype, on the GPU)
        ref_latents         (1, F, 22, h, w) 20 VAE-latent guidance channels + 2 scene-motion channels
        image_prompt_embeds (2, L, D) = [zeros, CLIP tokens] when guidance_scale > 1, else (1, L, D)
        eta, generator      DDIM's stochastic variant (reference :152-171 -> scheduler.step(eta=, generator=)): one N(0, 1) draw of
                            the latents' shape and dtype per step from `generator` (on ITS device, like diffusers' randn_tensor).
                            The kernels compute in fp16: the draw is ROUNDED TO fp16 on its way into md_cfg_ddim_step_eta, so an
                            fp32-dtype caller does not get a noise stream bit-comparable with an fp32 reference run
        window_parallel     mikudance_amd.dp.WindowParallel or None: the context windows of a step (:625-668, independent UNet evaluations)
                            are shared out over the ranks of a process group, ONE all_reduce(sum) of the per-frame accumulators per step
                            (:662-674) and the CFG + DDIM update replicated on every rank; every rank returns the full latents
        returns latents (1, 4, F, h, w) in the input dtype.
        """
        dev = latents.device
        ops.require_gpu(latents, "MikuDanceVideoPipeline.denoise")
        context_frames = context_frames or self.default_context_frames
        do_cfg = guidance_scale > 1.0
        nb = 2 if do_cfg else 1
        den, refu, sch = self.denoising_unet, self.reference_unet, self.scheduler
        sch.set_timesteps(num_inference_steps)
        timesteps = [int(t) for t in sch.timesteps]
        _, c, F_, hh, ww = latents.shape
        HW = hh * ww
        writer = ReferenceAttentionControl(refu, do_classifier_free_guidance=do_cfg, mode="write", batch_size=1, fusion_blocks="full")
        reader = ReferenceAttentionControl(den, do_classifier_free_guidance=do_cfg, mode="read", batch_size=1, fusion_blocks="full")
        reader_blocks = reader._blocks(den)

        # internal latents: (F, h, w, 4) fp16 NHWC frames
        st = latents.stride()
        lat = ops.pack_nhwc(latents, F_, F_, (0, st[2], st[1], st[3], st[4]), 0, c, 4, hh, ww)
        noise_sum = torch.zeros((nb, F_, HW, 4), device=dev, dtype=torch.float32)
        counter = torch.zeros((F_,), device=dev, dtype=torch.float32)
        windows = [list(w) for w in get_context_scheduler(context_schedule)(0, num_inference_steps, F_, context_frames,
                                                                            context_stride, context_overlap)]
        mm_len = getattr(den, "temporal_position_encoding_max_len", None)
        if mm_len is not None and max(len(w) for w in windows) > mm_len:
            raise ValueError(f"windows of {max(len(w) for w in windows)} frames exceed the motion module's positional-encoding "
                             f"table (temporal_position_encoding_max_len = {mm_len})")
        # A wrapped, dilated window (context_stride >= 2, F < 2*size) can name a frame twice.  The reference's
        # `noise_pred[:, :, c] = noise_pred[:, :, c] + pred` (:662-666) is an index_put with duplicate indices: the LAST
        # occurrence's value lands and the counter grows by one.  Earlier occurrences get slot -1 = "do not accumulate".
        win_dev = [torch.tensor([fr if fr not in w[j + 1:] else -1 for j, fr in enumerate(w)], dtype=torch.int32, device=dev)
                   for w in windows]
        win_long = [torch.tensor(w, dtype=torch.long, device=dev) for w in windows]     # gather indices: the real frames
        whole = len(windows) == 1 and windows[0] == list(range(F_))
        embeds = image_prompt_embeds
        bank_cache = {}
        refu.skip_dead_tail = True
        den.clear_context_cache()
        refu.clear_context_cache()
        try:
            for step_i, t in enumerate(timesteps):
                noise_sum.zero_()
                counter.zero_()
                for wi, win in enumerate(windows):
                    if window_parallel is not None and not window_parallel.mine(wi):
                        continue                                         # another rank's window (its share arrives in the all_reduce)
                    f = len(win)
                    # ---- reference UNet (write): once per window unless reference_reuse is off
                    if wi not in bank_cache or not self.reference_reuse:
                        self._write_banks(writer, reader, embeds, ref_latents, win_long[wi], f, do_cfg, literal=not self.reference_reuse)
                        banks = [blk.bank for blk in reader_blocks]
                        if self.reference_reuse:
                            bank_cache[wi] = banks
                    else:
                        for blk, bk in zip(reader_blocks, bank_cache[wi]):
                            blk.bank = bk
                    # ---- denoising UNet (read)
                    src = lat if whole else lat.index_select(0, win_long[wi])
                    x = ops.pack_nhwc(src, nb * f, f, (0, HW * 4, 1, ww * 4, 4), 0, 4, 64, hh, ww)
                    cross = den._cross(embeds[:nb], [i // f for i in range(nb * f)], dev)
                    # both clip-halves are packed from the SAME latents (batch stride 0 above): the layers in front of the first attention
                    # run once (self.share_first_layers = False: the literal evaluation of both halves, bit-identical)
                    pred = den.forward_nhwc(x, nb, f, torch.full((nb,), float(t)), cross, halves_identical=self.share_first_layers,
                                            two_queues=self.two_queues)
                    ops.window_accumulate(pred, noise_sum, counter, win_dev[wi], f, F_, HW, halves=nb)
                    reader.clear()
                    writer.clear()
                if window_parallel is not None:
                    window_parallel.reduce(noise_sum, counter)
                a_t, a_prev = sch.step_coefficients(t)
                z = None
                if eta > 0:
                    from .scheduler import randn_tensor
                    zn = randn_tensor(latents.shape, generator=generator, device=dev, dtype=latents.dtype)      # (1, 4, F, h, w)
                    zs = zn.stride()
                    z = ops.pack_nhwc(zn, F_, F_, (0, zs[2], zs[1], zs[3], zs[4]), 0, c, 4, hh, ww)
                ops.cfg_ddim_step(lat, noise_sum, counter, F_, HW, guidance_scale, a_t, a_prev, halves=nb, eta=float(eta), variance_noise=z)
                if callback is not None and step_i % callback_steps == 0:
                    callback(step_i, t, self._latents_out(lat, latents))
        finally:
            refu.skip_dead_tail = False
            reader.clear()
            writer.clear()
            den.clear_context_cache()
            refu.clear_context_cache()
        return self._latents_out(lat, latents)

    def _latents_out(self, lat, like):
        out = torch.empty(like.shape, device=like.device, dtype=like.dtype)
        _, c, F_, hh, ww = like.shape
        so = out.stride()
        ops.unpack_nhwc(lat, out, F_, F_, (0, so[2], so[1], so[3], so[4]), c, hh, ww)
        return out

    def _write_banks(self, writer, reader, embeds, ref_latents, win_long, f, do_cfg, literal):
        """Run the reference UNet in write mode for one window and hand its banks to the reader blocks."""
        refu = self.reference_unet
        dev = ref_latents.device
        g = ref_latents[0].index_select(0, win_long)                         # (f, 22, h, w)
        if literal and do_cfg:
            g = g.repeat(2, 1, 1, 1)                                         # [uncond f | cond f] (:636-643)
            index = [k % 2 for k in range(2 * f)]                            # embeds.repeat((f,1,1)) = [u,c,u,c,...] (:645)
        elif do_cfg:
            index = [(f + j) % 2 for j in range(f)]                          # what cond frame j sees under that interleaving
        else:
            index = [0] * f
        B, nch, hh, ww = g.shape
        st = g.stride()
        x = ops.pack_nhwc(g, B, 1, (st[0], 0, st[1], st[2], st[3]), 0, nch - 2, 64, hh, ww)

        def motion_at(h2, w2):
            return ops.pack_nhwc(g, B, 1, (st[0], 0, st[1], st[2], st[3]), nch - 2, 2, 64, h2, w2, hin=hh, win=ww)

        cross = refu._cross(embeds, index, dev)
        refu.forward_nhwc(x, motion_at, cross)
        reader.update(writer)

    # ------------------------------------------------------------------------------------------ VAE glue (caller's modules)
    def _encode(self, tensor):
        return self.vae.encode(tensor.to(dtype=self.vae.dtype, device=self.vae.device)).latent_dist.mean * 0.18215

    dedupe_encodes = True      # False: every image goes through the VAE, duplicates included (A/B and the bit-identity test)

    @staticmethod
    def _unique_images(x):
        """x (N, 3, H, W) on one device -> (rep, inverse): rep = indices of the first occurrence of every DISTINCT image, inverse[i] =
        position in rep of image i's representative.  Exact: a cheap per-image signature (two partial sums) only proposes candidates,
        membership is decided by an element-wise comparison with the candidate (one batched compare and one host sync per signature)."""
        n = x.shape[0]
        flat = x.reshape(n, -1)
        sig = torch.stack([flat.sum(1, dtype=torch.float32), flat[:, 1::3].sum(1, dtype=torch.float32)], 1).cpu().tolist()
        by_sig = {}
        for i, s in enumerate(sig):
            by_sig.setdefault(tuple(s), []).append(i)
        rep, inverse = [], [0] * n
        for idxs in by_sig.values():
            todo = idxs
            while todo:                                                  # a signature collision leaves several distinct images in one group
                r = todo[0]
                same = (flat[todo] == flat[r]).all(1).cpu().tolist() if len(todo) > 1 else [True]
                same[0] = True                                           # the representative itself (an image holding NaNs is not == itself)
                pos = len(rep)
                rep.append(r)
                for i, eq in zip(todo, same):
                    if eq:
                        inverse[i] = pos
                todo = [i for i, eq in zip(todo, same) if not eq]
        order = sorted(range(len(rep)), key=lambda k: rep[k])            # representatives in input order (stable batches)
        rank = {k: j for j, k in enumerate(order)}
        return [rep[k] for k in order], [rank[k] for k in inverse]

    def _encode_many(self, tensors):
        """The reference encodes the 3F + 2 condition images one at a time (:456-549).  Two result-preserving reductions:
          * the VAE is per-image arithmetic (its GroupNorms and its attention never mix samples), so batches of `vae_batch` images give
            the same latents with an eighth of the launches and full-size GEMM / conv tiles;
          * an image that occurs several times is encoded ONCE and its latent copied: when face / hand guidance is absent the script
            substitutes F black frames each (scripts/inference_video.py:156-180), i.e. 2F of the 3F + 2 inputs of configs[1] are the same
            image (32 of 50 encodes at F = 16).  Deterministic kernels + per-image arithmetic make the copy bit-identical to a second
            encode (tests/test_vae_cpu.py::test_deduped_encodes_are_bit_identical on the host side, tests/test_vae_gpu.py on the HIP path)."""
        dev, dt = self.vae.device, self.vae.dtype
        x = torch.cat([t.to(device=dev, dtype=dt) for t in tensors], dim=0)
        if self.dedupe_encodes and x.shape[0] > 1:
            rep, inverse = self._unique_images(x)
        else:
            rep, inverse = list(range(x.shape[0])), list(range(x.shape[0]))
        self.last_encode_stats = dict(images=x.shape[0], encoded=len(rep))
        u = x if len(rep) == x.shape[0] else x[torch.tensor(rep, device=x.device)]
        lat = torch.cat([self._encode(u[i:i + self.vae_batch]) for i in range(0, u.shape[0], self.vae_batch)], dim=0)
        return lat if len(rep) == x.shape[0] else lat[torch.tensor(inverse, device=lat.device)]

    def decode_latents(self, latents):
        """reference :115-130 -- per-frame VAE decode, (x/2+0.5).clamp(0,1), float32 numpy (b,c,f,h,w)."""
        video_length = latents.shape[2]
        latents = 1 / 0.18215 * latents
        latents = latents.permute(0, 2, 1, 3, 4).reshape((-1,) + tuple(latents.shape[1:2]) + tuple(latents.shape[3:]))
        video = [self.vae.decode(latents[i:i + self.vae_batch].to(self.vae.dtype)).sample for i in range(0, latents.shape[0], self.vae_batch)]
        video = torch.cat(video)                                             # per-frame arithmetic: batching changes nothing
        video = video.reshape((-1, video_length) + tuple(video.shape[1:])).permute(0, 2, 1, 3, 4)
        video = (video / 2 + 0.5).clamp(0, 1)
        return video.cpu().float().numpy()

    def decode_temporal(self, latents, decode_chunk_size=None):
        """reference :132-150 (AutoencoderKLTemporalDecoder in chunks of self.decode_chunk_size = 16 frames)."""
        decode_chunk_size = decode_chunk_size or self.decode_chunk_size
        video_length = latents.shape[2]
        latents = 1 / 0.18215 * latents
        latents = latents.permute(0, 2, 1, 3, 4).reshape((-1,) + tuple(latents.shape[1:2]) + tuple(latents.shape[3:]))
        video = []
        for i in range(0, latents.shape[0], decode_chunk_size):
            chunk = latents[i:i + decode_chunk_size]
            video.append(self.vae.decode(chunk.to(self.vae.dtype), num_frames=chunk.shape[0]).sample)
        video = torch.cat(video)
        video = video.reshape((-1, video_length) + tuple(video.shape[1:])).permute(0, 2, 1, 3, 4)
        video = (video / 2 + 0.5).clamp(0, 1)
        return video.cpu().float().numpy()

    def interpolate_latents(self, latents, interpolation_factor, device):
        """reference :317-360 -- (F - 1) * factor + 1 frames: every original frame, and factor - 1 blends between neighbours
        (method chosen with set_tensor_interpolation_method; a few elementwise ops on 1 MB of latents, once per clip)."""
        if interpolation_factor < 2:
            return latents
        method = get_tensor_interpolation_method()
        if method is None:
            raise TypeError("interpolate_latents: call set_tensor_interpolation_method(is_slerp) first (the reference's module-level "
                            "`tensor_interpolation` is None until then, src/pipelines/utils.py:3-12)")
        b, c, f, h, w = latents.shape
        out = torch.zeros((b, c, (f - 1) * interpolation_factor + 1, h, w), device=latents.device, dtype=latents.dtype)
        rate = [i / interpolation_factor for i in range(interpolation_factor)][1:]
        idx = 0
        v1 = None
        for i0 in range(f - 1):
            v0, v1 = latents[:, :, i0], latents[:, :, i0 + 1]
            out[:, :, idx] = v0
            idx += 1
            for r in rate:
                out[:, :, idx] = method(v0.to(device=device), v1.to(device=device), r).to(latents.device)
                idx += 1
        out[:, :, idx] = v1
        return out

    def clip_embeds(self, ref_image):
        """reference :406-416 -- all 257 tokens: last_hidden_state -> post_layernorm -> visual_projection.
        `image_encoder` is mikudance_amd.CLIPVisionModelWithProjection (HIP kernels) or any module with the transformers
        surface (`(pixel_values).last_hidden_state`, `.vision_model.post_layernorm`, `.visual_projection`)."""
        from .clip_vision import clip_preprocess
        clip_image = clip_preprocess(ref_image.resize((224, 224)))             # == CLIPImageProcessor().preprocess(...).pixel_values
        enc = self.image_encoder
        px = clip_image.to(self._device, dtype=enc.dtype)
        if hasattr(enc, "image_prompt_embeds"):
            return enc.image_prompt_embeds(px)
        emb = enc(px).last_hidden_state
        return enc.visual_projection(enc.vision_model.post_layernorm(emb))

    # ------------------------------------------------------------------------------------------ reference-compatible call
    @torch.no_grad()
    def __call__(self, ref_image, ref_skel_image, tgt_pose_images, tgt_face_images, tgt_hand_images, scene_motion_npy, width,
                 height, video_length, num_inference_steps, guidance_scale, num_images_per_prompt=1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None, output_type: Optional[str] = "tensor",
                 return_dict: bool = True, callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
                 callback_steps: Optional[int] = 1, context_schedule="uniform", context_frames=None, context_stride=1,
                 context_overlap=8, context_batch_size=1, interpolation_factor=1, **kwargs):
        # context_batch_size: the reference concatenates that many windows along the batch axis (:601-622).  With one window per
        # context batch (every clip of <= context_frames frames, whatever the value) that is the evaluation below; with two or
        # more windows in a batch the reference itself fails at `noise_pred[:, :, c] + pred` (:662, batch 2 vs 2k), so there is
        # no behaviour to reproduce: the windows are evaluated one at a time here, which is what the sum over a batch would be.
        if context_batch_size < 1:
            raise ValueError(f"context_batch_size must be >= 1, got {context_batch_size}")
        if context_batch_size > 1 and not getattr(self, "_warned_context_batch", False):
            import warnings
            warnings.warn("context_batch_size > 1: the windows of a context batch are evaluated one at a time (the reference itself "
                          "fails for two or more windows per batch, pipeline_mikudance.py:662); results are those of context_batch_size = 1")
            self._warned_context_batch = True
        height = height or 768
        width = width or 768
        device = self._execution_device
        do_cfg = guidance_scale > 1.0
        batch_size = 1
        image_prompt_embeds = self.clip_embeds(ref_image)
        if do_cfg:
            image_prompt_embeds = torch.cat([torch.zeros_like(image_prompt_embeds), image_prompt_embeds], dim=0)
        latents = self.prepare_latents(batch_size * num_images_per_prompt, self.denoising_unet.in_channels, width, height,
                                       video_length, image_prompt_embeds.dtype, device, generator)
        f = video_length
        rep = lambda z: z.unsqueeze(1).repeat(1, f, 1, 1, 1).reshape((-1,) + tuple(z.shape[1:]))
        # all 3F + 2 condition images in ONE pass through the VAE (reference :456-549 encodes them one by one, same arithmetic per image)
        groups = [list(tgt_pose_images), list(tgt_face_images), list(tgt_hand_images)]
        lat_all = self._encode_many([_pil_to_tensor(ref_image, height, width, True), _pil_to_tensor(ref_skel_image, height, width, False)]
                                    + [_pil_to_tensor(im, height, width, False) for grp in groups for im in grp])
        ref_image_latents, pose_ref_latents = rep(lat_all[0:1]), rep(lat_all[1:2])
        o = [2]
        for grp in groups:
            o.append(o[-1] + len(grp))
        pose_tgt, face_tgt, hand_tgt = (lat_all[a:b] for a, b in zip(o[:-1], o[1:]))
        tracker = torch.from_numpy(np.asarray(scene_motion_npy)).to(dtype=ref_image_latents.dtype, device=ref_image_latents.device)
        ref_latents = torch.cat([ref_image_latents, pose_ref_latents, pose_tgt, face_tgt, hand_tgt, tracker], dim=1)[None]
        latents = self.denoise(latents, ref_latents, image_prompt_embeds, num_inference_steps, guidance_scale, context_schedule,
                               context_frames, context_stride, context_overlap, callback, callback_steps, eta=eta, generator=generator)
        if interpolation_factor > 0:
            latents = self.interpolate_latents(latents, interpolation_factor, device)
        images = self.decode_temporal(latents) if self.video_decoder else self.decode_latents(latents)
        if output_type == "tensor":
            images = torch.from_numpy(images)
        if not return_dict:
            return images
        return MikuDanceVideoPipelineOutput(videos=images)

"""What the first and the last pipeline stage do around the denoising steps, as the reference's demo arranges it
(``/root/reference/scripts/generate_video_demo.py``): ``encode_image`` on rank 0 (``:92-151``) and ``decode_latents`` on
the last rank (``:154-195``), on the HIP engines of ``clip_hip.py`` / ``vae_hip.py``.

Same argument meaning and return values as the script's functions, minus what is host-side image handling there: the
``CLIPImageProcessor`` output (``pixel_values``) and the normalised image tensor are arguments instead of a PIL image.
"""

from __future__ import annotations

import torch

from .clip_hip import CLIPVisionHIP
from .vae_hip import ImageEncoderHIP, TemporalDecoderHIP


def _read_local_checkpoint(model_dir: str, subfolder: str) -> tuple[dict, dict]:
    """``(state_dict, config)`` of ``<model_dir>/<subfolder>`` in the hub layout (``*.safetensors`` -- the ``.fp16.`` variant
    when both are shipped -- and ``config.json``).  Local directories only: there is no network to fetch a model name."""
    import json
    import os

    sub = os.path.join(model_dir, subfolder)
    if not os.path.isdir(sub):
        raise ValueError(f"'{model_dir}' holds no '{subfolder}' directory: pass a LOCAL checkpoint directory in the hub "
                         f"layout (a model name cannot be fetched: no network access)")
    from safetensors.torch import load_file

    files = sorted(n for n in os.listdir(sub) if n.endswith(".safetensors"))
    fp16_files = [n for n in files if ".fp16." in n]
    sd = {}
    for name in (fp16_files or files):
        sd.update(load_file(os.path.join(sub, name)))
    if not sd:
        raise ValueError(f"no *.safetensors weights under '{sub}'")
    cfg = {}
    cfg_path = os.path.join(sub, "config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path) as fh:
            cfg = json.load(fh)
    return sd, cfg


def load_edge_engines(model_dir: str, device="cuda"):
    """``(clip, vae_encoder, vae_decoder)`` from a local Stable Video Diffusion checkpoint directory, as the reference's
    demo loads them by id (``CLIPVisionModelWithProjection.from_pretrained(.., subfolder="image_encoder")``,
    ``AutoencoderKLTemporalDecoder.from_pretrained(.., subfolder="vae")``: ref ``scripts/generate_video_demo.py:248-262``)."""
    from .clip_hip import CLIPVisionSpec
    from .vae_hip import VAEDecoderConfig

    csd, ccfg = _read_local_checkpoint(model_dir, "image_encoder")
    d = CLIPVisionSpec()
    spec = CLIPVisionSpec(ccfg.get("hidden_size", d.hidden_size), ccfg.get("intermediate_size", d.intermediate_size),
                          ccfg.get("num_hidden_layers", d.num_hidden_layers), ccfg.get("num_attention_heads", d.num_attention_heads),
                          ccfg.get("image_size", d.image_size), ccfg.get("patch_size", d.patch_size),
                          ccfg.get("projection_dim", d.projection_dim), ccfg.get("layer_norm_eps", d.layer_norm_eps),
                          ccfg.get("hidden_act", d.hidden_act))
    clip = CLIPVisionHIP(spec, csd, device)
    vsd, vcfg = _read_local_checkpoint(model_dir, "vae")
    v = VAEDecoderConfig()
    cfg = VAEDecoderConfig(latent_channels=vcfg.get("latent_channels", v.latent_channels),
                           out_channels=vcfg.get("out_channels", v.out_channels),
                           block_out_channels=tuple(vcfg.get("block_out_channels", v.block_out_channels)),
                           layers_per_block=vcfg.get("layers_per_block", v.layers_per_block),
                           scaling_factor=vcfg.get("scaling_factor", v.scaling_factor))
    enc = ImageEncoderHIP(cfg, {k: t for k, t in vsd.items() if k.startswith(("encoder.", "quant_conv."))}, device)
    dec = TemporalDecoderHIP(cfg, {k[len("decoder."):]: t for k, t in vsd.items() if k.startswith("decoder.")}, device)
    return clip, enc, dec


def encode_image(pixel_values: torch.Tensor, image_tensor: torch.Tensor, image_encoder: CLIPVisionHIP,
                 vae_encoder: ImageEncoderHIP, num_frames: int, noise: torch.Tensor | None = None,
                 noise_aug_strength: float = 0.0) -> tuple[torch.Tensor, torch.Tensor]:
    """``(image_embeddings (B,1,1024), image_latents (B,4,F,H/8,W/8))`` as ``StableVideoUNet.set_conditioning`` takes
    them.  ``pixel_values``: ``feature_extractor(images=image).pixel_values`` (ref ``:108-109``); ``image_tensor``:
    ``Normalize([0.5],[0.5])(ToTensor(image))`` (ref ``:117-124``).  Noise augmentation in PIXEL space (ref ``:126-129``):
    the reference draws ``randn_like`` on its device; here the caller passes the ``noise`` tensor it wants added (device
    RNG streams differ between platforms), scaled by ``noise_aug_strength``.  The latents are the distribution's mode,
    NOT multiplied by the scaling factor (ref ``:139-142``)."""
    dev = image_encoder.device
    emb = image_encoder(pixel_values.to(dev, torch.float16).contiguous()).unsqueeze(1)
    img = image_tensor.to(dev, torch.float16)
    if noise is not None and noise_aug_strength > 0:
        img = torch.add(img, noise.to(dev, torch.float16), alpha=noise_aug_strength)    # input preparation, once per video
    latents = vae_encoder.encode_image_latents(img.contiguous(), num_frames)
    return emb, latents


def decode_latents(latents: torch.Tensor, vae: TemporalDecoderHIP, num_frames: int,
                   decode_chunk_size: int = 14) -> torch.Tensor:
    """(B, 4, F, H, W) latents -> (B, 3, F, 8H, 8W) fp32 frames (ref ``:154-195``)."""
    return vae.decode_latents(latents.to(vae.device, torch.float16).contiguous(), num_frames,
                              decode_chunk_size=decode_chunk_size)


class FrameEmitter:
    """``decode_latents`` wired into the step pipeline, so that the node emits frames (ref
    ``scripts/generate_video_demo.py:418`` decodes every finished latent on the LAST rank, after the step loop).

    The temporal-VAE decode of one video costs about as much as a whole stage of an 8-GPU pipeline (three UNet steps).
    Every rank decodes what FINISHES on it, on a HIP stream of its own BESIDE its UNet steps, and never moves a finished
    latent:

      * ring schedule (``bench.py``'s default at N > 1): sample ``i`` finishes on rank ``(i mod N) - 1`` and is decoded
        right there, so per video each GPU carries 1/N of a decode;
      * chain of stages: everything finishes, and is decoded, on the last rank -- the reference's arrangement.

    Until round 4 the chain could also spread its decodes (sample ``i`` on rank ``i mod N``, the last rank forwarding the
    latent with ``isend(tag=7001)``).  That was removed: RCCL ignores tags and orders the un-batched P2P of a rank PAIR on
    one internal stream, so on the pair (N-2, N-1) the forwards shared a communicator with the stage hand-offs and, with
    two samples interleaved per rank, the two sides could issue the two directions in different orders (last rank:
    recv, recv, send -- rank N-2: recv, send, send) and wait for each other for ever.  Over Gloo (tags honoured, host
    threads) this cannot show.  Now no message exists whose order could cross: ``spread`` is accepted for older call
    sites and means "decode where the sample finishes", which balances the ring and leaves the chain on its last rank.

    Attach to a stage on EVERY rank (same arguments), run the pipeline, then ``finish(num_samples)``:

        emitter = FrameEmitter(decoder, stage, num_frames)            # every rank
        stage.run_many(K, input_supplier=...); stage.drain()
        frames = emitter.finish(K)                                    # {sample index: (B,3,F,8H,8W) fp32} decoded HERE
    """

    def __init__(self, decoder: TemporalDecoderHIP, stage, num_frames: int, *, decode_chunk_size: int = 14,
                 spread: bool = True, keep: str = "all", check_finite: bool = False) -> None:
        from ..pipeline.step_assignment import ring_finish_rank

        if keep not in ("all", "last", "none"):
            raise ValueError("keep must be 'all', 'last' or 'none'")
        self.decoder, self.stage, self.num_frames = decoder, stage, num_frames
        self.chunk, self.keep, self.check_finite = decode_chunk_size, keep, check_finite
        cfg = stage.config
        self.rank, self.world = cfg.rank, cfg.world_size
        self.ring = bool(cfg.ring and cfg.world_size > 1)
        self.spread = bool(spread and self.ring)         # decodes are spread exactly when the schedule spreads the finishes
        self.device = decoder.device
        self.stream = torch.cuda.Stream(device=self.device)          # decodes
        self._ring_finish_rank = ring_finish_rank
        self.frames: dict[int, torch.Tensor] = {}
        self.stats = {"decoded": 0, "forwarded": 0, "received": 0}   # (forwarded / received stay 0: kept for older readers)
        stage.finished_latent_hook = self._finished
        stage.after_sample_hook = None

    # ---------------------------------------------------------------- who decodes sample i
    def decoder_rank(self, idx: int) -> int:
        """The rank on which sample idx's last step runs."""
        return self._ring_finish_rank(idx, self.world) if self.ring else self.world - 1

    # ---------------------------------------------------------------- hook
    def _finished(self, idx: int, latent: torch.Tensor) -> None:
        """On the rank (and stream) where sample idx's last step was just enqueued."""
        if self.decoder_rank(idx) != self.rank:
            raise RuntimeError(f"sample {idx} finished on rank {self.rank}, the schedule names rank {self.decoder_rank(idx)}")
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        self._decode(idx, latent, ready)

    def _decode(self, idx: int, latent: torch.Tensor, ready) -> None:
        latent.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            out = self.decoder.decode_latents(latent.contiguous(), self.num_frames, decode_chunk_size=self.chunk)
        self.stats["decoded"] += 1
        if self.keep == "all":
            self.frames[idx] = out
        elif self.keep == "last":
            self.frames = {idx: out}

    # ---------------------------------------------------------------- end of a run
    def finish(self, num_samples: int) -> dict:
        """Wait for this rank's decodes and return ``{sample index: frames}`` for the samples decoded on THIS rank
        (``keep``: all of them, the last one, or none)."""
        self.stream.synchronize()
        if self.check_finite:
            for idx, out in self.frames.items():
                if not bool(torch.isfinite(out).all()):
                    raise FloatingPointError(f"frames of sample {idx} hold non-finite values (fp16 activation range)")
        return self.frames

    def reset(self) -> None:
        """Forget the previous run's bookkeeping (bench.py: warm-up, then the timed region)."""
        self.frames = {}

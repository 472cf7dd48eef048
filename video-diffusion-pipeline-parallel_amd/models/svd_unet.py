"""``StableVideoUNet`` – the ``model(latent, step)`` adapter of the SVD path, MI355X-native.

API mirror of ``/root/reference/src/models/svd_unet.py`` (same constructor / method names, argument
meaning and errors):

  ``__init__`` ``:42-75``  ``_init_scheduler`` ``:77-102``  ``from_pretrained`` ``:104-164``
  ``init_noise_sigma`` ``:196-199``  ``_default_timestep_schedule`` ``:201-217``
  ``set_conditioning`` ``:219-279``  ``set_dummy_conditioning`` ``:281-338``
  ``clear_conditioning`` ``:340-349``  ``forward`` ``:351-439``

What differs is the execution: ``unet`` is a :class:`SVDUNetHIP` (hand-written gfx950 kernels), the
input scale / concat / permute and the fp32 v-prediction Euler update (+ per-frame CFG mix) are two
fused HIP kernels (``sp_pack_input_f16`` / ``sp_euler_step_f16``), and nothing synchronises the device.
``step`` is an INDEX into the sigma table exactly as in the reference (``svd_unet.py:377-379``), so the
caller decides the order.
"""

from __future__ import annotations

import math
import os
from collections.abc import Sequence

import torch
import torch.nn as nn

from . import euler_schedule
from .unet_hip import SVDUNetHIP
from .unet_spec import UNetConfig, random_state_dict


class StableVideoUNet(nn.Module):
    def __init__(
        self,
        unet,
        timesteps: Sequence[int],
        dtype: torch.dtype = torch.float16,
        num_train_timesteps: int = 1000,
        batched_cfg: bool = False,
    ) -> None:
        """``batched_cfg`` (extension): run the unconditional and conditional passes of
        classifier-free guidance as ONE batch-2 UNet forward instead of two sequential passes
        (ref ``svd_unet.py:384-411`` runs them one after the other)."""
        super().__init__()
        self.batched_cfg = batched_cfg
        self._use_graphs = os.environ.get("VDPP_GRAPHS", "0") == "1"
        self._graphs: dict = {}          # (calling stream, step, latent shape) -> (graph, static_in, static_out)
        self._graph_lanes: dict = {}     # calling stream -> (capture stream, memory pool) of that lane
        if dtype != torch.float16:
            raise ValueError("the MI355X SVD path computes in float16 (fp32 accumulate)")
        if not isinstance(unet, SVDUNetHIP):
            # any module carrying a diffusers-named UNetSpatioTemporalConditionModel state_dict
            cfg = getattr(unet, "hip_config", None) or UNetConfig.svd()
            device = next(unet.parameters()).device
            unet = SVDUNetHIP(cfg, unet.state_dict(), device)
        self.unet = unet
        self.timesteps = list(timesteps)
        self.dtype = dtype
        self.num_train_timesteps = num_train_timesteps
        self._init_scheduler()
        self._image_embeddings = None
        self._added_time_ids = None
        self._image_latents = None
        self._conditioning_set = False
        self._guidance_scale = None
        self._uncond_embeddings = None
        self._uncond_image_latents = None
        self._guidance_scale_tensor = None
        self._guidance32 = None

    # ------------------------------------------------------------------ schedule
    def _init_scheduler(self) -> None:
        sig = euler_schedule.karras_sigma_table(len(self.timesteps))
        self.sigmas = sig                                   # host fp32 (N+1)
        self.scheduler_timesteps = euler_schedule.continuous_timesteps(sig)
        self._sigma_host = [float(s) for s in sig]
        self._t_dev = self.scheduler_timesteps.to(self.unet.device)   # device table, indexed per step
        self._init_noise_sigma = float((sig[0] ** 2 + 1) ** 0.5)

    @property
    def init_noise_sigma(self) -> float:
        return self._init_noise_sigma

    @staticmethod
    def _default_timestep_schedule(num_steps: int, num_train_timesteps: int = 1000) -> list[int]:
        ratio = num_train_timesteps // num_steps
        return list(range(num_train_timesteps - 1, -1, -ratio))[:num_steps]

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(
        cls,
        model_id: str = "stabilityai/stable-video-diffusion-img2vid-xt",
        timesteps: Sequence[int] | None = None,
        torch_dtype: torch.dtype = torch.float16,
        enable_memory_efficient_attention: bool = True,
        enable_sliced_attention: bool = False,
        attention_slice_size: int | str = "auto",
        device="cuda",
        **kwargs,
    ) -> "StableVideoUNet":
        """Load ``<model_id>/unet/diffusion_pytorch_model*.safetensors`` from a LOCAL directory.

        The attention toggles of the reference signature are accepted and ignored: the fused HIP
        attention kernels are always on.  A hub name cannot be fetched (no network): ValueError.
        """
        unet_dir = os.path.join(model_id, "unet")
        if not os.path.isdir(unet_dir):
            raise ValueError(
                f"'{model_id}' is not a local model directory; use StableVideoUNet.from_random_init() for "
                "synthetic weights (there is no network access to fetch checkpoints)."
            )
        import json

        from safetensors.torch import load_file

        files = sorted(n for n in os.listdir(unet_dir) if n.endswith(".safetensors"))
        fp16_files = [n for n in files if ".fp16." in n]       # the hub layout ships both variants: read one of them
        sd = {}
        for name in (fp16_files or files):
            sd.update(load_file(os.path.join(unet_dir, name)))
        if not sd:
            raise ValueError(f"no *.safetensors weights under '{unet_dir}'")
        cfg = UNetConfig.svd()
        cfg_path = os.path.join(unet_dir, "config.json")
        if os.path.exists(cfg_path):              # diffusers' UNetSpatioTemporalConditionModel config
            with open(cfg_path) as fh:
                raw = json.load(fh)
            heads = raw.get("num_attention_heads", cfg.num_attention_heads)
            boc = tuple(raw.get("block_out_channels", cfg.block_out_channels))
            if isinstance(heads, int):
                heads = (heads,) * len(boc)
            down_types = raw.get("down_block_types")
            cfg = UNetConfig(
                in_channels=raw.get("in_channels", cfg.in_channels),
                out_channels=raw.get("out_channels", cfg.out_channels),
                block_out_channels=boc,
                layers_per_block=raw.get("layers_per_block", cfg.layers_per_block),
                num_attention_heads=tuple(heads),
                cross_attention_dim=raw.get("cross_attention_dim", cfg.cross_attention_dim),
                addition_time_embed_dim=raw.get("addition_time_embed_dim", cfg.addition_time_embed_dim),
                projection_class_embeddings_input_dim=raw.get("projection_class_embeddings_input_dim",
                                                              cfg.projection_class_embeddings_input_dim),
                down_has_attn=tuple("CrossAttn" in t for t in down_types) if down_types else cfg.down_has_attn,
            )
        if timesteps is None:
            timesteps = cls._default_timestep_schedule(num_steps=25)
        return cls(unet=SVDUNetHIP(cfg, sd, device), timesteps=timesteps, dtype=torch_dtype)

    @classmethod
    def from_random_init(cls, timesteps: Sequence[int], *, config: UNetConfig | None = None, seed: int = 0,
                         device="cuda", fp8_attention: bool | None = None,
                         long_attention: bool | None = None) -> "StableVideoUNet":
        """Random weights of the exact SVD architecture (benchmarks / tests; no checkpoint needed)."""
        cfg = config or UNetConfig.svd()
        sd = random_state_dict(cfg, seed=seed, device=device, dtype=torch.float16)
        unet = SVDUNetHIP(cfg, sd, device, fp8_attention=fp8_attention, long_attention=long_attention)
        del sd
        return cls(unet=unet, timesteps=timesteps)

    def enable_graphs(self, enabled: bool = True) -> None:
        """Replay each diffusion step from a captured HIP graph (one graph per step index and latent shape).

        A step is ~1,100 kernel launches issued from Python; the graph removes that host work from the critical path
        (it matters when the host is slow or the latent is small; at the benchmark shape the GPU is the bottleneck
        either way).  Graphs, their static buffers, their memory pool and the engine's per-stream scratch are all
        private to the HIP stream the caller runs on, so several videos in flight on separate streams
        (``PipelineConfig.concurrent_samples``) never share replay state.  Conditioning changes invalidate them."""
        self._use_graphs = enabled
        if not enabled:
            self._graphs.clear()
            self._graph_lanes.clear()
            release = getattr(self.unet, "release_stream_state", None)
            if release is not None:
                release()

    def enable_memory_optimizations(self) -> None:
        """Kept for API compatibility (ref ``svd_unet.py:166-194``); nothing to toggle here."""

    def to(self, *args, **kwargs):  # weights already live on the engine's device
        return self

    # ------------------------------------------------------------------ conditioning
    def set_conditioning(
        self,
        image_embeddings: torch.Tensor,
        image_latents: torch.Tensor,
        fps: int = 6,
        motion_bucket_id: int = 127,
        noise_aug_strength: float = 0.02,
        guidance_scale: float | None = None,
        num_frames: int = 14,
    ) -> None:
        if image_embeddings.dim() == 2:
            image_embeddings = image_embeddings.unsqueeze(1)
        batch = image_embeddings.shape[0]
        dev = self.unet.device
        ids = torch.tensor([[fps - 1, motion_bucket_id, noise_aug_strength]], dtype=self.dtype, device=dev)
        self._added_time_ids = ids.repeat(batch, 1)
        # every row is the same triple by construction (ref svd_unet.py:252-259 builds it the same way), so the engine
        # evaluates the added-time embedding once per call; SVDUNetHIP.__call__ refuses rows that differ
        self._added_ids32 = ids[0].float().contiguous()
        self._image_embeddings = image_embeddings.to(dev, self.dtype).contiguous()
        self._image_latents = image_latents.to(dev, self.dtype).contiguous()
        self._conditioning_set = True
        self._num_frames = int(num_frames)
        self._graphs.clear()          # captured graphs hold pointers to the previous conditioning tensors
        self._guidance_scale = guidance_scale
        if guidance_scale is not None and guidance_scale > 1.0:
            self._uncond_embeddings = torch.zeros_like(self._image_embeddings)
            self._uncond_image_latents = torch.zeros_like(self._image_latents)
            gs = torch.linspace(1.0, guidance_scale, num_frames)
            self._guidance_scale_tensor = gs.view(1, 1, num_frames, 1, 1).to(dev, dtype=self.dtype)
            self._guidance32 = self._guidance_scale_tensor.flatten().float().contiguous()
        else:
            self._uncond_embeddings = None
            self._uncond_image_latents = None
            self._guidance_scale_tensor = None
            self._guidance32 = None

    def set_dummy_conditioning(
        self,
        batch_size: int,
        num_frames: int,
        height: int,
        width: int,
        device: torch.device,
        fps: int = 6,
        motion_bucket_id: int = 127,
        noise_aug_strength: float = 0.02,
        guidance_scale: float | None = None,
    ) -> None:
        emb = torch.randn(batch_size, 1, self.unet.cfg.cross_attention_dim, device=device, dtype=self.dtype)
        lat = torch.randn(batch_size, 4, num_frames, height, width, device=device, dtype=self.dtype)
        self.set_conditioning(emb, lat, fps=fps, motion_bucket_id=motion_bucket_id,
                              noise_aug_strength=noise_aug_strength, guidance_scale=guidance_scale,
                              num_frames=num_frames)

    def clear_conditioning(self) -> None:
        self._image_embeddings = None
        self._added_time_ids = None
        self._image_latents = None
        self._conditioning_set = False
        self._guidance_scale = None
        self._uncond_embeddings = None
        self._uncond_image_latents = None
        self._guidance_scale_tensor = None
        self._guidance32 = None
        self._graphs.clear()
        self._graph_lanes.clear()            # capture streams / pools are keyed by the calling stream's raw handle
        release = getattr(self.unet, "release_stream_state", None)
        if release is not None:
            release()

    # ------------------------------------------------------------------ one diffusion step
    def _unet_pass(self, latent, image_latents, embeddings, in_scale, step, euler=None):
        from ..hip import ops

        b, _, f, h, w = latent.shape
        rows = torch.empty((b * f * h * w, self.unet.cin_pad), dtype=torch.float16, device=latent.device)
        ops.pack_input(latent, image_latents, rows, in_scale=in_scale, b=b, frames=f, h=h, w=w,
                       cpad=self.unet.cin_pad)
        return self.unet.forward_rows(rows, b=b, frames=f, h=h, w=w, t_value=self._t_dev[step:step + 1],
                                      ctx16=embeddings.reshape(b, -1), added_ids32=self._added_ids32, euler=euler)

    @torch.inference_mode()
    def forward(self, latent: torch.Tensor, step: int) -> torch.Tensor:
        from ..hip import ops

        if not self._conditioning_set:
            raise RuntimeError(
                "Conditioning not set. Call set_conditioning() or set_dummy_conditioning() before forward()."
            )
        if not (0 <= step < len(self.timesteps)):
            raise ValueError(f"Step {step} out of range [0, {len(self.timesteps)})")
        if latent.dtype != torch.float16 or not latent.is_cuda:
            raise ValueError("latent must be a float16 tensor on the HIP device")
        self._check_shapes(latent)
        latent = latent.contiguous()
        if self._use_graphs:
            return self._forward_graph(latent, step)
        return self._forward_eager(latent, step)

    def _check_shapes(self, latent: torch.Tensor) -> None:
        """The kernels take raw pointers and the latent's (B, F, H, W): a conditioning tensor of another shape would be
        read out of bounds (the reference fails in ``torch.cat`` / on broadcast, ``svd_unet.py:385-411``)."""
        if latent.dim() != 5 or latent.shape[1] != 4:
            raise ValueError(f"latent must be (B, 4, F, H, W); got {tuple(latent.shape)}")
        if tuple(self._image_latents.shape) != tuple(latent.shape):
            raise ValueError(f"image_latents {tuple(self._image_latents.shape)} do not match the latent "
                             f"{tuple(latent.shape)} (set_conditioning was called for another batch / frame count / size)")
        emb = self._image_embeddings
        if emb.dim() != 3 or emb.shape[0] != latent.shape[0] or emb.shape[1] != 1 \
                or emb.shape[2] != self.unet.cfg.cross_attention_dim:
            raise ValueError(f"image_embeddings must be (B, 1, {self.unet.cfg.cross_attention_dim}) with B = "
                             f"{latent.shape[0]}; got {tuple(emb.shape)}")
        if self._guidance32 is not None and self._guidance32.numel() != latent.shape[2]:
            raise ValueError(f"guidance was set for num_frames={self._guidance32.numel()}, the latent has "
                             f"{latent.shape[2]} frames")

    def _forward_graph(self, latent: torch.Tensor, step: int) -> torch.Tensor:
        # Everything a replay touches is keyed by the CALLING stream (one lane of PipelineStage's interleave = one
        # stream): its own graph, static input/output, memory pool, and - because the engine keys its GroupNorm / fp8
        # scratch by the stream it is enqueued on - its own capture stream.  Two lanes replaying at once therefore
        # share nothing but the (read-only) weights and conditioning tensors.
        lane = torch.cuda.current_stream(latent.device).cuda_stream
        key = (lane, step, tuple(latent.shape))
        entry = self._graphs.get(key)
        if entry is None:
            self._forward_eager(latent, step)             # warm-up: lazy allocations, function attributes
            torch.cuda.synchronize(latent.device)
            lane_state = self._graph_lanes.get(lane)
            if lane_state is None:
                lane_state = self._graph_lanes[lane] = [torch.cuda.Stream(device=latent.device), None]
            static_in = latent.clone()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=lane_state[1], stream=lane_state[0]):
                static_out = self._forward_eager(static_in, step)
            if lane_state[1] is None:
                lane_state[1] = graph.pool()
            entry = self._graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        static_in.copy_(latent)
        graph.replay()
        return static_out.clone()                          # the static buffer is overwritten by this lane's next replay

    def _forward_eager(self, latent: torch.Tensor, step: int) -> torch.Tensor:
        from ..hip import ops

        b, _, f, h, w = latent.shape
        sigma, sigma_next = self._sigma_host[step], self._sigma_host[step + 1]
        in_scale = 1.0 / math.sqrt(sigma * sigma + 1.0)

        eps_u = None
        guided = self._guidance_scale is not None and self._guidance_scale > 1.0
        if guided and self.batched_cfg:
            both = self._unet_pass(torch.cat([latent, latent], dim=0),
                                   torch.cat([self._uncond_image_latents, self._image_latents], dim=0),
                                   torch.cat([self._uncond_embeddings, self._image_embeddings], dim=0), in_scale, step)
            half = both.shape[0] // 2
            eps_u, eps_c = both[:half], both[half:]
        else:
            # sequential passes (the reference's order): the conditional pass' last convolution applies the guidance
            # mix and the Euler update in its epilogue, so its eps rows never exist in HBM
            out = torch.empty_like(latent)
            tail = dict(latent=latent, out=out, sigma=sigma, sigma_next=sigma_next)
            if guided:
                eps_u = self._unet_pass(latent, self._uncond_image_latents, self._uncond_embeddings, in_scale, step)
                tail.update(eps_uncond=eps_u, guidance=self._guidance32, ld_eps=eps_u.shape[1])
            self._unet_pass(latent, self._image_latents, self._image_embeddings, in_scale, step, euler=tail)
            return out
        out = torch.empty_like(latent)
        ops.euler_step(latent, eps_c, eps_u, self._guidance32 if eps_u is not None else None, out,
                       ld_eps=eps_c.shape[1], sigma=sigma, sigma_next=sigma_next, b=b, frames=f, h=h, w=w)
        return out

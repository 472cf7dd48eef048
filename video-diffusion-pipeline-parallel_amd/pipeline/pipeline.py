"""Per-rank stage executor of the diffusion-step pipeline.

API mirror of ``/root/reference/src/pipeline/pipeline.py`` (symbol -> reference lines):

  ``LatentSpec``                 ``:25-34``     ``PipelineStage.run``            ``:100-111``
  ``PipelineConfig``             ``:37-48``     ``PipelineStage.run_many``       ``:113-132``
  ``PipelineStage.__init__``     ``:57-70``     ``_process_single_latent``       ``:134-157``
  ``_recv_latent/_send_latent``  ``:75-84``     ``run_single_latent``            ``:160-185``
  ``_run_local_steps``           ``:86-98``     ``run_pipeline_latents``         ``:188-208``

Same arguments, same return values, same ``ValueError`` / ``RuntimeError`` conditions, same
``[rank=N] ...`` INFO log prefix, and the same quirk that the *timestep value* is what gets
passed to ``model(latent, step)`` (ref ``:87-95``).

What is different is the transport on a GPU rank.  The reference blocks the host in
``dist.send`` / ``dist.recv`` on the compute stream.  Here, when the latent lives on a HIP
device, hand-offs go over RCCL point-to-point (``torch.distributed`` backend ``"nccl"`` on ROCm)
from a dedicated side stream:

* send: an event is recorded on the compute stream after the last local step, the side
  stream waits on it and issues ``isend``; the compute stream is free to start the next
  sample's first UNet step immediately.
* recv: ``run_many`` pre-posts the ``irecv`` of sample ``i+1`` (into its own fresh buffer) while
  sample ``i`` computes; the compute stream only waits on the event that marks its own latent as landed.
  The pre-posted receive is ordered behind the START OF SAMPLE ``i``'s LAST LOCAL STEP, not posted a whole stage
  ahead: an RCCL receive whose sender is not there yet is a resident kernel spinning on its channels' workgroups,
  and the persistent GEMM (``csrc/gemm_ps.hip``) wants one 160-KB-LDS workgroup on every CU.
* ``PipelineConfig.concurrent_samples = S`` (extension): ``run_many`` keeps S samples in flight on S HIP
  streams and issues their UNet steps round-robin, so kernels of independent videos fill each other's idle
  CUs (micro-batched stage).

No collective is involved; one 1-2 MB message per stage boundary per sample travels over the
xGMI link between neighbouring ranks.  CPU / Gloo ranks (the simulator path) keep the
reference's blocking behaviour, which is what the golden vectors in ``tests/golden`` pin.

Extension fields (not in the reference): ``PipelineConfig.balanced`` selects ``assign_steps_balanced`` so 25
steps can be split over 2/4/8 ranks; ``rotate`` additionally rotates the stages that own the extra step with the
sample index (``assign_steps_rotating``), so no stage is a permanent bottleneck.  ``ring`` (``run_many`` only) turns
the chain of stages into a ring: sample ``i`` starts on rank ``i mod N`` and visits ranks ``i, i+1, ...`` (mod N) for
stages ``0 .. N-1``, so every rank computes from the first moment (no pipeline fill or drain bubble) and every rank
runs every stage once per N samples (perfectly balanced for any split); see ``PipelineStage._run_many_ring``.
"""

from __future__ import annotations

import logging
import os
import time
from collections import deque
from collections.abc import Sequence
from dataclasses import dataclass
from typing import Callable, Deque, Optional

import torch
import torch.distributed as dist

from .step_assignment import (StepRange, assign_steps, assign_steps_balanced, assign_steps_rotating, ring_finish_rank,
                              ring_sample, stage_sizes)

LOGGER = logging.getLogger(__name__)


@dataclass(frozen=True)
class LatentSpec:
    """Shape / dtype / device of the latent every stage boundary carries."""

    shape: torch.Size
    dtype: torch.dtype
    device: torch.device

    def empty(self) -> torch.Tensor:
        return torch.empty(self.shape, dtype=self.dtype, device=self.device)


@dataclass(frozen=True)
class PipelineConfig:
    total_steps: int
    world_size: int
    rank: int
    timesteps: Sequence[int]
    latent_spec: LatentSpec
    send_tag: int = 0
    # --- extensions beyond the reference dataclass -------------------------------------
    balanced: bool = False          # use assign_steps_balanced (uneven contiguous split)
    async_comm: Optional[bool] = None  # None = auto (side-stream RCCL when latent is on a GPU)
    concurrent_samples: int = 1     # run_many: samples interleaved on separate HIP streams of this rank
    rotate: bool = False            # balanced split whose "+1 step" stages rotate with the sample index
    ring: bool = False              # run_many: ring schedule (sample i starts on rank i mod N); needs a supplier on every rank

    def __post_init__(self) -> None:
        if len(self.timesteps) != self.total_steps:
            raise ValueError("len(timesteps) must equal total_steps.")


InputSupplier = Callable[[int], torch.Tensor]


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def _gloo_moves_gpu_tensor(t: torch.Tensor) -> bool:
    """Gloo stages a GPU tensor through the host with no regard for the HIP stream that is still producing it (found
    with tools/pp_equivalence.py: rank 1 received the latent of two steps earlier).  RCCL sends are stream-ordered;
    for the Gloo rehearsal mode (PIPELINE_BACKEND=gloo with GPU latents) the producing stream is drained first."""
    return t.is_cuda and dist.is_initialized() and dist.get_backend() == "gloo"


class _SideStreamLink:
    """RCCL send/recv on a side HIP stream, fenced against the compute stream(s) by events.

    Every incoming latent lands in its own freshly allocated buffer (1-2 MB from the caching allocator), so a
    pre-posted receive can never overwrite a latent that an earlier sample's first step is still reading, whatever
    compute stream that sample runs on.  Sent tensors are kept alive until their ``isend`` retires.
    """

    def __init__(self, spec: LatentSpec, rank: int, tag: int) -> None:
        self.spec = spec
        self.rank = rank
        self.tag = tag
        self.stream = torch.cuda.Stream(device=spec.device)
        self._pending: Deque[tuple] = deque()      # (work, buffer) of pre-posted irecvs, in message order
        self._in_flight: Deque[tuple] = deque()
        # Gloo reads and writes a GPU tensor from host threads with no regard for HIP streams (see
        # _gloo_moves_gpu_tensor): in that rehearsal mode (PIPELINE_BACKEND=gloo VDPP_ASYNC_COMM=1, ranks sharing a card)
        # the events that RCCL would wait for on the side stream are waited for on the host instead.  Same call sequence,
        # same buffers, same bookkeeping between two real processes; only the overlap is lost.
        self.host_ordered = dist.is_initialized() and dist.get_backend() == "gloo"
        # True: the receive of the next sample is posted right AFTER this sample's send instead of at the start of its
        # last local step (bench.py sets it when its start-up probe finds that a parked receive from rank-1 holds up a
        # send to rank+1, i.e. when both directions share one internal RCCL stream)
        self.post_after_send = False
        self.stats = {"recv_posted": 0, "recv_taken": 0, "sent": 0, "recv_posted_ahead": 0}

    # -- receive -------------------------------------------------------------------------
    def post_recv(self) -> None:
        """Enqueue an ``irecv`` on the side stream, ordered behind everything the CURRENT stream holds at this moment
        (so the call site decides how early the receive kernel may become resident on the GPU)."""
        buf = self.spec.empty()
        buf.record_stream(self.stream)
        allocated = torch.cuda.Event()
        allocated.record(torch.cuda.current_stream(self.spec.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(allocated)      # the allocator may hand back memory still in use on this stream
            if self.host_ordered:
                allocated.synchronize()
            work = dist.irecv(buf, src=self.rank - 1, tag=self.tag)
        self._pending.append((work, buf))
        self.stats["recv_posted"] += 1

    def take(self) -> torch.Tensor:
        """Return the next latent; the *current compute stream* is made to wait for it, not the host."""

        if not self._pending:
            self.post_recv()
        else:
            self.stats["recv_posted_ahead"] += 1
        self.stats["recv_taken"] += 1
        work, buf = self._pending.popleft()
        with torch.cuda.stream(self.stream):
            work.wait()  # stream-level dependency on the RCCL recv
            landed = torch.cuda.Event()
            landed.record(self.stream)
        consumer = torch.cuda.current_stream(self.spec.device)
        consumer.wait_event(landed)
        buf.record_stream(consumer)      # it was allocated on whatever stream posted the receive
        return buf

    @property
    def posted(self) -> int:
        return len(self._pending)

    # -- send ----------------------------------------------------------------------------
    def send(self, latent: torch.Tensor) -> None:
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.spec.device))
        latent.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            if self.host_ordered:
                ready.synchronize()
            work = dist.isend(latent, dst=self.rank + 1, tag=self.tag)
        self._in_flight.append((work, latent))
        self.stats["sent"] += 1
        while self._in_flight and self._in_flight[0][0].is_completed():
            self._in_flight.popleft()

    def drain(self) -> None:
        while self._in_flight:
            work, _ = self._in_flight.popleft()
            work.wait()
        self.stream.synchronize()


class PipelineStage:
    """Execution + hand-off behaviour of one rank (one pipeline stage)."""

    def __init__(
        self,
        model,
        config: PipelineConfig,
        logger: logging.Logger | None = None,
    ) -> None:
        self.model = model
        self.config = config
        self.logger = logger or LOGGER
        splitter = assign_steps_balanced if config.balanced else assign_steps
        self.step_range: StepRange = splitter(
            total_steps=config.total_steps,
            world_size=config.world_size,
            rank=config.rank,
        )
        use_async = config.async_comm
        forced = os.environ.get("VDPP_ASYNC_COMM")
        if use_async is None and forced == "0":
            use_async = False  # escape hatch: reference-style blocking send/recv on the compute stream
        can_link = (config.latent_spec.device.type == "cuda" and config.world_size > 1 and dist.is_available()
                    and dist.is_initialized())
        if use_async is None:
            # RCCL by default; VDPP_ASYNC_COMM=1 also over Gloo with GPU latents (rehearsal of the link between real
            # processes on a box whose ranks share a card: _SideStreamLink.host_ordered)
            use_async = can_link and (dist.get_backend() == "nccl" or forced == "1")
        self._link: _SideStreamLink | None = (
            _SideStreamLink(config.latent_spec, config.rank, config.send_tag) if use_async else None
        )
        self._more_samples_expected = False
        # optional observer, called on the last rank right after a sample's final step has been enqueued
        # (on that sample's stream) with the sample index - used by bench.py to time completions with events
        self.sample_done_hook: Callable[[int], None] | None = None
        # edge-stage hooks (models/edge_stages.py::FrameEmitter; ref scripts/generate_video_demo.py:418 decodes on the last
        # rank): `finished_latent_hook(idx, latent)` on the rank where sample idx's last step ran, on that sample's
        # stream, right after the step was enqueued; `after_sample_hook(idx)` on EVERY rank once its share of sample idx
        # (steps + hand-off) has been issued
        self.finished_latent_hook: Callable[[int, torch.Tensor], None] | None = None
        self.after_sample_hook: Callable[[int], None] | None = None

    # ------------------------------------------------------------------ logging
    def _log(self, message: str) -> None:
        self.logger.info("[rank=%s] %s", self.config.rank, message)

    # ------------------------------------------------------------------ transport
    def _recv_latent(self) -> torch.Tensor:
        upstream = self.config.rank - 1
        self._log(f"waiting for latent from rank {upstream}")
        if self._link is not None:
            tensor = self._link.take()
        else:
            tensor = self.config.latent_spec.empty()
            dist.recv(tensor, src=upstream, tag=self.config.send_tag)
        self._log("received latent")
        return tensor

    def _send_latent(self, latent: torch.Tensor) -> None:
        downstream = self.config.rank + 1
        self._log(f"sending latent to rank {downstream}")
        if self._link is not None:
            self._link.send(latent)
        else:
            if _gloo_moves_gpu_tensor(latent):
                torch.cuda.current_stream(latent.device).synchronize()
            dist.send(latent, dst=downstream, tag=self.config.send_tag)

    # ------------------------------------------------------------------ compute
    def _owned_timesteps(self, sample_idx: int | None) -> list:
        """Timesteps this stage runs for one sample (fixed range, or the rotating balanced split)."""
        cfg = self.config
        rng = self.step_range
        if cfg.rotate:
            rng = assign_steps_rotating(cfg.total_steps, cfg.world_size, cfg.rank, sample_idx or 0)
        owned = list(cfg.timesteps[rng.start : rng.end])
        if len(owned) != rng.count:
            raise RuntimeError("Local timestep slice length mismatch with step range.")
        return owned

    def _run_local_steps(self, latent: torch.Tensor, sample_idx: int | None = None) -> torch.Tensor:
        owned = self._owned_timesteps(sample_idx)

        verbose = self.logger.isEnabledFor(logging.INFO)
        for pos, step in enumerate(owned):
            if (pos == len(owned) - 1 and self._link is not None and self.config.rank > 0
                    and self._more_samples_expected and self._link.posted == 0 and not self._link.post_after_send):
                self._link.post_recv()      # next sample's receive: resident no earlier than this last step
            began = time.time()
            latent = self.model(latent, step)  # the timestep VALUE is the argument (ref :95)
            if verbose:
                self._log(f"step {step} completed in {(time.time() - began) * 1000.0:.2f} ms")
        return latent

    # ------------------------------------------------------------------ public entry points
    def run(self, input_latent: torch.Tensor | None) -> torch.Tensor | None:
        """One latent through this stage.  Rank 0 supplies it, other ranks pass ``None``.

        Returns the final latent on the last rank and ``None`` elsewhere.
        """

        return self._process_single_latent(input_latent, sample_idx=None)

    def run_many(
        self,
        num_samples: int,
        *,
        input_supplier: InputSupplier | None = None,
    ) -> list[torch.Tensor] | None:
        if num_samples <= 0:
            raise ValueError("num_samples must be positive for pipeline execution")
        if self.config.ring and self.config.world_size > 1:
            return self._run_many_ring(num_samples, input_supplier)
        first_rank = self.config.rank == 0
        if first_rank and input_supplier is None:
            raise ValueError("rank 0 requires an input_supplier when processing multiple samples")

        if self.config.concurrent_samples > 1 and self.config.latent_spec.device.type == "cuda":
            return self._run_many_interleaved(num_samples, input_supplier)

        finished: list[torch.Tensor] = []
        for sample_idx in range(num_samples):
            self._more_samples_expected = sample_idx + 1 < num_samples
            result = self._process_single_latent(
                input_supplier(sample_idx) if first_rank else None,
                sample_idx=sample_idx,
            )
            if result is not None:
                finished.append(result)
        self._more_samples_expected = False
        return finished or None

    def _run_many_interleaved(self, num_samples: int, input_supplier) -> list[torch.Tensor] | None:
        """``concurrent_samples`` latents at a time, one HIP stream each, their UNet steps issued round-robin.

        A single video leaves CUs idle (tail rounds of tiles, HBM-bound norm kernels next to MFMA-bound GEMMs);
        kernels of independent videos on separate streams fill those gaps (measured +20 % videos/s with 3 streams
        on one MI355X).  Results are identical to the sequential order: samples never interact.
        """

        cfg = self.config
        dev = cfg.latent_spec.device
        if not hasattr(self, "_streams"):
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(cfg.concurrent_samples)]
        main = torch.cuda.current_stream(dev)
        first, last = cfg.rank == 0, cfg.rank == cfg.world_size - 1
        finished: list[torch.Tensor] = []
        for base in range(0, num_samples, cfg.concurrent_samples):
            group = list(range(base, min(base + cfg.concurrent_samples, num_samples)))
            if not first and self._link is not None:
                while self._link.posted < len(group):       # first group only; later ones were posted late, below
                    self._link.post_recv()
            following = min(cfg.concurrent_samples, num_samples - (base + len(group)))
            latents = []
            for j, idx in enumerate(group):
                st = self._streams[j]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    if first:
                        latents.append(input_supplier(idx).to(dev))
                    else:
                        self._more_samples_expected = False
                        latents.append(self._recv_latent())
            owned = [self._owned_timesteps(idx) for idx in group]
            rounds = max(len(o) for o in owned)
            for k in range(rounds):
                if k == rounds - 1 and not first and self._link is not None and not self._link.post_after_send:
                    with torch.cuda.stream(self._streams[0]):   # ordered behind lane 0's second-to-last step
                        for _ in range(following):
                            self._link.post_recv()
                for j in range(len(group)):
                    if k < len(owned[j]):
                        with torch.cuda.stream(self._streams[j]):
                            latents[j] = self.model(latents[j], owned[j][k])
            for j in range(len(group)):
                with torch.cuda.stream(self._streams[j]):
                    if last:
                        finished.append(latents[j])
                        if self.sample_done_hook is not None:
                            self.sample_done_hook(group[j])
                        if self.finished_latent_hook is not None:
                            self.finished_latent_hook(group[j], latents[j])
                    else:
                        self._send_latent(latents[j])
                if last:
                    latents[j].record_stream(main)
                    main.wait_stream(self._streams[j])
            if not first and self._link is not None and self._link.post_after_send:
                with torch.cuda.stream(self._streams[0]):
                    for _ in range(following):
                        self._link.post_recv()
            if self.after_sample_hook is not None:
                for idx in group:
                    self.after_sample_hook(idx)
            self._log(f"samples {group[0]}..{group[-1]} issued on {len(group)} streams")
        return finished or None

    def _run_many_ring(self, num_samples: int, input_supplier) -> list[torch.Tensor] | None:
        """Ring schedule (extension).  Samples are taken in batches of N = world_size; in slot ``s`` of a batch this
        rank runs stage ``s`` (the ``s``-th range of the balanced split) of the sample whose home rank is
        ``(rank - s) mod N``, then hands it to rank+1 and receives its slot ``s+1`` sample from rank-1.  In every slot
        all ranks run the same stage index, so slots line up and nobody waits for a pipeline to fill or drain.
        Transport (round 4): the SAME primitives as the chain of stages -- un-batched ``isend`` to rank+1 and ``irecv``
        from rank-1 (torch gives every rank pair its own communicator and stream), on a side stream behind events of
        the compute lanes -- so the ring needs nothing of RCCL that the chain does not; even ranks send first and odd
        ranks receive first, which is what keeps N = 2 (both neighbours are the same peer: one communicator, one
        stream) free of the send-waits-for-send deadlock and is harmless elsewhere.  ``concurrent_samples`` batches are
        interleaved on separate HIP streams: their kernels share the GPU, but the exchange is bulk-synchronous per slot
        (one exchange for all lanes, ordered behind every lane's last step of the slot); the 1-2 MB hand-off is
        microseconds on xGMI against >= 150 ms of stage compute.  Finished latents travel on to the last rank over the
        same neighbour links at the end (a sample that finished on rank f makes N-1-f hops), so the return contract is
        the chain's: list on the last rank, ``None`` elsewhere.  Every rank needs the ``input_supplier`` (it produces
        the inputs of its home samples)."""

        cfg = self.config
        n, r = cfg.world_size, cfg.rank
        if input_supplier is None:
            raise ValueError("ring schedule: every rank needs the input_supplier (samples start on rank i mod N)")
        dev = cfg.latent_spec.device
        cuda = dev.type == "cuda"
        sizes = stage_sizes(cfg.total_steps, n, balanced=True)
        starts = [sum(sizes[:j]) for j in range(n)]
        nxt_rank, prv_rank = (r + 1) % n, (r - 1) % n
        # interleave lanes: one HIP stream each on a GPU rank; on a CPU/Gloo rank they are plain bookkeeping (same
        # exchange pattern, so the lane logic is covered by the world_size 5/6/8 Gloo tests)
        conc = max(1, cfg.concurrent_samples)
        nbatch = (num_samples + n - 1) // n
        if cuda:
            if not hasattr(self, "_streams") or len(self._streams) < conc:
                self._streams = [torch.cuda.Stream(device=dev) for _ in range(conc)]
            if not hasattr(self, "_ring_stream"):
                self._ring_stream = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
        finished: dict[int, torch.Tensor] = {}

        def on(j):  # compute stream of interleave lane j (no-op context on CPU)
            return torch.cuda.stream(self._streams[j]) if cuda else _NullCtx()

        def neighbour_p2p(sends, recvs):
            """isend every tensor of `sends` to rank+1, irecv every tensor of `recvs` from rank-1 (un-batched: the chain's
            primitives); even ranks send first, odd ranks receive first.  Returns the works in issue order."""
            works = []
            first, second = (("s", sends), ("r", recvs)) if r % 2 == 0 else (("r", recvs), ("s", sends))
            for kind, tensors in (first, second):
                for t in tensors:
                    works.append(dist.isend(t, dst=nxt_rank, tag=cfg.send_tag) if kind == "s"
                                 else dist.irecv(t, src=prv_rank, tag=cfg.send_tag))
            return works

        def exchange(outgoing, incoming_lanes):
            """outgoing: [(lane, tensor)] to rank+1; incoming_lanes: lanes that receive from rank-1.
            Returns {lane: received tensor}, each lane's stream already ordered behind the transfer."""
            if not outgoing and not incoming_lanes:
                return {}
            got = {}
            if cuda:
                side = self._ring_stream
                for j, t in outgoing:
                    if _gloo_moves_gpu_tensor(t):
                        self._streams[j].synchronize()
                    ev = torch.cuda.Event(); ev.record(self._streams[j]); side.wait_event(ev)
                    t.record_stream(side)
                with torch.cuda.stream(side):
                    for j in incoming_lanes:
                        got[j] = cfg.latent_spec.empty()
                    if incoming_lanes and _gloo_moves_gpu_tensor(got[incoming_lanes[0]]):
                        side.synchronize()              # Gloo writes from the host: the buffers' memory must be free by now
                    for w in neighbour_p2p([t for _, t in outgoing], [got[j] for j in incoming_lanes]):
                        w.wait()                        # RCCL: stream-level (the side stream waits, not the host)
                    done = torch.cuda.Event(); done.record(side)
                for j in incoming_lanes:
                    self._streams[j].wait_event(done)
                    got[j].record_stream(self._streams[j])
                self._ring_keepalive = [t for _, t in outgoing]   # until the next exchange has been ordered behind this one
            else:
                for j in incoming_lanes:
                    got[j] = cfg.latent_spec.empty()
                for w in neighbour_p2p([t for _, t in outgoing], [got[j] for j in incoming_lanes]):
                    w.wait()
            return got

        for g0 in range(0, nbatch, conc):
            lanes = list(range(min(conc, nbatch - g0)))          # lane j <-> batch g0 + j
            cur: dict[int, torch.Tensor] = {}
            if cuda:
                for j in lanes:
                    self._streams[j].wait_stream(main)
            for s in range(n):
                vid = {j: ring_sample(r, g0 + j, s, n) for j in lanes}
                live = [j for j in lanes if vid[j] < num_samples]
                if s == 0:
                    for j in live:
                        with on(j):
                            cur[j] = input_supplier(vid[j]).to(dev)
                steps = list(cfg.timesteps[starts[s]: starts[s] + sizes[s]])
                if len(steps) != sizes[s]:
                    raise RuntimeError("Local timestep slice length mismatch with step range.")
                for step in steps:                                # round-robin over the interleaved samples
                    for j in live:
                        with on(j):
                            cur[j] = self.model(cur[j], step)
                if s == n - 1:
                    for j in live:
                        finished[vid[j]] = cur.pop(j)
                        if self.sample_done_hook is not None:
                            with on(j):
                                self.sample_done_hook(vid[j])
                        if self.finished_latent_hook is not None:
                            with on(j):
                                self.finished_latent_hook(vid[j], finished[vid[j]])
                    continue
                incoming = [j for j in lanes if ring_sample(r, g0 + j, s + 1, n) < num_samples]
                got = exchange([(j, cur[j]) for j in live], incoming)
                cur = got
            self._log(f"ring batches {g0}..{g0 + len(lanes) - 1} issued")
        if cuda:
            for st in self._streams[:conc]:
                main.wait_stream(st)
            for t in finished.values():
                t.record_stream(main)

        # ---- collect on the last rank: sample i finished on rank f = ring_finish_rank(i) and makes N-1-f hops over the
        # neighbour links r -> r+1 (never the wrap-around link: rank N-1 only receives, rank 0 only sends).  In round d every
        # rank forwards what it holds and receives what rank-1 holds, in sample order (both sides know the lists).
        last = n - 1
        holder = {i: ring_finish_rank(i, n) for i in range(num_samples)}
        have: dict[int, torch.Tensor] = dict(finished)
        host_staged = cuda and dist.is_initialized() and dist.get_backend() == "gloo"     # (see _gloo_moves_gpu_tensor)
        for _ in range(n - 1):
            moving = sorted(i for i, hr in holder.items() if hr != last)
            if not moving:
                break
            out_ids = [i for i in moving if holder[i] == r]
            in_ids = [i for i in moving if holder[i] == r - 1]
            bufs = {i: cfg.latent_spec.empty() for i in in_ids}
            if host_staged:
                torch.cuda.synchronize(dev)          # what is sent has been produced, what is received into is free
            works = []
            order = (("s", out_ids), ("r", in_ids)) if r % 2 == 0 else (("r", in_ids), ("s", out_ids))
            for kind, ids in order:
                for i in ids:
                    works.append(dist.isend(have[i], dst=r + 1, tag=cfg.send_tag) if kind == "s"
                                 else dist.irecv(bufs[i], src=r - 1, tag=cfg.send_tag))
            for w in works:
                w.wait()                             # RCCL: orders the current stream behind the transfer
            for i in out_ids:
                del have[i]                          # (the process group keeps a sent tensor alive until its send has run)
            have.update(bufs)
            for i in moving:
                holder[i] += 1
        return [have[i] for i in range(num_samples)] if r == last else None

    def _process_single_latent(
        self, input_latent: torch.Tensor | None, sample_idx: int | None
    ) -> torch.Tensor | None:
        label = "" if sample_idx is None else f"sample {sample_idx} "
        cfg = self.config

        if cfg.rank == 0:
            if input_latent is None:
                raise ValueError("rank 0 requires an input latent tensor")
            latent = input_latent.to(cfg.latent_spec.device)
            self._log(f"{label}input prepared")
        else:
            if input_latent is not None:
                raise ValueError("non-zero ranks should not receive an eager latent")
            latent = self._recv_latent()
            self._log(f"{label}received latent")

        latent = self._run_local_steps(latent, sample_idx)

        if cfg.rank == cfg.world_size - 1:
            self._log(f"{label}final rank completed")
            if self.sample_done_hook is not None and sample_idx is not None:
                self.sample_done_hook(sample_idx)
            if self.finished_latent_hook is not None and sample_idx is not None:
                self.finished_latent_hook(sample_idx, latent)
            if self.after_sample_hook is not None and sample_idx is not None:
                self.after_sample_hook(sample_idx)
            return latent

        self._send_latent(latent)
        if (self._link is not None and self._link.post_after_send and cfg.rank > 0 and self._more_samples_expected
                and self._link.posted == 0):
            self._link.post_recv()
        if self.after_sample_hook is not None and sample_idx is not None:
            self.after_sample_hook(sample_idx)
        return None

    def drain(self) -> None:
        """Wait for outstanding side-stream sends (no-op on the blocking transport)."""

        if self._link is not None:
            self._link.drain()

    @property
    def transport(self) -> dict:
        """What moves this stage's latents (bench.py prints it per rank): the side-stream link with its counters, or
        the reference's blocking send/recv."""
        if self.config.ring and self.config.world_size > 1:
            return {"kind": "ring: un-batched isend to rank+1 / irecv from rank-1 on a side stream (Gloo: on the host), "
                            "even ranks send first, odd ranks receive first"}
        if self._link is None:
            return {"kind": "blocking send/recv on the compute stream"}
        return dict(self._link.stats, kind="side-stream link", host_ordered=self._link.host_ordered,
                    post_after_send=self._link.post_after_send)


def _make_stage(model, *, total_steps, timesteps, world_size, rank, latent_spec, logger,
                balanced: bool = False) -> PipelineStage:
    config = PipelineConfig(
        total_steps=total_steps,
        world_size=world_size,
        rank=rank,
        timesteps=timesteps,
        latent_spec=latent_spec,
        balanced=balanced,
    )
    return PipelineStage(model=model, config=config, logger=logger)


def run_single_latent(
    model,
    *,
    total_steps: int,
    timesteps: Sequence[int],
    world_size: int,
    rank: int,
    latent_spec: LatentSpec,
    input_latent: torch.Tensor | None,
    logger: logging.Logger | None = None,
    balanced: bool = False,
) -> torch.Tensor | None:
    """All ranks call this once per latent; rank 0 passes the latent, the others ``None``."""

    stage = _make_stage(model, total_steps=total_steps, timesteps=timesteps, world_size=world_size,
                        rank=rank, latent_spec=latent_spec, logger=logger, balanced=balanced)
    out = stage.run(input_latent=input_latent)
    stage.drain()
    return out


def run_pipeline_latents(
    model,
    *,
    total_steps: int,
    timesteps: Sequence[int],
    world_size: int,
    rank: int,
    latent_spec: LatentSpec,
    num_samples: int,
    input_supplier: InputSupplier | None,
    logger: logging.Logger | None = None,
    balanced: bool = False,
) -> list[torch.Tensor] | None:
    stage = _make_stage(model, total_steps=total_steps, timesteps=timesteps, world_size=world_size,
                        rank=rank, latent_spec=latent_spec, logger=logger, balanced=balanced)
    out = stage.run_many(num_samples, input_supplier=input_supplier)
    stage.drain()
    return out

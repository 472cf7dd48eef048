#!/usr/bin/env python3
"""Headline benchmark: steady-state videos/s of SVD 14-frame x 25-step denoising on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" of this benchmark is ONE VIDEO: a synthetic 14-frame 576x1024 latent (1,4,14,72,128) fp16
pushed through all 25 diffusion steps of the SVD UNet; videos travel through the pipeline in micro-batches of two (one
pipeline sample = latent (2,4,14,72,128): --micro-batch) (random weights of the exact architecture, dummy
conditioning like the reference's ``set_dummy_conditioning``, no CFG = the reference benchmark default,
``/root/reference/src/modes/benchmark.py:60``).  With N > 1 the 25 steps are split into contiguous
stages over the ranks (balanced split, e.g. [4,3,3,3,3,3,3,3]) and latents move stage-to-stage by RCCL
send/recv on a side stream (``vdpp_amd.pipeline.PipelineStage``).

The JSON line carries: metric/value (K videos / max-over-ranks wall time of the barrier-bracketed timed
region), ``roofline`` for the dominant kernel (the implicit-GEMM MFMA kernel: algorithmic FLOPs of its
launches in one UNet forward / their summed HIP-event durations, vs the 2.5 PFLOP/s dense fp16 peak),
``step_roofline`` (whole UNet forward; both with ``clock_ghz_live`` = the shader clock rank 0's chip held over the timed
region, from stream-ordered stamps of s_memtime against the 100 MHz counter), ``cpu_baseline`` (the oracle's fp32 UNet restatement timed on the
host cores on a bounded sample, rank 0 at N=1 only) and ``cpu_simulator`` (the reference's CPU
simulator-mode path = DummyUNet step pipeline, as re-implemented here + the C oracle).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# NOTHING that can initialise the GPU is imported at module level: with `--gpus N` and no launcher the parent only
# starts N fresh ranks (launch_ranks, stdlib only) and relays their output; torch and the package are imported by
# _imports() inside the ranks (tests/test_bench_launcher_cpu.py pins this).
torch = dist = None


def _imports():
    global torch, dist, finalize_distributed, init_distributed, resolve_backend
    global LatentSpec, PipelineConfig, PipelineStage, stage_sizes
    import torch as _torch
    import torch.distributed as _dist

    import vdpp_amd  # noqa: F401
    from vdpp_amd.distributed import finalize_distributed, init_distributed, resolve_backend
    from vdpp_amd.pipeline import LatentSpec, PipelineConfig, PipelineStage, stage_sizes
    torch, dist = _torch, _dist


PEAK_FP16_TFLOPS = 2500.0   # MI355X dense fp16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
FRAMES, LAT_H, LAT_W, TOTAL_STEPS = 14, 72, 128, 25


class Watchdog:
    """Hard wall-clock guard for multi-rank runs: a rank that makes no progress for `limit` seconds prints where it
    was and leaves with exit status 3 (a fresh exit of this process; nothing is re-executed), so that a stuck
    collective or hand-off yields a diagnosable log instead of a silent hang until the launcher's own limit."""

    def __init__(self, limit: float, rank: int):
        import threading
        self.limit, self.rank = limit, rank
        self.where, self.last = "start", time.monotonic()
        self._stop = threading.Event()
        if limit > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def beat(self, where: str) -> None:
        self.where, self.last = where, time.monotonic()

    def stop(self) -> None:
        self._stop.set()

    def _run(self) -> None:
        while not self._stop.wait(1.0):
            idle = time.monotonic() - self.last
            if idle > self.limit:
                print(f"[rank {self.rank}] WATCHDOG: no progress for {idle:.0f} s at '{self.where}'; exiting with status 3",
                      file=sys.stderr, flush=True)
                status = os.environ.get("VDPP_BENCH_STATUS")       # read by this rank's supervisor (supervise_rank)
                if status:
                    try:
                        with open(status, "w") as fh:
                            fh.write(self.where)
                    except OSError:
                        pass
                os._exit(3)


def describe_rank(rank, n, device, ring, rotating, conc, selftest, transport=None, probe=None):
    """One stderr line per rank before the timed region: what a failed scaling run needs for its post-mortem."""
    backend = dist.get_backend() if n > 1 else "none"
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as exc:  # noqa: BLE001
        rccl = f"unavailable ({exc!r})"
    sched = "single GPU" if n == 1 else ("ring" if ring else "chain" + (" + rotating extra step" if rotating else ""))
    name = torch.cuda.get_device_name(device) if device.type == "cuda" else "host cores"
    # torch's ProcessGroupNCCL gives every un-batched isend/irecv PAIR its own two-rank communicator and stream, keyed
    # "low:high" (pointToPoint -> getKeySendRecv); grouped batch_isend_irecv uses the device-keyed world communicator
    keys = {"from_rank-1": f"{rank - 1}:{rank}" if rank > 0 else None, "to_rank+1": f"{rank}:{rank + 1}" if rank < n - 1 else None}
    print(f"[rank {rank}/{n}] device {device} ({name}), torch {torch.__version__}, "
          f"backend {backend}, RCCL {rccl}, schedule {sched}, {conc} videos in flight, ring self-test: {selftest}, "
          f"transport {transport}, p2p communicator keys {keys}, p2p probe {probe}, "
          f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, "
          f"NCCL_MAX_P2P_NCHANNELS={os.environ.get('NCCL_MAX_P2P_NCHANNELS')}", file=sys.stderr, flush=True)


def p2p_probe(rank, n, device, delay=0.4):
    """Does a receive from rank-1 that is parked (its sender has not sent yet) hold up this rank's send to rank+1?
    The chain posts the receive of video i+1 one UNet step BEFORE it sends video i (pipeline._SideStreamLink), which is
    only harmless if torch keeps the two directions on separate communicators / streams (it keys un-batched P2P by the
    rank pair; that is version behaviour, so it is MEASURED here once per run, N >= 3): rank 0 sends `delay` seconds
    late; every middle rank posts its receive first and sends at once; a rank whose message arrives at about `delay`
    instead of at about zero sat behind its upstream's parked receive.  Every rank returns the same dict; with
    `serialised` the caller posts receives after sends (PipelineStage link.post_after_send).  ~delay seconds, untimed."""
    if n < 3:
        return {"ran": False, "why": "needs >= 3 ranks"}
    cuda = device.type == "cuda"
    out_t = torch.full((256,), float(rank), device=device)
    in_t = torch.empty(256, device=device)
    side = torch.cuda.Stream(device=device) if cuda else None
    if cuda:
        torch.cuda.synchronize(device)      # Gloo reads GPU tensors from the host, outside any stream order

    def on_side():
        return torch.cuda.stream(side) if cuda else _Null()

    def round_trip(late):
        """irecv(rank-1) posted first, then isend(rank+1) (rank 0 `late` seconds late); seconds until this rank's message
        had landed.  work.wait() orders the side stream behind an RCCL transfer (it does not block the host), Gloo's blocks."""
        dist.barrier()
        t0 = time.perf_counter()
        works = []
        with on_side():
            if rank > 0:
                works.append(dist.irecv(in_t, src=rank - 1))
            if rank == 0 and late:
                time.sleep(late)
            if rank < n - 1:
                works.append(dist.isend(out_t, dst=rank + 1))
            for w in works:
                w.wait()
        arrived = 0.0
        if rank > 0:
            if cuda:
                side.synchronize()      # (a middle rank's send has completed by then as well: rank+1 posts at once)
            arrived = time.perf_counter() - t0
        if cuda:
            torch.cuda.synchronize(device)
        return arrived

    round_trip(0.0)      # creates the pair communicators (lazy, blocking the host until both peers call) outside the timing
    arrived = round_trip(delay)
    ok = rank == 0 or float(in_t[0]) == float(rank - 1)
    mine = {"rank": rank, "arrived_s": round(arrived, 4), "ok": bool(ok)}
    allr = [None] * n
    dist.all_gather_object(allr, mine)
    # rank 1 must wait for the late sender; ranks >= 2 must NOT (their upstream sent at once unless it was held up)
    held = [r["rank"] for r in allr if r["rank"] >= 2 and r["arrived_s"] > 0.5 * delay]
    return {"ran": True, "delay_s": delay, "arrived_s": [r["arrived_s"] for r in allr], "data_ok": all(r["ok"] for r in allr),
            "ranks_that_waited_behind_a_parked_receive": held, "serialised": bool(held)}


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


class _HostEvent:
    """torch.cuda.Event stand-in of --rehearse-cpu (CPU work is synchronous: record() is a timestamp)."""

    def __init__(self, enable_timing=True):
        self.t = None

    def record(self, *_):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)


def pmc_dominant_template():
    """MFMA-busy / wait fractions of the contraction template with the most time, from the committed PMC pass of two
    forwards (separate --pmc runs: tools/pmc_forward.sh -> profiles/*_two_forwards_pmc_per_kernel.txt); static, like
    `traffic`.  busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CU cycles), wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES."""
    for name in ("r05_a_two_forwards_pmc_per_kernel.txt", "r04_a_two_forwards_pmc_per_kernel.txt"):
        try:
            rows = [ln.split() for ln in open(os.path.join(ROOT, "profiles", name)) if ln.startswith("gemm_")]
        except OSError:
            continue
        best = None
        for r in rows:
            try:
                busy, wait = float(r[-2]), float(r[-1])
                ms = float(r[-6])
            except (ValueError, IndexError):
                continue
            if best is None or ms > best[0]:
                best = (ms, " ".join(r[:-7]), busy, wait)
        if best:
            return {"kernel": best[1], "mfma_busy": best[2], "wait": best[3], "ms_in_two_forwards": best[0],
                    "source": f"profiles/{name}", "static": True}
    return None


def usable_cores():
    """Host cores this process may actually use: the affinity mask, cut by the cgroup CPU quota if there is one."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    cores = min(cores, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return cores


def ring_selftest(rank, n, device):
    """The ring's hand-off with small payloads, exactly as pipeline._run_many_ring issues it: un-batched isend to rank+1
    and irecv from rank-1 on a side stream, even ranks sending first and odd ranks receiving first (N = 2: both directions
    share one communicator), twice (the first round creates the pair communicators).  True on every rank iff every rank
    received its neighbour's payload both times."""
    cuda = device.type == "cuda"
    ok = torch.ones(1, device=device)
    try:
        side = torch.cuda.Stream(device=device) if cuda else None
        for rnd in range(2):
            out_t = torch.full((1024,), float(100 * rnd + rank), device=device)
            in_t = torch.full((1024,), -1.0, device=device)
            if cuda:
                torch.cuda.synchronize(device)
            with (torch.cuda.stream(side) if cuda else _Null()):
                ops = [("s", out_t), ("r", in_t)] if rank % 2 == 0 else [("r", in_t), ("s", out_t)]
                works = [dist.isend(t, dst=(rank + 1) % n) if k == "s" else dist.irecv(t, src=(rank - 1) % n) for k, t in ops]
                for w in works:
                    w.wait()
            if cuda:
                side.synchronize()
                torch.cuda.synchronize(device)
            if float(in_t[0]) != float(100 * rnd + (rank - 1) % n) or float(in_t[-1]) != float(in_t[0]):
                ok.zero_()
    except Exception as exc:  # noqa: BLE001
        print(f"[rank {rank}] ring self-test failed ({exc!r}); using the chain schedule", file=sys.stderr, flush=True)
        ok.zero_()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return bool(ok.item() > 0)


def measure_in_flight_table(model, device, args, n, dog):
    """ms per video and UNet forward for (videos per call, streams) in {1,2}^2, timed on THIS job's GPUs during start-up
    (two rounds of one UNet step per stream after one warm round; ~3 s), max over ranks so that every rank chooses alike.
    None if anything goes wrong (the static table decides then)."""
    try:
        table = {}
        T = args.total_steps
        for b in (1, 2):
            torch.manual_seed(args.seed)
            model.set_dummy_conditioning(b, args.frames, args.height, args.width, device, guidance_scale=args.guidance_scale)
            lat = (torch.randn((b, 4, args.frames, args.height, args.width), device=device) * 10).half()
            streams = [torch.cuda.Stream(device=device) for _ in range(2)]
            for c in (1, 2):
                dog.beat(f"in-flight table {b}x{c}")
                times = []
                for rnd in range(3):
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    for st in streams[:c]:
                        with torch.cuda.stream(st), torch.no_grad():
                            model(lat, T // 2)
                    torch.cuda.synchronize(device)
                    times.append(time.perf_counter() - t0)
                table[(b, c)] = 1e3 * min(times[1:]) / (b * c)
        model.unet.release_stream_state()
        keys = sorted(table)
        t = torch.tensor([table[k] for k in keys], dtype=torch.float64, device=device)
        if n > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return {k: float(v) for k, v in zip(keys, t.tolist())}
    except Exception as exc:  # noqa: BLE001
        print(f"bench.py: in-flight table not measured ({exc!r}); using the static one", file=sys.stderr, flush=True)
        return None


def choose_in_flight(steps, n, ring, total_steps, args, measured=None):
    """(videos per UNet call, streams per GPU) with the smallest predicted time of the job.  More in flight raises a busy
    stage's rate and lengthens the chain's fill and drain (ring: leaves the last group of batches emptier).  The four
    per-forward times are a STATIC table (ms per video and UNet forward on one MI355X at 14 frames fp16, round 4,
    tools/batch_vs_streams.py -> profiles/r04_batch_vs_streams.txt; their ORDER is what matters and holds at 25 frames too)."""
    MS = {(1, 1): 51.9, (1, 2): 48.6, (2, 1): 48.5, (2, 2): 46.8}
    static = measured is None
    if measured is not None:
        MS = dict(measured)

    def predicted(b, c):
        per_stage = total_steps / n
        if not ring or n == 1:
            return (steps / (b * c) + n - 1) * b * c * per_stage * MS[(b, c)]
        nbatch = -(-(steps // b) // n)                  # batches of N samples
        full, rest = divmod(nbatch, c)                  # groups of c interleaved batches, then `rest` lanes
        t = full * n * per_stage * b * c * MS[(b, c)]
        if rest:
            t += n * per_stage * b * rest * MS[(b, rest)]
        return t

    cands = [(b_, c_) for (b_, c_) in MS if steps % b_ == 0
             and (args.micro_batch is None or args.micro_batch == b_) and (args.concurrent is None or args.concurrent == c_)]
    if cands:
        mb, conc = min(cands, key=lambda bc: predicted(*bc))
        how = ("static table of four per-forward times measured in round 4 (tools/batch_vs_streams.py, profiles/r04_batch_vs_streams.txt)"
               if static else "four per-forward times measured on this job's GPUs at start-up (max over ranks)") + ", smallest predicted job time"
    else:       # explicit values outside the table
        mb = args.micro_batch if args.micro_batch is not None else 1
        conc = max(1, args.concurrent if args.concurrent is not None else 2)
        how = "given on the command line"
    return mb, conc, how, {f"{b_}x{c_}": v for (b_, c_), v in MS.items()}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="videos in the timed region")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up videos")
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--height", type=int, default=LAT_H)
    ap.add_argument("--width", type=int, default=LAT_W)
    ap.add_argument("--total-steps", type=int, default=TOTAL_STEPS)
    ap.add_argument("--guidance-scale", type=float, default=None)
    ap.add_argument("--concurrent", type=int, default=None,
                    help="videos kept in flight per GPU on separate HIP streams (1 = the reference's sequential "
                         "order; default 2)")
    ap.add_argument("--micro-batch", type=int, default=None,
                    help="videos per pipeline sample = batch dimension of every UNet call (default 2 when --steps is even: "
                         "the 2,016- and 8,064-row levels fill the chip better, +3.5 %% videos/s measured; 1 = the "
                         "reference benchmark's batch_size=1, ref src/modes/benchmark.py:101)")
    ap.add_argument("--no-rotate", action="store_true",
                    help="N>1: keep the extra step of the balanced split on the first ranks for every video "
                         "(default: rotate it with the video index so no stage is a permanent bottleneck)")
    ap.add_argument("--ring", action="store_true",
                    help="(default at N > 1 since round 4; kept for older command lines)  Ring schedule: video i starts on "
                         "rank i mod N and visits every rank once, so no pipeline fill / drain sits inside the timed region.  "
                         "Same transport primitives as the chain (un-batched isend to rank+1 / irecv from rank-1); a "
                         "self-test of exactly those calls runs first and every rank falls back to the chain if it fails")
    ap.add_argument("--chain", "--no-ring", dest="chain", action="store_true",
                    help="N>1: the reference's chain of stages (rank 0 feeds, last rank finishes; the extra step of the "
                         "balanced split rotates with the video index unless --no-rotate) instead of the ring")
    ap.add_argument("--no-fallback", action="store_true",
                    help="N>1: run this rank in the launched process itself instead of under its supervisor (no second attempt "
                         "on the chain / the blocking transport when the first one stalls)")
    ap.add_argument("--watchdog", type=float, default=120.0,
                    help="seconds without progress after which a rank prints where it is and exits with status 3")
    ap.add_argument("--long-attention", action="store_true", help="(default since round 4; kept for older command lines)")
    ap.add_argument("--no-long-attention", action="store_true",
                    help="level-0 spatial attention through the ordinary kernel instead of the frozen-reference kernel "
                         "(csrc/attention_long.hip; its worst case is bounded at ~1.1-1.25x the ordinary kernel)")
    ap.add_argument("--fp8-attention", action="store_true",
                    help="spatial self-attention on fp8-e4m3 MFMA (BASELINE config 5; use with --frames 25 --total-steps 30)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true",
                    help="skip the edge-stage timings (temporal-VAE decode, CLIP / VAE image encode; reported beside the "
                         "headline metric, never inside it)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-batch1-leg", action="store_true",
                    help="skip the short one-video-at-a-time leg (`reference_order_batch1_one_at_a_time`; profiling runs)")
    ap.add_argument("--emit-frames", action="store_true",
                    help="also time the pipeline WITH its last edge stage (secondary figure `frames_out`, never `value`): every "
                         "finished latent is decoded to frames by the temporal VAE on a stream of its own beside the UNet "
                         "steps, on the rank where it finishes (models/edge_stages.py::FrameEmitter; ref "
                         "scripts/generate_video_demo.py:418).  Default at N = 1 unless --no-decode; opt-in at N > 1")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="NOT a measurement: run this script's whole multi-rank path (launcher, process group, rotating "
                         "chain / ring, watchdog, all_gather_object, steady-state arithmetic, JSON line) with the simulator's "
                         "DummyUNet on CPU tensors over Gloo, so that `--gpus 8` can be rehearsed on a box without 8 GPUs")
    ap.add_argument("--seed", type=int, default=42)
    return ap.parse_args()


def cpu_baseline(frames_full, h, w, total_steps):
    """Oracle fp32 UNet (plain PyTorch CPU) timed on this host: ONE full forward at the benchmark's own shape with
    min(usable cores, 64) threads (about 15-40 s of host work), x total_steps = seconds per video; nothing is
    extrapolated.  On a host with fewer than 12 usable cores (the build container) the sample is 4 of the frames, scaled
    by the algorithmic FLOP ratio, and the record says so."""
    from oracle.svd_unet_ref import SVDUNetConfig, SVDUNetRef, unet_flops

    cores = os.cpu_count() or 1
    usable = usable_cores()
    threads = max(1, min(usable, 64))
    torch.set_num_threads(threads)
    cfg = SVDUNetConfig.svd()
    with torch.device("meta"):
        ref = SVDUNetRef(cfg)
    ref = ref.to_empty(device="cpu").eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.fill_(0.01)
    full = usable >= 12
    sample_frames = frames_full if full else min(4, frames_full)
    x = torch.randn(1, sample_frames, 8, h, w)
    ctx = torch.randn(1, 1, cfg.cross_attention_dim)
    ids = torch.tensor([[5.0, 127.0, 0.02]])
    t0 = time.perf_counter()
    with torch.no_grad():
        ref(x, 1.0, ctx, ids)
    dt = time.perf_counter() - t0
    scale = unet_flops(cfg, frames_full, h, w)["total"] / unet_flops(cfg, sample_frames, h, w)["total"]
    videos_per_s = 1.0 / (dt * scale * total_steps)
    del ref
    return {"value": videos_per_s, "unit": "videos/s", "cores": threads, "host_cores": cores, "usable_cores": usable,
            "kind": "port", "extrapolated": not full, "forward_s": dt,
            "sample": f"oracle fp32 UNet (torch CPU, {threads} threads), 1 forward at {sample_frames} of {frames_full} frames "
                      f"{h}x{w} in {dt:.1f}s" + ("" if full else f", scaled by FLOP ratio {scale:.2f}")
                      + f", x {total_steps} steps per video"}


def _sim_worker(rank, ws, init_file, out_file, c, hid, shape, steps, reps, threads):
    import logging

    os.dup2(2, 1)       # Gloo announces its peers on stdout; the parent's stdout carries exactly one JSON line
    _imports()

    from vdpp_amd.models import DummyUNet
    from vdpp_amd.pipeline import run_single_latent

    torch.set_num_threads(threads)
    init_distributed(backend="gloo", rank=rank, world_size=ws, init_method=f"file://{init_file}")
    torch.manual_seed(1234)                      # same weights on every rank (the reference CLI forgets this)
    model = DummyUNet(c, hid)
    x = torch.randn(shape)
    spec = LatentSpec(shape=x.shape, dtype=torch.float32, device=torch.device("cpu"))
    ts = list(reversed(range(steps)))
    quiet = logging.getLogger("bench.quiet"); quiet.setLevel(logging.ERROR)
    kw = dict(total_steps=steps, timesteps=ts, world_size=ws, rank=rank, latent_spec=spec,
              input_latent=x if rank == 0 else None, logger=quiet)
    with torch.no_grad():
        run_single_latent(model, **kw)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            run_single_latent(model, **kw)
        dist.barrier()
        dt = (time.perf_counter() - t0) / reps
    if rank == 0:
        torch.save(dt, out_file)
    finalize_distributed()


def _sim_time(ws, c, hid, shape, steps, reps, threads):
    """Seconds per sample of the reference's simulator-mode path (ref src/modes/simulator.py:95-164: DummyUNet, Gloo,
    CPU) through this repo's executor, `ws` processes with `threads` torch threads each."""
    import tempfile

    import torch.multiprocessing as mp

    with tempfile.TemporaryDirectory() as td:
        out_file = os.path.join(td, "t.pt")
        mp.spawn(_sim_worker, args=(ws, os.path.join(td, "init"), out_file, c, hid, shape, steps, reps, threads),
                 nprocs=ws, join=True)
        return float(torch.load(out_file))


def cpu_simulator():
    """The reference's CPU simulator-mode path timed on this host (BASELINE.md section 3 / SURVEY 8d): DummyUNet(8,16) on
    (1,8,8,32,32) fp32, 8 steps, world_size 1 and 2 over Gloo; DummyUNet(4,64) on the SVD latent (1,4,14,72,128); and
    the scalar C oracle of the same arithmetic."""
    import numpy as np

    from oracle import dummy_ref
    from vdpp_amd.models import DummyUNet

    cores = os.cpu_count() or 1
    out = {"host_cores": cores, "rows": []}
    for ws, c, hid, shape, steps, reps in ((1, 8, 16, (1, 8, 8, 32, 32), 8, 5), (2, 8, 16, (1, 8, 8, 32, 32), 8, 5),
                                           (1, 4, 64, (1, 4, 14, 72, 128), 8, 2), (2, 4, 64, (1, 4, 14, 72, 128), 8, 2)):
        threads = max(1, min(usable_cores(), 16) // ws)
        dt = _sim_time(ws, c, hid, shape, steps, reps, threads)
        out["rows"].append({"workload": f"DummyUNet({c},{hid}) {tuple(shape)} fp32, {steps} steps", "world_size": ws,
                            "backend": "gloo", "threads_per_rank": threads, "samples_per_s": 1.0 / dt,
                            "ms_per_step": 1e3 * dt / steps})
    torch.manual_seed(1234)
    model = DummyUNet(8, 16)
    x = torch.randn(1, 8, 8, 32, 32)
    ts = list(reversed(range(8)))
    params = {k: v.numpy() for k, v in model.state_dict().items()}
    dummy_ref.run_steps(x.numpy(), ts, 0, 8, params)
    t0 = time.perf_counter()
    dummy_ref.run_steps(x.numpy(), ts, 0, 8, params)
    out["c_oracle"] = {"workload": "DummyUNet(8,16) (1,8,8,32,32) fp32, 8 steps (oracle/dummy_unet_ref.c, scalar)",
                       "threads": 1, "samples_per_s": 1.0 / (time.perf_counter() - t0)}
    return out


def vae_decode_leg(device, frames, h, w):
    """Temporal VAE decoder (random weights of the SVD architecture) on one synthetic latent video: ms per video."""
    from vdpp_amd.models.vae_hip import TemporalDecoderHIP, VAEDecoderConfig, param_count, random_state_dict

    cfg = VAEDecoderConfig.svd()
    dec = TemporalDecoderHIP(cfg, random_state_dict(cfg, seed=0), device)
    gen = torch.Generator(device=device).manual_seed(7)
    lat = (torch.randn((1, 4, frames, h, w), generator=gen, device=device) * cfg.scaling_factor).half()
    chunk = 14                                           # the reference's decode_chunk_size
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(device)           # peak of THIS leg, not of the process
    with torch.no_grad():
        dec.decode_latents(lat, frames, decode_chunk_size=chunk)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            vid = dec.decode_latents(lat, frames, decode_chunk_size=chunk)
        e1.record()
        torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / reps
    # multiply-add FLOPs of the decoder calls (same model as oracle/vae_temporal_decoder_ref.py::decoder_flops)
    fl = 0.0
    for i in range(0, frames, chunk):
        fl += _vae_flops(cfg, min(chunk, frames - i), h, w)
    res = {"decode_ms": ms, "tflop": fl / 1e12, "achieved_tflops": fl / 1e9 / ms, "frac_of_fp16_peak": fl / 1e9 / ms / PEAK_FP16_TFLOPS,
           "output": list(vid.shape), "output_dtype": "fp32", "decode_chunk_size": chunk, "params": param_count(cfg),
           "peak_memory_gb": round(torch.cuda.max_memory_allocated(device) / 2**30, 2),
           "note": "AutoencoderKLTemporalDecoder restated (random init, fp16 storage / fp32 accumulate); NOT part of "
                   "`value`: the benchmark's videos end as latents, like the reference benchmark's"}
    del dec, vid
    torch.cuda.empty_cache()
    return res


def frames_out_leg(args, stage, supplier, device, n, rank, ring, mb, conc, fence, dog):
    """The same pipeline with its last edge stage attached (secondary figure, never `value`): every finished pipeline
    sample is decoded to frames (B,3,F,8H,8W) fp32 by the temporal VAE (random weights of the SVD architecture) on a HIP
    stream of its own beside the UNet steps, on the rank where it finishes: with the ring schedule that is rank (i mod N) - 1,
    so no stage carries a whole decode per video; with the chain it is the last rank (ref scripts/generate_video_demo.py:418
    decodes there too, after the loop).  No finished latent is forwarded (models/edge_stages.py says why).  Barrier-bracketed
    like the headline figure; the last decode's completion is inside the timed region."""
    from vdpp_amd.models.edge_stages import FrameEmitter
    from vdpp_amd.models.vae_hip import TemporalDecoderHIP, VAEDecoderConfig, random_state_dict

    cfg = VAEDecoderConfig.svd()
    dec = TemporalDecoderHIP(cfg, random_state_dict(cfg, seed=0), device)
    emitter = FrameEmitter(dec, stage, args.frames, decode_chunk_size=14, spread=True, keep="last")
    hook = stage.sample_done_hook
    stage.sample_done_hook = None
    per = mb * conc
    videos = max(2 * per, 8 if n == 1 else 8 * n)
    videos = -(-videos // per) * per
    samples = videos // mb
    with torch.no_grad():
        stage.run_many(max(1, (n if n > 1 else 1) * conc), input_supplier=supplier if (rank == 0 or ring) else None)   # warm-up
        stage.drain()
        emitter.finish(max(1, (n if n > 1 else 1) * conc))
        emitter.reset()
        fence()
        t0 = time.perf_counter()
        stage.run_many(samples, input_supplier=(lambda i: supplier(1000 + i)) if (rank == 0 or ring) else None)
        stage.drain()
        emitter.finish(samples)
        fence()
        dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    decoded = [emitter.stats["decoded"]]
    if n > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        decoded = [None] * n
        dist.all_gather_object(decoded, emitter.stats["decoded"])
    dt = float(t.item())
    shape = None
    for fr in emitter.frames.values():
        shape = list(fr.shape)
    stage.finished_latent_hook = stage.after_sample_hook = None
    stage.sample_done_hook = hook
    del emitter, dec
    torch.cuda.empty_cache()
    return {"videos_per_s": videos / dt, "ms_per_video": 1e3 * dt / videos, "videos": videos,
            "decodes_per_rank_incl_warmup": decoded, "frames_per_sample": shape, "output_dtype": "fp32",
            "where": ("every sample is decoded on the rank where it finishes (" + ("ring: rank (i mod N) - 1" if ring else
                      "chain: the last rank") + "), on a HIP stream of its own beside the UNet steps")
                     if n > 1 else "on a HIP stream of its own beside the UNet steps of the following videos",
            "note": "temporal VAE decoder with random weights of the SVD architecture; NOT `value`: the reference benchmark's "
                    "videos end as latents (ref src/modes/benchmark.py), its demo decodes them on the last rank "
                    "(ref scripts/generate_video_demo.py:418)"}


def image_encode_leg(device, frames, h, w):
    """First stage's encode_image (ref scripts/generate_video_demo.py:92-151): CLIP ViT-H/14 image embeddings and the
    VAE encoder's image latents, random weights of the SVD architectures; ms per video (once per video, not per step)."""
    from vdpp_amd.models.clip_hip import CLIPVisionHIP, CLIPVisionSpec
    from vdpp_amd.models.vae_hip import ImageEncoderHIP, VAEDecoderConfig, random_encoder_state_dict

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1) / reps

    gen = torch.Generator(device=device).manual_seed(11)
    spec = CLIPVisionSpec.svd()
    c, L, I = spec.hidden_size, spec.num_hidden_layers, spec.intermediate_size
    sd = {}

    def rnd(*shape, std):
        return (torch.randn(shape, generator=gen, device=device) * std).half()

    sd["vision_model.embeddings.class_embedding"] = rnd(c, std=0.02)
    sd["vision_model.embeddings.patch_embedding.weight"] = rnd(c, 3, spec.patch_size, spec.patch_size, std=0.02)
    sd["vision_model.embeddings.position_embedding.weight"] = rnd((spec.image_size // spec.patch_size) ** 2 + 1, c, std=0.02)
    for n in ("pre_layrnorm", "post_layernorm"):
        sd[f"vision_model.{n}.weight"], sd[f"vision_model.{n}.bias"] = torch.ones(c).half(), torch.zeros(c).half()
    for i in range(L):
        p = f"vision_model.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[f"{p}.self_attn.{n}.weight"], sd[f"{p}.self_attn.{n}.bias"] = rnd(c, c, std=0.02), torch.zeros(c).half()
        for n in ("layer_norm1", "layer_norm2"):
            sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"] = torch.ones(c).half(), torch.zeros(c).half()
        sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"] = rnd(I, c, std=0.02), torch.zeros(I).half()
        sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"] = rnd(c, I, std=0.02), torch.zeros(c).half()
    sd["visual_projection.weight"] = rnd(spec.projection_dim, c, std=0.02)
    clip = CLIPVisionHIP(spec, sd, device)
    del sd
    px = torch.randn((1, 3, spec.image_size, spec.image_size), generator=gen, device=device).half()
    cfg = VAEDecoderConfig.svd()
    enc = ImageEncoderHIP(cfg, random_encoder_state_dict(cfg, seed=0), device)
    img = torch.randn((1, 3, 8 * h, 8 * w), generator=gen, device=device).clamp(-1, 1).half()
    with torch.no_grad():
        clip_ms = timed(lambda: clip(px))
        enc_ms = timed(lambda: enc.encode_image_latents(img, frames))
    res = {"clip_image_embeddings_ms": clip_ms, "vae_image_latents_ms": enc_ms,
           "image": [1, 3, 8 * h, 8 * w], "clip_pixel_values": [1, 3, spec.image_size, spec.image_size],
           "note": "CLIPVisionModelWithProjection (ViT-H/14) and the AutoencoderKLTemporalDecoder encoder, random init; once "
                   "per video on the first stage, NOT part of `value`"}
    del clip, enc
    torch.cuda.empty_cache()
    return res


def _vae_flops(cfg, frames, h, w):
    ch = list(cfg.block_out_channels)
    px = frames * h * w

    def res(cin, cout, px):
        return 2.0 * px * (9 * cin * cout + 9 * cout * cout + 2 * 3 * cout * cout + (cin * cout if cin != cout else 0))

    c = ch[-1]
    tot = 2.0 * px * 9 * cfg.latent_channels * c + 2 * res(c, c, px) + 4 * 2.0 * px * c * c + 2 * 2.0 * frames * (h * w) ** 2 * c
    prev = c
    for i, co in enumerate(reversed(ch)):
        for j in range(cfg.layers_per_block + 1):
            tot += res(prev if j == 0 else co, co, px)
        if i != len(ch) - 1:
            px *= 4
            tot += 2.0 * px * 9 * co * co
        prev = co
    return tot + 2.0 * px * 9 * ch[0] * cfg.out_channels + 2.0 * px * 3 * cfg.out_channels ** 2


def visible_gpus():
    """GPUs this process tree can use, WITHOUT touching the HIP runtime: KFD topology nodes with SIMDs, narrowed by the
    *_VISIBLE_DEVICES lists.  None if the topology cannot be read (the ranks themselves check again with torch)."""
    import glob
    nodes = 0
    try:
        for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            for line in open(prop):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    nodes += 1
    except OSError:
        return None
    if nodes == 0:
        return None if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes") else 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is not None:
            nodes = min(nodes, len([v for v in val.split(",") if v.strip() != ""]))
    return nodes


def p2p_env(env):
    """RCCL settings of the 1-2 MB stage-to-stage hand-off.  A receive posted ahead of its sender is a resident RCCL
    kernel that holds one workgroup per P2P channel; the persistent GEMM wants one 160-KB-LDS workgroup on every
    CU, so the hand-off gets two channels (a 1 MB message over one xGMI link needs no more) unless the caller says
    otherwise.  Returned dict = what was in force (printed by every rank)."""
    env.setdefault("NCCL_MAX_P2P_NCHANNELS", "2")
    env.setdefault("NCCL_MIN_P2P_NCHANNELS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC is the only one this driver supports
    return {k: env.get(k) for k in ("NCCL_MAX_P2P_NCHANNELS", "NCCL_MIN_P2P_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                    "HSA_ENABLE_IPC_MODE_LEGACY")}


# ---- the first RCCL run must not end without a number ------------------------------------------------------------------
# Every schedule / transport of the multi-rank path has only ever run over Gloo (1-GPU leases).  If the default (ring) stalls
# on RCCL, the watchdog ends every rank with status 3 -- and the one scaling measurement the driver takes would yield nothing.
# So with N > 1 every RANK process is a small supervisor (standard library only: it never touches the GPU) that runs the
# actual benchmark rank as a FRESH child process (never an exec), and when the attempt fails on any rank all supervisors
# start the next rung of the ladder together, with a fresh process group on a fresh port:
#     ring  ->  chain on the side-stream link  ->  chain with the reference's blocking send/recv (ref src/pipeline/pipeline.py:
#     75-84,134-157; `--chain`, `VDPP_ASYNC_COMM=0`)
# The supervisors agree through files in a directory under the temp dir keyed by their common parent (the torchrun agent).
# The driver launches `python -m torch.distributed.run ... bench.py --gpus N`, so this has to live in the ranks; the
# launcher-less `python bench.py --gpus N` reaches the same code through its own torchrun child (launch_ranks).
LADDER = (
    {"key": "ring", "schedule": "ring", "transport": "un-batched isend/irecv on a side stream", "flags": [], "env": {}},
    {"key": "chain", "schedule": "chain", "transport": "side-stream link (isend / pre-posted irecv)", "flags": ["--chain"], "env": {}},
    {"key": "blocking", "schedule": "chain", "transport": "blocking send/recv on the compute stream (the reference's)",
     "flags": ["--chain"], "env": {"VDPP_ASYNC_COMM": "0"}},
)
RC_USAGE = 2          # a worker that refuses its arguments: no other rung would accept them either


def _usage(msg):
    """Refused arguments / environment: exit status RC_USAGE, which the supervisors do not answer with another attempt."""
    print(msg, file=sys.stderr, flush=True)
    sys.exit(RC_USAGE)


def ladder_for(args, env):
    if env.get("VDPP_ASYNC_COMM") == "0":
        return [LADDER[2]]
    return list(LADDER[1:] if args.chain else LADDER)


def _proc_start_time(pid):
    try:
        with open(f"/proc/{pid}/stat") as fh:
            return fh.read().rsplit(")", 1)[1].split()[19]        # field 22: starttime in clock ticks since boot
    except (OSError, IndexError):
        return "0"


def supervise_rank(args, argv):
    """One rank's supervisor (see the comment above LADDER).  Returns the exit status of this rank."""
    import signal
    import socket
    import subprocess
    import tempfile

    env0 = dict(os.environ)
    rank, world = int(env0.get("RANK", 0)), int(env0["WORLD_SIZE"])
    ppid = os.getppid()
    job = env0.get("VDPP_BENCH_JOB") or f"{ppid}_{_proc_start_time(ppid)}_{env0.get('MASTER_PORT', '0')}"
    rdv = os.path.join(tempfile.gettempdir(), f"vdpp_bench_{job}")
    os.makedirs(rdv, exist_ok=True)
    ladder = ladder_for(args, env0)
    limit = max(5.0, args.watchdog)
    proc = [None]

    def on_signal(signum, _frame):          # torchrun tears a failed job down with SIGTERM: take the child along (exact PID)
        if proc[0] is not None and proc[0].poll() is None:
            proc[0].terminate()
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)

    def put(name, text):
        tmp = os.path.join(rdv, f".{name}.{rank}.tmp")
        with open(tmp, "w") as fh:
            fh.write(text)
        os.replace(tmp, os.path.join(rdv, name))

    def get(name):
        try:
            with open(os.path.join(rdv, name)) as fh:
                return fh.read()
        except OSError:
            return None

    def wait_for(name, seconds):
        end = time.monotonic() + seconds
        while time.monotonic() < end:
            v = get(name)
            if v is not None:
                return v
            time.sleep(0.1)
        return None

    def leave(status):
        """Best-effort tidy-up of the rendezvous files (rank 0 last: the others only ever read files of finished attempts)."""
        try:
            if rank == 0:
                time.sleep(1.0)
                for name in os.listdir(rdv):
                    os.unlink(os.path.join(rdv, name))
                os.rmdir(rdv)
        except OSError:
            pass
        return status

    history, rc = [], 1
    for k, rung in enumerate(ladder):
        env = dict(env0, VDPP_BENCH_WORKER="1", VDPP_BENCH_RUNG=rung["key"], VDPP_BENCH_ATTEMPT=str(k),
                   VDPP_BENCH_STATUS=os.path.join(rdv, f"a{k}.r{rank}.where"), VDPP_BENCH_PREV=json.dumps(history))
        env.update(rung["env"])
        if k > 0:
            # a fresh rendezvous: the agent's store still holds the failed group's keys
            if rank == 0:
                with socket.socket() as sock:
                    sock.bind(("127.0.0.1", 0))
                    put(f"a{k}.port", str(sock.getsockname()[1]))
            port = wait_for(f"a{k}.port", 120)
            if port is None:
                print(f"[rank {rank}] supervisor: no port for attempt {k}; giving up", file=sys.stderr, flush=True)
                return rc or 1
            env["MASTER_PORT"] = port
            env["MASTER_ADDR"] = "127.0.0.1"
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        if rank == 0 and (k > 0 or len(ladder) < len(LADDER)):
            print(f"bench.py: attempt {k}: schedule {rung['schedule']}, transport {rung['transport']}", file=sys.stderr, flush=True)
        cmd = [sys.executable, os.path.abspath(__file__)] + list(argv) + [f for f in rung["flags"] if f not in argv]
        proc[0] = subprocess.Popen(cmd, env=env)
        killed_after = None
        while True:
            rc = proc[0].poll()
            if rc is not None:
                break
            bad = [r for r in range(world) if r != rank and (get(f"a{k}.r{r}.rc") or "0").split()[0] != "0"]
            if bad and killed_after is None:
                killed_after = time.monotonic() + 5.0          # a peer has failed: this attempt is lost, do not sit out the watchdog
            if killed_after is not None and time.monotonic() > killed_after:
                proc[0].terminate()
                try:
                    proc[0].wait(10)
                except subprocess.TimeoutExpired:
                    proc[0].kill()
                rc = proc[0].wait()
                if rc == 0:
                    rc = 5
                put(f"a{k}.r{rank}.where", f"ended by its supervisor after rank {bad[0]} failed")
                break
            time.sleep(0.25)
        put(f"a{k}.r{rank}.rc", str(rc))
        rcs = []
        for r in range(world):
            v = wait_for(f"a{k}.r{r}.rc", limit + 90)
            rcs.append(int(v.split()[0]) if v is not None else -1)
        if all(v == 0 for v in rcs):
            return leave(0)
        where = {str(r): (get(f"a{k}.r{r}.where") or "").strip() for r in range(world) if rcs[r] != 0}
        history.append({"attempt": k, "schedule": rung["schedule"], "transport": rung["transport"], "rc_per_rank": rcs,
                        "watchdog_where": where})
        if rank == 0:
            print(f"bench.py: attempt {k} ({rung['key']}) FAILED: exit status per rank {rcs}, {where}", file=sys.stderr, flush=True)
        if RC_USAGE in rcs:
            return leave(RC_USAGE)
        if rc == 0:
            rc = 6                                            # this rank was fine, the job was not
    return leave(rc)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (ref scripts/benchmark_comparison.sh:85-120 wraps
    every GPU count in torchrun; so does this): start `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a FRESH child process (never an exec; this parent has not touched the GPU and never will), pass the ranks'
    stderr through, relay rank 0's JSON line as the only stdout line, and return the child's exit status."""
    import socket
    import subprocess

    n = args.gpus
    have = visible_gpus()
    shared = os.environ.get("VDPP_SHARE_GPU") == "1"
    if have is not None and have < n and not shared and not args.rehearse_cpu:
        print(f"bench.py: --gpus {n} but {have} GPU(s) are visible to this process (KFD topology / *_VISIBLE_DEVICES); "
              f"refusing to run a smaller job under the name of a larger one.  (Rehearsal on fewer cards: "
              f"VDPP_SHARE_GPU=1 PIPELINE_BACKEND=gloo.)", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["VDPP_BENCH_LAUNCHED_BY_PARENT"] = "1"
    p2p_env(env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"bench.py: WORLD_SIZE is unset; starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line_out = None
    for line in proc.stdout:
        txt = line.strip()
        is_result = False
        if txt.startswith("{"):
            try:
                is_result = "metric" in json.loads(txt)
            except ValueError:
                pass
        if is_result:
            line_out = txt
        else:
            sys.stderr.write(line)      # anything else a rank wrote to stdout (Gloo's peer announcements, ...)
    rc = proc.wait()
    if rc == 0 and line_out is None:
        print("bench.py: the ranks exited cleanly but printed no result line", file=sys.stderr)
        rc = 4
    if line_out is not None and rc == 0:
        print(line_out, flush=True)
    return rc


def main():
    args = parse()
    if args.gpus < 1:
        _usage("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    world = int(os.environ.get("WORLD_SIZE", 1))
    # Supervised ranks need a launcher whose ranks share one parent process on one node (torchrun: TORCHELASTIC_RUN_ID is
    # set by its agent; this script's own launcher goes through torchrun too) or an explicit job name for the rendezvous
    # directory (VDPP_BENCH_JOB); under anything else the rank runs in the launched process itself, as before round 5.
    supervisable = any(os.environ.get(k) for k in ("TORCHELASTIC_RUN_ID", "VDPP_BENCH_JOB"))
    if world == args.gpus and world > 1 and os.environ.get("VDPP_BENCH_WORKER") != "1" and not args.no_fallback and supervisable:
        sys.exit(supervise_rank(args, sys.argv[1:]))        # standard library only up to here: the supervisor never touches the GPU
    if world != args.gpus:
        # never report one job size under the name of another
        _usage(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus} "
                         f"(or plain `python bench.py --gpus {args.gpus}`, which starts its own ranks)")
    rehearse = args.rehearse_cpu
    if rehearse:
        os.environ["PIPELINE_BACKEND"] = "gloo"
    rccl_env = p2p_env(os.environ) if world > 1 else {}
    _imports()
    shared = os.environ.get("VDPP_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < world and not shared and not rehearse:
        _usage(f"bench.py: --gpus {world} but torch sees {have} device(s); refusing to run (rehearsal on fewer "
                         f"cards: VDPP_SHARE_GPU=1 PIPELINE_BACKEND=gloo)")
    n = world
    # N>1: 32N videos so that filling/draining the chain (N-1 stage times inside the bracketed region: the barriers on
    # both sides drain it) stays near 5-10 %; ~20-40 s at every N
    steps = args.steps if args.steps is not None else (16 if n == 1 else 32 * n)
    # How many videos a GPU keeps in flight: `mb` videos travel together as ONE pipeline sample of shape (mb,4,F,H,W)
    # (north_star: "micro-batched pipeline"), `conc` samples are interleaved on separate HIP streams: chosen below, once
    # the schedule is known (choose_in_flight).  `steps` and `value` keep counting VIDEOS.
    if shared:
        # rehearsal only (PIPELINE_BACKEND=gloo on a one-GPU box): ranks share the cards that exist; RCCL refuses this
        local_rank %= max(1, torch.cuda.device_count())
    device = torch.device("cpu") if rehearse else torch.device(f"cuda:{local_rank}")
    cuda = device.type == "cuda"
    if cuda:
        torch.cuda.set_device(device)
    else:
        torch.set_num_threads(max(1, usable_cores() // n))
    new_event = torch.cuda.Event if cuda else _HostEvent

    def sync():
        if cuda:
            torch.cuda.synchronize(device)

    dog = Watchdog(args.watchdog if n > 1 else 0.0, rank)
    if n > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dog.beat("init_process_group")
        init_distributed(backend=resolve_backend(None, simulator=rehearse), rank=rank, world_size=n)
        dog.beat("first barrier")
        dist.barrier()      # create the world communicator collectively, before the first grouped send/recv needs it
        dog.beat("model construction")
    # what the process group itself saw (the JSON line reports THIS, not the --gpus argument)
    me = {"rank": rank, "device": str(device), "name": torch.cuda.get_device_name(device) if cuda else "host cores",
          "uuid": str(getattr(torch.cuda.get_device_properties(device), "uuid", "")) if cuda else f"pid {os.getpid()}"}
    ranks_seen = [me]
    if n > 1:
        ranks_seen = [None] * n
        dist.all_gather_object(ranks_seen, me)
        if dist.get_world_size() != args.gpus:
            _usage(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
        if not shared and len({(r["device"], r["uuid"]) for r in ranks_seen}) != n:
            _usage(f"bench.py: {n} ranks but they do not sit on {n} distinct devices: {ranks_seen}")

    # ---- schedule: ring by default at N > 1 (no fill / drain inside the timed region), after a self-test of its hand-off
    T = args.total_steps
    ring = n > 1 and not args.chain
    selftest = "not requested" if n > 1 else "single GPU"
    if ring:
        dog.beat("ring self-test")
        full_limit, dog.limit = dog.limit, min(dog.limit, 30.0) if dog.limit > 0 else dog.limit
        ring = ring_selftest(rank, n, device)      # a STALL here ends the attempt after 30 s; the supervisors then take the chain
        dog.beat("ring self-test done")
        dog.limit = full_limit
        selftest = "passed" if ring else "FAILED on some rank (chain schedule used)"
    from vdpp_amd.models.unet_spec import UNetConfig, forward_flops
    from vdpp_amd.hip import ops
    from vdpp_amd.models.svd_unet import StableVideoUNet

    lat_dtype = torch.float16
    measured_table = None
    if rehearse:
        from vdpp_amd.models import DummyUNet
        torch.manual_seed(0)
        model = DummyUNet(4, 16)                 # the simulator's model (ref src/models/dummy_unet.py), same weights on every rank
        model.init_noise_sigma = 1.0
        lat_dtype = torch.float32
    else:
        model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(T), seed=0, device=device,
                                                 fp8_attention=args.fp8_attention, long_attention=False if args.no_long_attention else None)
        if n > 1 and (args.micro_batch is None or args.concurrent is None):
            # N > 1: how many videos a GPU keeps in flight trades a busy stage's rate against fill and drain; the four
            # per-forward times that decide it are measured on this job's own GPUs (the static table is from a 1-GPU box)
            measured_table = measure_in_flight_table(model, device, args, n, dog)
    mb, conc, how_chosen, ms_table = choose_in_flight(steps, n, ring, T, args, measured_table)
    if mb < 1 or steps % mb:
        _usage(f"bench.py: --steps {steps} videos is not a whole number of micro-batches of {mb}")
    warmup_requested = args.warmup
    warmup = args.warmup if args.warmup is not None else (2 * mb * conc if n == 1 else max(2 * n, 2 * mb * conc))
    n_samples = steps // mb
    warm_samples = -(-warmup // mb)
    warmup = warm_samples * mb              # whole micro-batches: --warmup 5 with micro-batches of two runs 6
    if not rehearse:
        torch.manual_seed(args.seed)  # same dummy conditioning on every rank
        model.set_dummy_conditioning(mb, args.frames, args.height, args.width, device,
                                     guidance_scale=args.guidance_scale)
    passes = 2 if (args.guidance_scale or 0) > 1.0 else 1
    shape = torch.Size((mb, 4, args.frames, args.height, args.width))
    spec = LatentSpec(shape=shape, dtype=lat_dtype, device=device)
    import logging
    quiet = logging.getLogger("bench.quiet"); quiet.setLevel(logging.ERROR)
    rotating = n > 1 and not ring and not args.no_rotate
    stage = PipelineStage(model, PipelineConfig(total_steps=T, world_size=n, rank=rank, timesteps=list(range(T)),
                                                latent_spec=spec, balanced=True, concurrent_samples=conc,
                                                rotate=rotating, ring=ring),
                          logger=quiet)
    probe = {"ran": False, "why": "single rank" if n == 1 else "ring schedule (its own self-test ran: " + selftest + ")"}
    if n > 1 and not ring:
        dog.beat("p2p probe")
        probe = p2p_probe(rank, n, device)
        if probe.get("serialised") and stage._link is not None:
            stage._link.post_after_send = True
    describe_rank(rank, n, device, ring, rotating, conc, selftest, stage.transport, probe)
    steps_done = [0]
    inner_model = model.forward

    if n > 1:           # heartbeat: every UNet step enqueued on this rank counts as progress
        def beating_forward(latent, step):
            steps_done[0] += 1
            dog.beat(f"UNet step {step} (#{steps_done[0]} on this rank)")
            return inner_model(latent, step)
        model.forward = beating_forward
    gen = torch.Generator(device=device)

    def supplier(i):
        gen.manual_seed(args.seed + i)
        return torch.randn(shape, generator=gen, device=device, dtype=lat_dtype) * model.init_noise_sigma

    def fence():
        sync()
        if n > 1:
            dist.barrier()
        sync()

    # which rung of the ladder this process is (the ring may have fallen back to the chain in-process after its self-test)
    rung_key = "ring" if ring else ("blocking" if os.environ.get("VDPP_ASYNC_COMM") == "0" else "chain")
    # test hook (tests/test_bench_launcher_cpu.py): VDPP_BENCH_FAULT="ring:5,chain:2" parks rank 5 of the ring attempt and
    # rank 2 of the chain attempt in front of their first hand-off, as a stalled RCCL transfer would
    for item in filter(None, os.environ.get("VDPP_BENCH_FAULT", "").split(",")):
        key, _, who = item.partition(":")
        if n > 1 and key == rung_key and who.isdigit() and int(who) == rank:
            print(f"[rank {rank}] INJECTED FAULT: parking in the {rung_key} attempt (VDPP_BENCH_FAULT)", file=sys.stderr, flush=True)
            dog.beat(f"first hand-off of the {rung_key} schedule (injected fault: parked)")
            while True:
                time.sleep(1.0)
    with torch.no_grad():
        if warmup > 0:
            dog.beat("warm-up")
            stage.run_many(warm_samples, input_supplier=supplier if (rank == 0 or ring) else None)
            stage.drain()
        dog.beat("fence before the timed region")
        fence()
        if cuda:
            torch.cuda.reset_peak_memory_stats(device)      # ref src/modes/benchmark.py:240-249: peak of the timed region
        done_events = []
        # shader clock over the timed region (rank 0, real model): a ~2 us stamp of the shader-clock and 100 MHz counters at the
        # start and behind every finished sample, in stream order (sp_clock_stamp) -- `roofline.clock_ghz_live`
        stamps = None
        if cuda and not rehearse and rank == 0:
            try:
                stamps = ops.ClockStamps(device, n_samples + 3)
            except Exception as exc:  # noqa: BLE001
                print(f"bench.py: no clock stamps ({exc!r})", file=sys.stderr, flush=True)

        def on_done(_idx):   # runs on the finishing sample's stream, right after its last step was enqueued
            ev = new_event(enable_timing=True); ev.record(); done_events.append(ev)
            if stamps is not None:
                stamps.stamp()

        stage.sample_done_hook = on_done
        t0 = time.perf_counter()
        start_ev = new_event(enable_timing=True); start_ev.record()
        if stamps is not None:
            stamps.stamp()
        stage.run_many(n_samples, input_supplier=(lambda i: supplier(warm_samples + i)) if (rank == 0 or ring) else None)
        stage.drain()
        dog.beat("fence after the timed region")
        fence()
        elapsed = time.perf_counter() - t0
        clock_live = None
        if stamps is not None:
            try:
                if stamps.n < 2:                 # chain: rank 0 finishes no sample itself -- a closing stamp behind the fence
                    stamps.stamp()
                    torch.cuda.synchronize(device)
                clock_live = stamps.ghz()        # (GHz, seconds between the first and the last stamp, XCDs paired)
            except Exception as exc:  # noqa: BLE001
                print(f"bench.py: clock stamps unreadable ({exc!r})", file=sys.stderr, flush=True)
    frames_out = None
    if not rehearse and not args.no_decode and (args.emit_frames or n == 1):
        dog.beat("frames-out leg")
        frames_out = frames_out_leg(args, stage, supplier, device, n, rank, ring, mb, conc, fence, dog)
    dog.beat("reduction of the timings")
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    my_peak = torch.cuda.max_memory_allocated(device) / 2**30 if cuda else 0.0
    peaks = [my_peak]
    transports = [stage.transport]
    if n > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        peaks = [None] * n
        dist.all_gather_object(peaks, my_peak)
        transports = [None] * n
        dist.all_gather_object(transports, stage.transport)
    elapsed = float(tmax.item())

    # steady-state figure in the reference's definition (benchmark.py:254-267): successive completion
    # times on the last rank, first N-1 samples of the timed region (pipe fill) dropped
    steady = None
    fill = None
    if rank == n - 1:
        times = sorted(start_ev.elapsed_time(e) / 1e3 for e in done_events) or [elapsed]
        fill = times[0]
        # completions arrive in groups of `conc` (videos interleaved on streams finish together): drop whole groups
        # covering the pipeline fill (N-1 samples) and rate the rest group-to-group.  Ring schedule: the last rank
        # finishes one video in N (those that started on rank 0), the others finish elsewhere at the same moments.
        per_event = (n if ring else 1) * mb          # videos behind one completion event
        groups_dropped = 1 if ring else max(1, -(-(n - 1) // conc))
        d = groups_dropped * conc
        if len(times) - d >= 1 and times[-1] > times[d - 1]:
            steady = per_event * (len(times) - d) / (times[-1] - times[d - 1])
        else:
            steady = per_event * len(times) / times[-1]
    info = torch.tensor([steady or 0.0, fill or 0.0], dtype=torch.float64, device=device)
    if n > 1:
        dist.broadcast(info, src=n - 1)
    steady, fill = float(info[0]), float(info[1])

    out = None
    if rank == 0:
        value = steps / elapsed
        flops_all = forward_flops(UNetConfig.svd(), args.frames, args.height, args.width)
        flops_exec = forward_flops(UNetConfig.svd(), args.frames, args.height, args.width,
                                count_cross_attn_qo=False)["total"]
        out = {
            "metric": ("REHEARSAL, NOT A MEASUREMENT (DummyUNet on CPU tensors over Gloo through the benchmark's multi-rank "
                       "path): " if rehearse else "") + "steady-state videos/sec (whole node), SVD 14f x 25step",
            "value": value, "unit": "videos/s", "n_gpus": n, "steps": steps, "warmup": warmup,
            "warmup_requested": warmup_requested,     # --warmup is rounded up to whole micro-batches
            "in_flight_choice": {"micro_batch": mb, "streams": conc, "how": how_chosen, "table_ms_per_video_forward": ms_table},
            "ring_selftest": selftest,
            # what ran before this line was produced (supervise_rank): failed attempts with their exit status per rank and
            # where each watchdog fired, then this one
            "attempts": json.loads(os.environ.get("VDPP_BENCH_PREV", "[]")) + [
                {"attempt": int(os.environ.get("VDPP_BENCH_ATTEMPT", "0")),
                 "schedule": "ring" if ring else ("chain" if n > 1 else "single GPU"),
                 "transport": ("n/a" if n == 1 else LADDER[[r["key"] for r in LADDER].index(rung_key)]["transport"]),
                 "rc": 0, "supervised": os.environ.get("VDPP_BENCH_WORKER") == "1"}],
            "timestep_order": f"ascending step indices 0..{T - 1} (the reference benchmark feeds {T - 1}..0, ref "
                              "src/modes/benchmark.py:178; the value only indexes the sigma table, every step costs the same)",
            "transport_per_rank": transports, "p2p_probe": probe,
            "world_size_seen_by_process_group": dist.get_world_size() if n > 1 else 1,
            "backend": dist.get_backend() if n > 1 else "none", "ranks": ranks_seen, "rccl_env": rccl_env,
            "gpus_shared_between_ranks": bool(shared and n > 1), "micro_batch": mb, "streams_per_gpu": conc,
            "peak_memory_gb_per_rank": [round(m, 3) for m in peaks], "max_peak_memory_gb": round(max(peaks), 3),
            "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f16+fp8 attention" if args.fp8_attention else "f16", "data": "synthetic",
            "config": {"workload": ("DummyUNet(4,16) fp32 on CPU (rehearsal), " if rehearse else
                                    "SVD img2vid UNet (1.52B params, random init), ") +
                                   f"latent (1,4,{args.frames},{args.height},{args.width}) fp16 per video, every UNet call on "
                                   f"({mb},4,{args.frames},{args.height},{args.width}) [= ({mb},{args.frames},8,{args.height},"
                                   f"{args.width}) UNet sample], {T} steps, {passes} UNet pass/step "
                                   f"(guidance_scale={args.guidance_scale}), micro-batches of {mb} video(s) per UNet "
                                   f"call (the reference benchmark's batch_size=1 is --micro-batch 1), {conc} micro-batches "
                                   f"in flight per GPU on separate HIP streams"
                                   + (", spatial attention on fp8-e4m3 MFMA" if args.fp8_attention else "")
                                   + (", level-0 attention through attn_spatial_kernel" if args.no_long_attention else
                                      ", level-0 attention (rows >= 8,192 tokens) through attn_long_kernel"),
                       "stage_steps": stage_sizes(T, n, balanced=True),
                       "stage_steps_rotate_with_video_index": bool(rotating),
                       "schedule": ("ring: video i starts on rank i mod N and visits every rank once" if ring else
                                    "chain: rank 0 feeds, last rank finishes") if n > 1 else "single GPU",
                       "parallelism": f"step-pipeline pp{n}" if n > 1 else "single GPU (no pipeline split)"},
            "steady_state_videos_per_s_last_rank": steady, "first_video_latency_s": fill,
            "first_video_latency_note": f"completion of the first pipeline sample = {mb} video(s) travelling together, with {conc} "
                                        f"sample(s) sharing the GPU: longer than one video alone (--micro-batch 1 --concurrent 1)",
            "unet_forward_tflop_algorithmic": flops_all["total"] / 1e12,
            "unet_forward_tflop_executed": flops_exec / 1e12,
        }
        # time per UNet forward: at N=1 the whole timed region is forwards; at N>1 the node finishes one video
        # per (bottleneck stage) x (its steps), so this is the per-forward time of the most loaded rank
        per_video = elapsed / steps
        bottleneck_steps = T / n if (rotating or ring) else max(stage_sizes(T, n, balanced=True))
        ms_forward = 1e3 * per_video / (bottleneck_steps * passes)
        out["ms_per_unet_forward" if n == 1 else "ms_per_unet_forward_bottleneck_stage"] = ms_forward
        if frames_out is not None:
            frames_out["latents_out_videos_per_s"] = value
            frames_out["frames_out_over_latents_out"] = frames_out["videos_per_s"] / value
            out["frames_out"] = frames_out
        out["step_roofline"] = {"bound": "mfma", "achieved": flops_exec / 1e12 / (ms_forward / 1e3),
                                "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                                "frac": flops_exec / 1e12 / (ms_forward / 1e3) / PEAK_FP16_TFLOPS,
                                "note": "executed FLOPs of one UNet forward / time per forward"
                                        + ("" if n == 1 else " on the bottleneck stage (includes pipeline fill of the timed region)")}
        if clock_live and clock_live[0]:
            # the same fraction against the peak at the shader clock rank 0's chip actually held over the timed region
            out["step_roofline"]["clock_ghz_live"] = clock_live[0]
            out["step_roofline"]["frac_of_live_clock_peak"] = out["step_roofline"]["achieved"] / (PEAK_FP16_TFLOPS * clock_live[0] / 2.4)

    # ---- per-kernel roofline of the dominant kernel, measured live with events on the launch stream
    if rehearse:
        if n > 1:
            dog.beat("final barrier")
            dist.barrier()      # the line is printed only once every rank has come this far: a failed attempt prints nothing
        if rank == 0:
            out["rehearsal"] = "cpu"
            out.pop("step_roofline", None)
            print(json.dumps(out), flush=True)
        if n > 1:
            finalize_distributed()
        dog.stop()
        return
    if rank == 0 and not args.no_roofline:
        with torch.no_grad():
            lat = supplier(0)
            for _ in range(2):     # (profiled forwards run launch by launch, not as a graph replay: the first one
                ops.PROFILE = []   #  re-populates the allocator and runs on a GPU the host keeps waiting; keep the second)
                model(lat, 0)
                torch.cuda.synchronize(device)
            prof, ops.PROFILE = ops.PROFILE, None
        by, shapes, templates = {}, {}, {}
        for kind, fl, e0, e1, nb, tag in prof:
            sec = e0.elapsed_time(e1) / 1e3
            acc = by.setdefault(kind, [0.0, 0.0, 0, 0.0])
            acc[0] += fl; acc[1] += sec; acc[2] += 1; acc[3] += nb
            if kind == "gemm":
                sh = shapes.setdefault(tag[:5], [0.0, 0.0, 0, tag[5]])
                sh[0] += fl; sh[1] += sec; sh[2] += 1
                tp = templates.setdefault(tag[5], [0.0, 0.0, 0])      # kernel instantiation that took the launch
                tp[0] += fl; tp[1] += sec; tp[2] += 1
        gf, gt, gn, gb = by["gemm"]
        # HBM/fabric bytes per launch come from PMC passes (FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE) that
        # cannot run inside this process; they are collected with tools/pmc_forward.sh on a named commit and committed.
        # The line carries that commit and the algorithmic bytes (every operand / output / residual element once,
        # measured from this run's launches) so that the over-fetch ratio can be read off directly.
        traffic, traffic_src, traffic_commit = None, None, None
        for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03n_pmc_traffic.json", "r03k_pmc_traffic.json", "r03i_pmc_traffic.json", "r03f_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", name)))
                if pj.get("micro_batch", 1) != mb:      # bytes per launch scale with the videos per UNet call
                    continue
                traffic = (pj["fabric_read_bytes"] + pj["write_bytes"]) / pj["launches"]
                traffic_src, traffic_commit = pj["source"], pj.get("commit", "f08fa7c (round 1 final build)")
                break
            except Exception:
                continue
        # The clock the chip holds inside these K loops (2.4 GHz nominal is what the 2.5 PFLOP/s peak assumes) comes from
        # in-kernel stamps of the experiments build (tools/clock_in_kernel.py): a static record too.
        clock = None
        for name in ("r05_clock_in_kernel.json", "r04_clock_in_kernel.json"):
            try:
                clock = json.load(open(os.path.join(ROOT, "profiles", name)))
                break
            except Exception:
                continue
        clk_ghz = clock["clock_ghz_in_kernel"] if clock else None
        out["roofline"] = {"bound": "mfma",
                           "kernel": "gemm_pp_kernel / gemm_ps_kernel / gemm_f16_kernel (implicit-GEMM conv/linear)",
                           "achieved": gf / gt / 1e12, "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                           "frac": gf / gt / 1e12 / PEAK_FP16_TFLOPS, "traffic": traffic,
                           "traffic_static": True,      # measured by separate --pmc passes on the commit below, not in this run
                           "clock_ghz_in_kernel": clk_ghz, "clock_static": True,
                           "clock_measured_at_commit": clock.get("commit") if clock else None,
                           "clock_source": clock.get("source") if clock else None,
                           "clock_adjusted_peak": PEAK_FP16_TFLOPS * clk_ghz / 2.4 if clk_ghz else None,
                           "frac_of_clock_adjusted_peak": (gf / gt / 1e12) / (PEAK_FP16_TFLOPS * clk_ghz / 2.4) if clk_ghz else None,
                           # measured in THIS run: the shader clock the chip held over the timed region (every kernel of the
                           # step, not only these), from stamps of s_memtime against the 100 MHz counter (sp_clock_stamp)
                           "clock_ghz_live": clock_live[0] if clock_live else None,
                           "clock_live_window_s": clock_live[1] if clock_live else None,
                           "clock_live_xcds": clock_live[2] if clock_live else None,
                           "frac_of_live_clock_peak": ((gf / gt / 1e12) / (PEAK_FP16_TFLOPS * clock_live[0] / 2.4)
                                                       if clock_live and clock_live[0] else None),
                           "traffic_source": traffic_src, "traffic_measured_at_commit": traffic_commit,
                           "pmc_dominant_template": pmc_dominant_template(),
                           "algorithmic_bytes_per_launch": gb / gn,
                           "traffic_over_algorithmic": (traffic / (gb / gn)) if traffic else None,
                           "launches_per_forward": gn, "avg_launch_us": 1e6 * gt / gn,
                           "flop_per_launch_avg": gf / gn,
                           # the ten shapes with the most time: [rows, columns, channels per tap, gather mode, GEGLU]
                           "top_shapes": [{"shape": list(k), "launches": v[2], "ms": round(1e3 * v[1], 3),
                                           "tflops": round(v[0] / v[1] / 1e12, 1), "kernel": v[3]}
                                          for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:10]],
                           # FLOPs per kernel instantiation of one forward (names as rocprofv3 prints them): with
                           # profiles/*_kernel_stats.csv the fraction of peak per template can be recomputed
                           "per_template": [{"kernel": k, "launches": v[2], "tflop": round(v[0] / 1e12, 4),
                                             "ms": round(1e3 * v[1], 3), "tflops": round(v[0] / v[1] / 1e12, 1)}
                                            for k, v in sorted(templates.items(), key=lambda kv: -kv[1][1])]}
        # spatial attention: the long rows (level 0: 9,216 tokens) run attn_long_kernel (+ its near-empty second-pass
        # launch, inside the timed call), the shorter levels attn_spatial_kernel
        for key, kind, kname in (("roofline_attention", "attn_spatial_long", "attn_long_kernel (+ flagged second pass)"),
                                 ("roofline_attention_short_rows", "attn_spatial", "attn_spatial_kernel")):
            if kind in by:
                af, at, an, _ = by[kind]
                out[key] = {"bound": "mfma", "kernel": kname, "achieved": af / at / 1e12, "peak": PEAK_FP16_TFLOPS,
                            "unit": "TFLOP/s", "frac": af / at / 1e12 / PEAK_FP16_TFLOPS, "launches_per_forward": an,
                            "avg_launch_us": 1e6 * at / an}
        if "roofline_attention" not in out and "roofline_attention_short_rows" in out:
            out["roofline_attention"] = out.pop("roofline_attention_short_rows")
    # ---- the reference benchmark's own order, one video per UNet call and one at a time (ref src/modes/benchmark.py:101
    # batch_size=1), beside the headline so that rounds and configurations stay comparable: a short leg, never `value`
    if rank == 0 and n == 1 and not rehearse and not args.no_batch1_leg and (mb, conc) != (1, 1):
        with torch.no_grad():
            torch.manual_seed(args.seed)
            model.set_dummy_conditioning(1, args.frames, args.height, args.width, device, guidance_scale=args.guidance_scale)
            shape1 = torch.Size((1, 4, args.frames, args.height, args.width))
            stage1 = PipelineStage(model, PipelineConfig(total_steps=T, world_size=1, rank=0, timesteps=list(range(T)),
                                                         latent_spec=LatentSpec(shape=shape1, dtype=torch.float16, device=device),
                                                         balanced=True, concurrent_samples=1), logger=quiet)

            def supplier1(i):
                gen.manual_seed(args.seed + 5000 + i)
                return torch.randn(shape1, generator=gen, device=device, dtype=torch.float16) * model.init_noise_sigma

            stage1.run_many(1, input_supplier=supplier1)
            sync()
            t1 = time.perf_counter()
            stage1.run_many(3, input_supplier=supplier1)
            sync()
            dt1 = (time.perf_counter() - t1) / 3
        out["reference_order_batch1_one_at_a_time"] = {
            "videos_per_s": 1.0 / dt1, "ms_per_unet_forward": 1e3 * dt1 / (T * passes), "videos": 3,
            "note": "micro-batch 1, one video in flight (the reference benchmark's batch_size=1 loop); the headline runs "
                    f"micro-batches of {mb} on {conc} streams"}
    # ---- what the last stage would add per video if it also decoded (ref scripts/generate_video_demo.py:
    # 154-195: decode_latents, decode_chunk_size 14).  Outside the headline metric: the benchmark's videos are latents.
    if rank == 0 and not args.no_decode:
        out["vae_decode"] = vae_decode_leg(device, args.frames, args.height, args.width)
        out["image_encode"] = image_encode_leg(device, args.frames, args.height, args.width)
    if rank == 0 and n == 1 and not args.no_cpu_baseline:
        del model, stage
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline(args.frames, args.height, args.width, T)
        out["cpu_simulator"] = cpu_simulator()
    if n > 1:
        dog.beat("final barrier (rank 0 measures the per-kernel roofline alone before it)")
        dist.barrier()          # rank 0 ran the roofline leg alone; leave together
    if rank == 0:
        # (behind the barrier: an attempt in which some rank never got here prints no line, so that the supervisors' next
        # attempt cannot put a second one on stdout)
        print(json.dumps(out), flush=True)
    if n > 1:
        finalize_distributed()
    dog.stop()


if __name__ == "__main__":
    main()

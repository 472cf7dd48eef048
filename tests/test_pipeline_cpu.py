"""CPU tests of the launcher / executor API and of the simulator path against the vectors minted from the
reference (tests/golden/*.npz): DummyUNet pipeline over Gloo with world_size 1, 2 (and 4) must reproduce the
reference's final latent BIT-EXACTLY (same torch CPU ops, same order)."""

import logging
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from vdpp_amd.distributed import finalize_distributed, init_distributed, resolve_backend
from vdpp_amd.models import DummyUNet
from vdpp_amd.pipeline import LatentSpec, PipelineConfig, PipelineStage, run_pipeline_latents, run_single_latent


def _load(golden_dir, name, c, hid):
    z = np.load(os.path.join(golden_dir, name))
    model = DummyUNet(c, hid)
    model.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")})
    return z, model.eval()


def test_dummy_unet_matches_reference_per_step(golden_dir):
    z, model = _load(golden_dir, "dummy_c4h64.npz", 4, 64)
    lat = torch.from_numpy(z["input"])
    with torch.no_grad():
        for i, s in enumerate(z["timesteps"].tolist()):
            lat = model(lat, s)
            assert lat.numpy().tobytes() == z["per_step"][i].tobytes(), f"step {s} differs from the reference"


def test_dummy_unet_shapes_like_reference_tests():
    # mirrors what /root/reference/tests/test_dummy_unet.py checks (shape preservation, step accepted)
    for c, shape in ((8, (1, 8, 8, 32, 32)), (4, (2, 4, 8, 32, 32)), (4, (4, 4, 8, 16, 16)), (8, (1, 8, 8, 64, 64))):
        m = DummyUNet(channels=c)
        for step in (0, 10, 27):
            assert m(torch.randn(shape), step=step).shape == torch.Size(shape)
    assert sum(p.numel() for p in DummyUNet(8, 16).parameters()) == 6952


def _rot_worker(rank, ws, golden_dir, init_file, out_file):
    torch.set_num_threads(2)
    z, model = _load(golden_dir, "dummy_c4h64.npz", 4, 64)
    init_distributed(backend="gloo", rank=rank, world_size=ws, init_method=f"file://{init_file}")
    x = torch.from_numpy(z["input"])
    ts = [6, 5, 4, 3, 2, 1, 0]                                  # 7 steps over 3 ranks: 3,2,2 rotating
    spec = LatentSpec(shape=x.shape, dtype=torch.float32, device=torch.device("cpu"))
    quiet = logging.getLogger("quiet"); quiet.setLevel(logging.ERROR)
    stage = PipelineStage(model, PipelineConfig(total_steps=7, world_size=ws, rank=rank, timesteps=ts, latent_spec=spec,
                                                balanced=True, rotate=True), logger=quiet)
    with torch.no_grad():
        outs = stage.run_many(4, input_supplier=(lambda i: x * (i + 1)) if rank == 0 else None)
    if rank == ws - 1:
        torch.save(outs, out_file)
    finalize_distributed()


def test_gloo_rotating_split_world3(golden_dir):
    """Rotating balanced split (7 steps on 3 ranks) over Gloo == plain loop for every sample, bit-exact."""
    with tempfile.TemporaryDirectory() as td:
        out_file = os.path.join(td, "out.pt")
        mp.spawn(_rot_worker, args=(3, golden_dir, os.path.join(td, "init"), out_file), nprocs=3, join=True)
        outs = torch.load(out_file)
    z, model = _load(golden_dir, "dummy_c4h64.npz", 4, 64)
    x = torch.from_numpy(z["input"])
    assert len(outs) == 4
    for i, got in enumerate(outs):
        lat = x * (i + 1)
        with torch.no_grad():
            for s in [6, 5, 4, 3, 2, 1, 0]:
                lat = model(lat, s)
        assert torch.equal(got, lat)


def _ring_worker(rank, ws, golden_dir, init_file, out_file, num_samples):
    torch.set_num_threads(2)
    z, model = _load(golden_dir, "dummy_c4h64.npz", 4, 64)
    init_distributed(backend="gloo", rank=rank, world_size=ws, init_method=f"file://{init_file}")
    x = torch.from_numpy(z["input"])
    ts = [6, 5, 4, 3, 2, 1, 0]
    spec = LatentSpec(shape=x.shape, dtype=torch.float32, device=torch.device("cpu"))
    quiet = logging.getLogger("quiet"); quiet.setLevel(logging.ERROR)
    stage = PipelineStage(model, PipelineConfig(total_steps=7, world_size=ws, rank=rank, timesteps=ts, latent_spec=spec,
                                                balanced=True, ring=True), logger=quiet)
    started = []

    def supplier(i):
        started.append(i)
        return x * (i + 1)
    with torch.no_grad():
        outs = stage.run_many(num_samples, input_supplier=supplier)
    assert started == [i for i in range(num_samples) if i % ws == rank]       # samples start on rank i mod N
    if rank == ws - 1:
        torch.save(outs, out_file)
    else:
        assert outs is None
    finalize_distributed()


@pytest.mark.parametrize("ws,num_samples", [(2, 4), (2, 5), (3, 7), (4, 8), (4, 2), (3, 1)])
def test_gloo_ring_schedule_equals_plain_loop(golden_dir, ws, num_samples):
    """Ring schedule (sample i starts on rank i mod N and walks the ring through stages 0..N-1) over Gloo: every
    sample equals the plain 7-step loop bit for bit, in order, on the last rank - for sample counts that are and are
    not multiples of the world size, and for fewer samples than ranks."""
    with tempfile.TemporaryDirectory() as td:
        out_file = os.path.join(td, "out.pt")
        mp.spawn(_ring_worker, args=(ws, golden_dir, os.path.join(td, "init"), out_file, num_samples), nprocs=ws, join=True)
        outs = torch.load(out_file)
    z, model = _load(golden_dir, "dummy_c4h64.npz", 4, 64)
    x = torch.from_numpy(z["input"])
    assert len(outs) == num_samples
    for i, got in enumerate(outs):
        lat = x * (i + 1)
        with torch.no_grad():
            for s in [6, 5, 4, 3, 2, 1, 0]:
                lat = model(lat, s)
        assert torch.equal(got, lat), f"sample {i}"


def _sched_worker(rank, ws, golden_dir, init_file, out_file, num_samples, total_steps, mode, conc):
    torch.set_num_threads(1)
    z, model = _load(golden_dir, "dummy_c4h64.npz", 4, 64)
    init_distributed(backend="gloo", rank=rank, world_size=ws, init_method=f"file://{init_file}")
    x = torch.from_numpy(z["input"])[:1, :, :2, :4, :6].contiguous()      # tiny latent: the test is about index arithmetic
    ts = list(reversed(range(total_steps)))
    spec = LatentSpec(shape=x.shape, dtype=torch.float32, device=torch.device("cpu"))
    quiet = logging.getLogger("quiet"); quiet.setLevel(logging.ERROR)
    cfg = PipelineConfig(total_steps=total_steps, world_size=ws, rank=rank, timesteps=ts, latent_spec=spec, balanced=True,
                         ring=mode == "ring", rotate=mode == "rotate", concurrent_samples=conc)
    stage = PipelineStage(model, cfg, logger=quiet)
    supplier = (lambda i: x * (1.0 + 0.25 * i)) if (mode == "ring" or rank == 0) else None
    with torch.no_grad():
        outs = stage.run_many(num_samples, input_supplier=supplier)
    if rank == ws - 1:
        torch.save(outs, out_file)
    else:
        assert outs is None
    finalize_distributed()


@pytest.mark.parametrize("ws,num_samples,total_steps,mode,conc", [
    (5, 13, 25, "ring", 2),      # 3 batches on 2 lanes: the last group runs one lane only
    (6, 14, 25, "ring", 2),      # the N = 6 rehearsal that printed nothing in round 1 (3 batches, 2 lanes, ragged last batch)
    (6, 54, 25, "ring", 2),      # its real size: 48 + 6 videos
    (8, 32, 25, "ring", 2),      # BASELINE config 4 at its real sizes: 8 ranks, 25 steps [4,3,3,3,3,3,3,3], 32 samples
    (8, 20, 25, "ring", 3),      # 3 batches on 3 lanes, ragged
    (8, 32, 25, "chain", 1),     # config 4 through the reference's chain (fixed balanced split)
    (8, 32, 25, "rotate", 1),    # ... and with the extra step rotating over the stages
    (8, 32, 30, "chain", 1),     # config 5's split: 30 steps on 8 ranks [4,4,4,4,4,4,3,3]
])
def test_gloo_schedules_at_real_sizes_equal_plain_loop(golden_dir, ws, num_samples, total_steps, mode, conc):
    """Chain / rotating chain / ring over Gloo at the world sizes and sample counts of BASELINE configs 3-5
    (ref src/pipeline/pipeline.py:113-157): every sample equals the plain loop over steps T-1..0, bit for bit, in
    order, on the last rank.  Sample counts are chosen so that batches do not fill the interleave lanes evenly."""
    with tempfile.TemporaryDirectory() as td:
        out_file = os.path.join(td, "out.pt")
        mp.spawn(_sched_worker, args=(ws, golden_dir, os.path.join(td, "init"), out_file, num_samples, total_steps, mode,
                                      conc), nprocs=ws, join=True)
        outs = torch.load(out_file)
    z, model = _load(golden_dir, "dummy_c4h64.npz", 4, 64)
    x = torch.from_numpy(z["input"])[:1, :, :2, :4, :6].contiguous()
    assert len(outs) == num_samples
    threads = torch.get_num_threads()
    torch.set_num_threads(1)        # as in the workers: the convolution's summation order depends on the thread count
    try:
        for i, got in enumerate(outs):
            lat = x * (1.0 + 0.25 * i)
            with torch.no_grad():
                for s in reversed(range(total_steps)):
                    lat = model(lat, s)
            assert torch.equal(got, lat), f"sample {i}"
    finally:
        torch.set_num_threads(threads)


def test_ring_schedule_requires_supplier_everywhere():
    spec = LatentSpec(shape=torch.Size((1, 4, 2, 4, 4)), dtype=torch.float32, device=torch.device("cpu"))
    stage = PipelineStage(lambda l, s: l, PipelineConfig(total_steps=4, world_size=2, rank=1, timesteps=[3, 2, 1, 0],
                                                          latent_spec=spec, balanced=True, ring=True))
    with pytest.raises(ValueError, match="every rank needs the input_supplier"):
        stage.run_many(2, input_supplier=None)


def _worker(rank, ws, golden_dir, name, c, hid, init_file, out_file, many):
    torch.set_num_threads(2)
    z, model = _load(golden_dir, name, c, hid)
    init_distributed(backend=resolve_backend(None, simulator=True), rank=rank, world_size=ws,
                     init_method=f"file://{init_file}")
    x = torch.from_numpy(z["input"])
    ts = z["timesteps"].tolist()
    spec = LatentSpec(shape=x.shape, dtype=torch.float32, device=torch.device("cpu"))
    quiet = logging.getLogger("quiet"); quiet.setLevel(logging.ERROR)
    with torch.no_grad():
        if many:
            outs = run_pipeline_latents(model, total_steps=len(ts), timesteps=ts, world_size=ws, rank=rank,
                                        latent_spec=spec, num_samples=3,
                                        input_supplier=(lambda i: x * (1.0 if i != 1 else 0.5)) if rank == 0 else None,
                                        logger=quiet)
            if rank == ws - 1:
                torch.save(outs, out_file)
            else:
                assert outs is None
        else:
            out = run_single_latent(model, total_steps=len(ts), timesteps=ts, world_size=ws, rank=rank,
                                    latent_spec=spec, input_latent=x if rank == 0 else None, logger=quiet)
            if rank == ws - 1:
                torch.save(out, out_file)
            else:
                assert out is None
    finalize_distributed()


@pytest.mark.parametrize("name,c,hid,ws", [("dummy_c8h16.npz", 8, 16, 2), ("dummy_c4h64.npz", 4, 64, 2),
                                           ("dummy_c4h64.npz", 4, 64, 4)])
def test_gloo_pipeline_reproduces_reference_bit_exact(golden_dir, name, c, hid, ws):
    with tempfile.TemporaryDirectory() as td:
        out_file = os.path.join(td, "out.pt")
        mp.spawn(_worker, args=(ws, golden_dir, name, c, hid, os.path.join(td, "init"), out_file, False),
                 nprocs=ws, join=True)
        got = torch.load(out_file).numpy()
    want = np.load(os.path.join(golden_dir, name))["final"]
    assert got.tobytes() == want.tobytes()


def test_gloo_run_many_world2(golden_dir):
    with tempfile.TemporaryDirectory() as td:
        out_file = os.path.join(td, "out.pt")
        mp.spawn(_worker, args=(2, golden_dir, "dummy_c4h64.npz", 4, 64, os.path.join(td, "init"), out_file, True),
                 nprocs=2, join=True)
        outs = torch.load(out_file)
    want = np.load(os.path.join(golden_dir, "dummy_c4h64.npz"))["final"]
    assert len(outs) == 3
    assert outs[0].numpy().tobytes() == want.tobytes() and outs[2].numpy().tobytes() == want.tobytes()
    assert not np.array_equal(outs[1].numpy(), want)


def test_single_rank_matches_golden_and_passes_timestep_value(golden_dir):
    z, model = _load(golden_dir, "dummy_c8h16.npz", 8, 16)
    x = torch.from_numpy(z["input"])
    ts = z["timesteps"].tolist()
    spec = LatentSpec(shape=x.shape, dtype=torch.float32, device=torch.device("cpu"))
    seen = []

    def spy(latent, step):
        seen.append(step)
        return model(latent, step)

    with torch.no_grad():
        out = run_single_latent(spy, total_steps=8, timesteps=ts, world_size=1, rank=0, latent_spec=spec, input_latent=x)
    assert seen == ts == [7, 6, 5, 4, 3, 2, 1, 0]          # the timestep VALUE is passed (ref pipeline.py:95)
    assert out.numpy().tobytes() == z["final"].tobytes()
    assert abs(float(out.norm()) - float(z["per_step_norm"][-1])) < 1e-2


def test_executor_argument_errors():
    spec = LatentSpec(shape=torch.Size((1, 2, 1, 2, 2)), dtype=torch.float32, device=torch.device("cpu"))
    with pytest.raises(ValueError):
        PipelineConfig(total_steps=4, world_size=1, rank=0, timesteps=[3, 2, 1], latent_spec=spec)
    cfg = PipelineConfig(total_steps=4, world_size=1, rank=0, timesteps=[3, 2, 1, 0], latent_spec=spec)
    stage = PipelineStage(lambda l, s: l, cfg)
    with pytest.raises(ValueError):
        stage.run(None)                                     # rank 0 needs a latent
    with pytest.raises(ValueError):
        stage.run_many(0, input_supplier=lambda i: torch.zeros(spec.shape))
    with pytest.raises(ValueError):
        stage.run_many(2)                                   # rank 0 needs a supplier
    with pytest.raises(ValueError):
        PipelineStage(lambda l, s: l, PipelineConfig(total_steps=5, world_size=2, rank=0,
                                                     timesteps=list(range(5)), latent_spec=spec))
    mid = PipelineStage(lambda l, s: l, PipelineConfig(total_steps=4, world_size=2, rank=1,
                                                       timesteps=[3, 2, 1, 0], latent_spec=spec))
    with pytest.raises(ValueError):
        mid.run(torch.zeros(spec.shape))                    # non-zero ranks must pass None
    bal = PipelineStage(lambda l, s: l, PipelineConfig(total_steps=5, world_size=2, rank=1, timesteps=list(range(5)),
                                                       latent_spec=spec, balanced=True))
    assert (bal.step_range.start, bal.step_range.end) == (3, 5)
    assert spec.empty().shape == spec.shape


def test_backend_resolution(monkeypatch):
    monkeypatch.delenv("PIPELINE_BACKEND", raising=False)
    assert resolve_backend(None, simulator=True) == "gloo"
    assert resolve_backend(None, simulator=False) == "nccl"
    assert resolve_backend("GLOO") == "gloo"
    monkeypatch.setenv("PIPELINE_BACKEND", "gloo")
    assert resolve_backend(None, simulator=False) == "gloo"
    assert resolve_backend("nccl", simulator=True) == "nccl"    # argument beats env
    monkeypatch.setenv("PIPELINE_BACKEND", "mpi")
    with pytest.raises(ValueError):
        resolve_backend(None)
    with pytest.raises(ValueError):
        resolve_backend("ucc")


def test_init_is_idempotent_and_finalize_safe():
    finalize_distributed()                                   # nothing initialised: no-op
    with tempfile.TemporaryDirectory() as td:
        init_distributed(backend="gloo", rank=0, world_size=1, init_method=f"file://{td}/i")
        init_distributed(backend="gloo", rank=0, world_size=1, init_method=f"file://{td}/j")  # second call ignored
        assert torch.distributed.is_initialized()
        finalize_distributed()
        assert not torch.distributed.is_initialized()


def test_simulator_cli_is_reproducible_across_world_sizes(tmp_path):
    """python -m vdpp_amd.modes.simulator under torchrun: same final norm for 1 and 2 ranks (the reference CLI
    is not reproducible because it builds the model unseeded; this one seeds every rank)."""
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    entry = tmp_path / "sim.py"
    entry.write_text("import sys; sys.path.insert(0, %r); import vdpp_amd\n"
                     "from vdpp_amd.modes import simulator; simulator.main(sys.argv[1:])\n" % root)
    norms = []
    for ws, port in ((1, 29541), (2, 29542)):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ws),
                            "--master-addr", "127.0.0.1", "--master-port", str(port), str(entry), "--total-steps", "8"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        m = re.search(r"Final latent norm: ([0-9.eE+-]+)", r.stderr + r.stdout)
        assert m, r.stderr[-2000:]
        norms.append(float(m.group(1)))
    assert norms[0] == norms[1]


# ---------------------------------------------------------------------------------------------------------------------
# Hand-off ORDER under RCCL's rules (tests/p2p_emulation.py): one FIFO per rank pair and side, both directions in it, tags
# ignored, a transfer only when a send at one head meets a receive at the other.  Gloo cannot fail these; RCCL hangs.
# ---------------------------------------------------------------------------------------------------------------------
def _threads_as_ranks(world, fn, timeout=180):
    import threading
    results, errors = {}, []

    def main(rank):
        try:
            results[rank] = fn(rank)
        except Exception as exc:       # noqa: BLE001  (reported in the main thread)
            errors.append((rank, repr(exc)))

    ts = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in ts: t.start()
    for t in ts: t.join(timeout=timeout)
    assert all(not t.is_alive() for t in ts), f"a rank is stuck; errors so far: {errors}"
    return results, errors


@pytest.mark.parametrize("world,num_samples,conc,ring", [(2, 5, 1, True), (2, 6, 2, True), (3, 7, 2, True), (4, 9, 1, True),
                                                        (8, 20, 2, True), (2, 4, 1, False), (3, 5, 1, False), (4, 6, 1, False)])
def test_handoff_order_is_safe_under_rccl_pair_fifo_rules(monkeypatch, golden_dir, world, num_samples, conc, ring):
    """Ring and chain schedules with ranks as threads over the pair-FIFO transport: no pair may ever have two sends or two
    receives facing each other, every operation must be met, results == the plain loop bit for bit, and (N > 2) no rank
    pair may carry traffic in both directions at all -- the property that makes an order crossing impossible."""
    import vdpp_amd.pipeline.pipeline as pl
    from tests.p2p_emulation import PairFifoTransport

    net = PairFifoTransport(timeout=60)
    net.install(monkeypatch, pl.dist)
    z, model = _load(golden_dir, "dummy_c4h64.npz", 4, 64)
    x = torch.from_numpy(z["input"])[:, :, :2, :8, :8].contiguous()
    ts = [6, 5, 4, 3, 2, 1, 0, 7, 9]                             # 9 steps: uneven over every world size tried
    spec = LatentSpec(shape=x.shape, dtype=torch.float32, device=torch.device("cpu"))
    quiet = logging.getLogger("quiet"); quiet.setLevel(logging.ERROR)

    def rank_main(rank):
        net.bind(rank)
        stage = PipelineStage(model, PipelineConfig(total_steps=len(ts), world_size=world, rank=rank, timesteps=ts,
                                                    latent_spec=spec, balanced=True, ring=ring, rotate=not ring,
                                                    concurrent_samples=conc), logger=quiet)
        with torch.no_grad():
            return stage.run_many(num_samples, input_supplier=(lambda i: x * (i + 1)) if (ring or rank == 0) else None)

    results, errors = _threads_as_ranks(world, rank_main)
    assert not errors, errors
    assert net.crossed is None and net.idle()
    outs = results[world - 1]
    assert len(outs) == num_samples and all(results[r] is None for r in range(world - 1))
    for i, got in enumerate(outs):
        lat = x * (i + 1)
        with torch.no_grad():
            for s in ts:
                lat = model(lat, s)
        assert torch.equal(got, lat), f"sample {i}"
    both_ways = {p: d for p, d in net.directions_per_pair().items() if len(d) > 1}
    if world > 2:
        assert not both_ways, f"pairs with traffic in both directions: {both_ways}"
    elif ring:
        assert set(both_ways) == {(0, 1)}                       # N = 2: one peer, one communicator -- ordered by the even/odd rule


def test_pair_fifo_rules_catch_the_round_4_frame_forwarding_order():
    """The detector detects: the order ADVICE r04 found in FrameEmitter's removed forwarding -- on the pair (N-2, N-1) the last
    rank issues recv, recv, send(finished) while rank N-2 issues recv(finished), send, send -- is a crossing."""
    from tests.p2p_emulation import P2POrderError, PairFifoTransport

    net = PairFifoTransport(timeout=5)
    t = torch.zeros(4)
    net.bind(1)                       # "last rank": two pre-posted stage receives, then the forward of a finished latent
    net.irecv(t.clone(), 0); net.irecv(t.clone(), 0); net.isend(t, 0)
    net.bind(0)                       # "rank N-2": the receive of that forward first, then its two stage sends
    with pytest.raises(P2POrderError, match="both sides have receives"):
        net.irecv(t.clone(), 1)
    assert net.crossed
    ok = PairFifoTransport(timeout=5)
    ok.bind(1); ok.irecv(t.clone(), 0); w = ok.isend(t + 1, 0)
    ok.bind(0); ok.isend(t + 2, 1); buf = t.clone(); ok.irecv(buf, 1).wait()
    assert w.is_completed() and ok.idle() and float(buf[0]) == 1.0

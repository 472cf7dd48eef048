"""Single-rank smoke tests of the mode entry points on the GPU (counterparts of the reference's
``python -m src.modes.{benchmark,benchmark_data_parallel,production}``): they run end to end in this process, print the
reference's ``BENCHMARK_JSON=`` line (ref src/modes/benchmark.py:269-313, benchmark_data_parallel.py:232-274) and the
numbers in it are consistent."""

import json
import logging
import re

import pytest
import torch

pytestmark = pytest.mark.gpu


def _env(monkeypatch, tmp_path):
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "1"); monkeypatch.setenv("LOCAL_RANK", "0")
    return ["--backend", "gloo", "--init-method", f"file://{tmp_path}/rendezvous", "--log-level", "WARNING"]


def _json_line(capsys):
    out = capsys.readouterr().out
    m = re.search(r"^BENCHMARK_JSON=(\{.*\})$", out, flags=re.M)
    assert m, out[-2000:]
    return json.loads(m.group(1))


@pytest.mark.parametrize("model", ["dummy", "svd"])
def test_benchmark_mode_prints_the_reference_json(monkeypatch, tmp_path, capsys, model):
    from vdpp_amd.modes import benchmark
    argv = _env(monkeypatch, tmp_path) + ["--model", model, "--total-steps", "4", "--num-samples", "3", "--warmup-samples", "1"]
    size = ["2", "8", "8"] if model == "svd" else ["4", "16", "16"]
    argv += ["--latent-frames", size[0], "--latent-height", size[1], "--latent-width", size[2]]
    benchmark.main(argv)
    r = _json_line(capsys)
    assert r["world_size"] == 1 and r["total_steps"] == 4 and r["steps_per_gpu"] == 4 and r["model"] == model
    assert r["num_samples_measured"] == 3 and r["warmup_samples"] == 1 and r["fsdp"] is False
    assert len(r["per_sample_times_ms"]) == 4                      # warm-up + measured completions on the last rank
    measured = r["per_sample_times_ms"][1:]
    assert abs(r["avg_sample_time_s"] - sum(measured) / 3e3) < 1e-3
    assert abs(r["throughput_samples_per_s"] - 3e3 / sum(measured)) <= 0.02 * r["throughput_samples_per_s"]
    assert r["latent_shape"][1] == 4 and r["max_peak_memory_gb"] >= 0.0 and len(r["peak_memory_gb_per_rank"]) == 1
    assert not torch.distributed.is_initialized()


def test_data_parallel_mode_prints_the_reference_json(monkeypatch, tmp_path, capsys):
    from vdpp_amd.modes import benchmark_data_parallel as dp
    argv = _env(monkeypatch, tmp_path) + ["--model", "dummy", "--total-steps", "3", "--num-samples", "4", "--warmup-samples", "2",
                                         "--latent-frames", "4", "--latent-height", "16", "--latent-width", "16"]
    dp.main(argv)
    r = _json_line(capsys)
    assert r["mode"] == "data_parallel" and r["world_size"] == 1 and r["num_samples_measured"] == 4
    assert r["samples_per_rank"] == 4 and len(r["per_sample_times_ms"]) == 4 and r["warmup_samples"] == 2
    assert r["throughput_samples_per_s"] > 0 and r["wall_clock_s"] > 0 and r["steps_per_gpu"] == 3
    assert not torch.distributed.is_initialized()


def test_production_mode_runs_the_svd_pipeline(monkeypatch, tmp_path, caplog):
    from vdpp_amd.modes import production
    argv = _env(monkeypatch, tmp_path)[:-2] + ["--log-level", "INFO", "--random-init", "--total-steps", "3", "--num-samples", "2",
                                               "--latent-frames", "2", "--latent-height", "8", "--latent-width", "8"]
    with caplog.at_level(logging.INFO):
        production.main(argv)
    norms = [float(m.group(1)) for m in re.finditer(r"final latent norm: ([0-9.eE+-]+)", caplog.text)]
    assert len(norms) == 2 and all(n > 0 and n == n and n != float("inf") for n in norms)
    assert not torch.distributed.is_initialized()


def test_rccl_grouped_p2p_on_a_side_stream_single_rank_self_loop():
    """The only RCCL execution a one-GPU box allows (RCCL refuses two ranks on one device): a single-rank NCCL group
    whose rank sends a latent to itself with ONE grouped isend + irecv on a side stream, ordered against the compute
    stream with events and record_stream exactly as the ring hand-off in pipeline._run_many_ring does.  Own process:
    the test session's process groups are Gloo."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_selfloop.py")], env=env, timeout=180,
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "self-loop p2p ok" in res.stdout and "backend nccl" in res.stdout


def test_step_pipeline_of_the_real_model_is_bit_identical_across_world_sizes(tmp_path):
    """BASELINE configs 3/4 on their own workload (the real 1.52 B-parameter UNet, benchmark latent, 25 steps, three
    videos, two in flight per rank), minus the transport hardware: 1, 2 and 4 ranks share the one GPU and hand the
    latent over Gloo (tools/pp_equivalence.py).  The finished latents must be bit-identical at every world size and for
    the rotating chain as well as the ring schedule: a stage boundary only moves the fp16 latent and every kernel is
    deterministic."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = {}
    # (world, schedule, hand-off): "link" = pipeline._SideStreamLink between real processes (pre-posted irecv into fresh
    # buffers, isend behind an event of the compute stream, take / drain) -- over Gloo its events are waited for on the
    # host, the call sequence and the bookkeeping are the RCCL path's.  4 ranks + their launcher + this test process = the
    # six processes that may have this box's one GPU open at a time.
    for i, (world, schedule, link) in enumerate([(1, "rotate", False), (2, "rotate", True), (4, "rotate", True),
                                                 (2, "ring", False)]):
        out = tmp_path / f"w{world}_{schedule}.pt"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(29650 + i),
               os.path.join(root, "tools", "pp_equivalence.py"), "--out", str(out), "--schedule", schedule]
        if link:
            cmd.append("--async-link")
        res = subprocess.run(cmd, env=env, timeout=400, capture_output=True, text=True)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        if link:
            # every downstream rank received its three latents through the link, and all but the first of them had been
            # posted ahead of time (one UNet step before the previous video's send)
            seen = re.findall(r"\[rank (\d+)\] transport (\{.*\})", res.stdout)
            assert len(seen) == world, res.stdout[-3000:]
            for rk, txt in seen:
                tr = eval(txt)
                assert tr["kind"] == "side-stream link" and tr["host_ordered"] is True
                assert tr["recv_taken"] == (3 if int(rk) > 0 else 0) and tr["sent"] == (3 if int(rk) < world - 1 else 0)
                if int(rk) > 0:
                    assert tr["recv_posted_ahead"] >= 2, tr
        outs[(world, schedule)] = torch.load(out)
    base = outs[(1, "rotate")]
    assert len(base) == 3 and all(torch.isfinite(t.float()).all() for t in base)
    assert not torch.equal(base[0], base[1])
    for key, got in outs.items():
        for i, (a, b) in enumerate(zip(base, got)):
            assert torch.equal(a, b), f"world/schedule {key}: video {i} differs from the single-rank result"


def test_frames_out_of_the_pipelined_run_equal_decode_of_the_plain_loop(tmp_path):
    """The last edge stage inside the pipeline (ref scripts/generate_video_demo.py:418 calls decode_latents on the last rank
    after the step loop; models/edge_stages.py::FrameEmitter decodes every sample where it FINISHES, on a stream of its own
    beside the UNet steps: the ring's rank (i mod N) - 1, the chain's last rank; no finished latent is forwarded, so no
    message order can cross on an RCCL pair communicator).  1, 2 and 3 ranks share the GPU (Gloo); the frames every rank
    decoded, put together, must be bit-identical to `decode_latents` of the single-rank run's finished latents, each sample
    must have been decoded exactly once and on the rank the schedule names."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    K = 5

    def run(world, *extra):
        out = tmp_path / f"w{world}_{'_'.join(a.strip('-') for a in extra) or 'spread'}"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(29700 + world + 10 * len(extra)),
               os.path.join(root, "tools", "pp_frames.py"), "--out-dir", str(out), "--samples", str(K), *extra]
        res = subprocess.run(cmd, env=env, timeout=300, capture_output=True, text=True)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        return [torch.load(out / f"rank{r}.pt") for r in range(world)]

    base = run(1)[0]
    assert sorted(base["frames"]) == list(range(K)) and len(base["latents"]) == K
    # what the single-rank pipeline emitted IS decode_latents of its finished latents (same kernels, same order: bit for bit)
    from vdpp_amd.models.vae_hip import TemporalDecoderHIP, VAEDecoderConfig, random_state_dict
    vcfg = VAEDecoderConfig.tiny(64)
    dec = TemporalDecoderHIP(vcfg, random_state_dict(vcfg, seed=19), "cuda:0")
    for i in range(K):
        want = dec.decode_latents(base["latents"][i].to("cuda:0"), 3).cpu()
        assert base["frames"][i].shape == (1, 3, 3, 64, 128) and torch.isfinite(want).all()
        assert torch.equal(base["frames"][i], want), f"sample {i}: emitted frames differ from decode_latents"
    ring_rank = lambda i, n: (i % n - 1) % n              # noqa: E731  (step_assignment.ring_finish_rank)
    for world, extra, where in ((2, (), lambda i, n: n - 1), (3, (), lambda i, n: n - 1), (3, ("--no-spread",), lambda i, n: n - 1),
                                (2, ("--schedule", "ring"), ring_rank), (3, ("--schedule", "ring", "--concurrent", "1"), ring_rank)):
        ranks = run(world, *extra)
        seen = {}
        for r, rec in enumerate(ranks):
            for i, fr in rec["frames"].items():
                assert i not in seen, f"sample {i} decoded twice"
                seen[i] = r
                assert torch.equal(fr, base["frames"][i]), f"world {world} {extra}: frames of sample {i} differ"
            assert rec["stats"]["decoded"] == len(rec["frames"])
        assert sorted(seen) == list(range(K))
        if where is not None:
            assert all(seen[i] == where(i, world) for i in range(K)), (world, extra, seen)
        assert all(rec["stats"]["forwarded"] == 0 and rec["stats"]["received"] == 0 for rec in ranks)

"""`python bench.py --gpus N` without a launcher must start its own N ranks (ref scripts/benchmark_comparison.sh:85-120
wraps every GPU count in torchrun) from a parent that has imported nothing GPU-related, must relay exactly one JSON
line, and must never report a smaller job under the name of a larger one."""

import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VDPP_SHARE_GPU"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT, env=e, capture_output=True, text=True,
                          timeout=120)


def test_parent_spawns_torchrun_without_importing_torch():
    r = _run("""
        import json, subprocess, sys
        sys.argv = ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"]
        import bench
        assert "torch" not in sys.modules, "bench.py imports torch at module level"
        seen = {}
        real_popen = subprocess.Popen

        def fake_popen(cmd, env=None, **kw):
            seen["cmd"], seen["env"] = cmd, env
            child = ("import json; print('gloo: peer noise on stdout'); print(json.dumps({'not': 'the result'}));"
                     "print(json.dumps({'metric': 'm', 'value': 1.5, 'n_gpus': 4}))")
            return real_popen([sys.executable, "-c", child], env=env, **kw)

        subprocess.Popen = fake_popen
        bench.visible_gpus = lambda: 8
        try:
            bench.main()
        except SystemExit as exc:
            rc = exc.code
        assert rc == 0, rc
        assert "torch" not in sys.modules, "the launching parent imported torch"
        cmd, env = seen["cmd"], seen["env"]
        assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd, cmd
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"], cmd
        assert env["NCCL_MAX_P2P_NCHANNELS"] == "2" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert "RANK" not in env and "WORLD_SIZE" not in env
    """)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line on stdout
    assert json.loads(lines[0]) == {"metric": "m", "value": 1.5, "n_gpus": 4}
    assert "gloo: peer noise" in r.stderr                 # everything else is passed on as diagnostics


def test_parent_refuses_more_ranks_than_devices():
    r = _run("""
        import subprocess, sys
        sys.argv = ["bench.py", "--gpus", "8"]
        import bench
        bench.visible_gpus = lambda: 1
        def boom(*a, **k):
            raise AssertionError("must not spawn")
        subprocess.Popen = boom
        try:
            bench.main()
        except SystemExit as exc:
            assert exc.code not in (0, None), exc.code
            sys.exit(0)
        sys.exit(1)
    """)
    assert r.returncode == 0, r.stderr
    assert "--gpus 8 but 1 GPU(s) are visible" in r.stderr and r.stdout.strip() == ""


def test_child_failure_is_passed_on_and_no_line_is_printed():
    r = _run("""
        import subprocess, sys
        sys.argv = ["bench.py", "--gpus", "2"]
        import bench
        bench.visible_gpus = lambda: 2
        real_popen = subprocess.Popen
        subprocess.Popen = lambda cmd, env=None, **kw: real_popen(
            [sys.executable, "-c", "import sys; print('{\\"metric\\": \\"m\\", \\"value\\": 1}'); sys.exit(3)"], env=env, **kw)
        try:
            bench.main()
        except SystemExit as exc:
            sys.exit(0 if exc.code == 3 else 1)
        sys.exit(1)
    """)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == ""


def test_world_size_mismatch_is_an_error_not_a_smaller_run():
    r = _run("""
        import sys
        sys.argv = ["bench.py", "--gpus", "8"]
        import bench
        bench.main()
    """, env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 8 but WORLD_SIZE=1" in r.stderr


def test_visible_gpus_honours_visible_devices_lists(monkeypatch, tmp_path):
    sys.path.insert(0, ROOT)
    import bench

    import glob as globmod
    nodes = []
    for i, simd in enumerate((0, 256, 256, 256)):         # node 0 = the CPU
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count {simd}\n")
        nodes.append(str(d / "properties"))
    monkeypatch.setattr(globmod, "glob", lambda pat: nodes)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus() == 2


import pytest


@pytest.mark.parametrize("schedule", ["ring", "chain"])
def test_eight_rank_job_through_the_launcher_on_cpu_tensors(schedule):
    """`bench.py --gpus 8` end to end -- launcher, torchrun, process group, ring self-test / p2p probe, the ring schedule
    (the default at N > 1) and the rotating chain [4,3,3,3,3,3,3,3] (--chain), watchdog, all_gather_object, steady-state
    arithmetic, ONE JSON line -- with the simulator's DummyUNet on CPU tensors over Gloo (--rehearse-cpu: a box without 8
    GPUs cannot run the real thing, and at most 6 processes may share the one GPU of the test box).  The line must say what
    the process group saw and must label itself a rehearsal.  20 videos with 5 of warm-up: the driver's own flags."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VDPP_SHARE_GPU", "PIPELINE_BACKEND"):
        e.pop(k, None)
    extra = ["--chain"] if schedule == "chain" else []
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5", "--rehearse-cpu",
                        "--frames", "4", "--height", "16", "--width", "16"] + extra, cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["metric"].startswith("REHEARSAL, NOT A MEASUREMENT") and d["rehearsal"] == "cpu"
    assert d["n_gpus"] == 8 and d["world_size_seen_by_process_group"] == 8 and d["backend"] == "gloo"
    assert d["config"]["stage_steps"] == [4, 3, 3, 3, 3, 3, 3, 3]
    assert d["steps"] == 20 and d["warmup"] == 5 and d["warmup_requested"] == 5
    assert len(d["ranks"]) == 8 and sorted(x["rank"] for x in d["ranks"]) == list(range(8))
    assert len(d["transport_per_rank"]) == 8
    if schedule == "ring":
        assert d["config"]["schedule"].startswith("ring") and d["ring_selftest"] == "passed"
        assert not d["config"]["stage_steps_rotate_with_video_index"] and not d["p2p_probe"]["ran"]
        assert (d["micro_batch"], d["streams_per_gpu"]) == (1, 2)      # 3 batches of 8: two interleaved, then one
        assert all(t["kind"].startswith("ring") for t in d["transport_per_rank"])
    else:
        assert d["config"]["schedule"].startswith("chain") and d["config"]["stage_steps_rotate_with_video_index"]
        assert d["p2p_probe"]["ran"] and d["p2p_probe"]["data_ok"] and not d["p2p_probe"]["serialised"]
        assert len(d["p2p_probe"]["arrived_s"]) == 8 and d["p2p_probe"]["arrived_s"][1] >= 0.3   # rank 1 waited for the late sender
        assert (d["micro_batch"], d["streams_per_gpu"]) == (1, 1)      # a short job for 8 ranks: shortest fill and drain
    assert d["value"] > 0 and d["steady_state_videos_per_s_last_rank"] > 0 and d["first_video_latency_s"] > 0
    assert abs(d["ms_per_step"] * d["value"] - 1e3) < 1e-6 * 1e3
    for k in range(8):
        assert f"[rank {k}/8]" in r.stderr


def _rehearse(extra_args, env_extra, n=8, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VDPP_SHARE_GPU", "PIPELINE_BACKEND", "VDPP_ASYNC_COMM", "VDPP_BENCH_WORKER"):
        e.pop(k, None)
    e.update(env_extra)
    return subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "16", "--warmup", "4", "--rehearse-cpu",
                           "--frames", "4", "--height", "16", "--width", "16", "--watchdog", "6"] + extra_args,
                          cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("fault,ends_on", [("ring:5", "chain"), ("ring:0,chain:3", "blocking")])
def test_a_stalled_attempt_falls_back_and_the_line_records_it(fault, ends_on):
    """The first RCCL run must not end without a number.  With N > 1 every rank is a torch-free supervisor that runs the
    benchmark rank as a fresh child process; VDPP_BENCH_FAULT parks one rank in front of its first hand-off the way a
    stalled RCCL transfer would (its watchdog exits it with status 3, the peers' supervisors end their ranks), and all
    supervisors then start the next rung together on a fresh process group: ring -> chain on the side-stream link -> chain
    on the reference's blocking send/recv (ref src/pipeline/pipeline.py:134-157).  The 8-rank job must come back with ONE
    JSON line whose `attempts` name the failed rungs, their exit status per rank and where the watchdog fired."""
    r = _rehearse([], {"VDPP_BENCH_FAULT": fault})
    assert r.returncode == 0, r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    att = d["attempts"]
    assert [a["schedule"] for a in att] == (["ring", "chain"] if ends_on == "chain" else ["ring", "chain", "chain"])
    assert att[-1]["rc"] == 0 and att[-1]["supervised"] is True and att[-1]["attempt"] == len(att) - 1
    parked = int(fault.split(",")[0].split(":")[1])
    first = att[0]
    assert first["rc_per_rank"][parked] == 3 and all(v != 0 for v in first["rc_per_rank"])
    assert "injected fault" in first["watchdog_where"][str(parked)]
    assert d["config"]["schedule"].startswith("chain") and d["n_gpus"] == 8 and d["value"] > 0
    if ends_on == "blocking":
        assert att[1]["rc_per_rank"][3] == 3 and "blocking" in att[2]["transport"]
    assert "INJECTED FAULT" in r.stderr and "attempt 1" in r.stderr


def test_a_job_that_cannot_succeed_on_any_rung_returns_nonzero_without_a_line():
    r = _rehearse([], {"VDPP_BENCH_FAULT": "ring:1,chain:1,blocking:1"}, n=2)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert r.stderr.count("FAILED") >= 3


def test_refused_arguments_are_not_retried():
    r = _rehearse(["--micro-batch", "3"], {}, n=2)             # 16 videos are not a whole number of micro-batches of 3
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "not a whole number of micro-batches" in r.stderr
    assert "attempt 1" not in r.stderr

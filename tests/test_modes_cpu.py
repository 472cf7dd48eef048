"""CPU tests of the benchmark harnesses' arithmetic (the GPU entry points are smoke-tested under -m gpu)."""

import argparse
import os
import tempfile
import time

import torch
import torch.multiprocessing as mp

from vdpp_amd.distributed import finalize_distributed, init_distributed
from vdpp_amd.models import DummyUNet
from vdpp_amd.modes import benchmark_data_parallel as dp


def _dp_worker(rank, ws, init_file, out_file, num_samples, warmup):
    torch.set_num_threads(1)
    init_distributed(backend="gloo", rank=rank, world_size=ws, init_method=f"file://{init_file}")
    torch.manual_seed(0)
    model = DummyUNet(4, 8)
    calls = []

    def spy(latent, step):
        calls.append(step)
        time.sleep(0.004)               # samples of >= 12 ms: the report rounds times to 0.1 ms
        return model(latent, step)

    args = argparse.Namespace(total_steps=3, num_samples=num_samples, warmup_samples=warmup, seed=7, model="dummy")
    shape = torch.Size((1, 4, 2, 4, 4))
    res = dp.measure(spy, shape=shape, dtype=torch.float32, device=torch.device("cpu"), scale=1.0, rank=rank, world=ws,
                     args=args)
    per_rank = -(-num_samples // ws)
    mine = max(min((rank + 1) * per_rank, num_samples) - rank * per_rank, 0)
    assert len(calls) == (warmup + mine) * 3            # every rank warms up; only its own share is measured
    if rank == 0:
        torch.save(res, out_file)
    else:
        assert res is None
    finalize_distributed()


def test_data_parallel_measurement_follows_the_reference():
    """ref src/modes/benchmark_data_parallel.py:168-247: warm-up on every rank outside the timed region, measured samples
    split in contiguous shares, throughput = all measured samples / slowest rank's measured time."""
    for ws, n, warm in ((2, 5, 1), (3, 2, 2)):          # ragged share; more ranks than samples
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "o.pt")
            mp.spawn(_dp_worker, args=(ws, os.path.join(td, "init"), out, n, warm), nprocs=ws, join=True)
            res = torch.load(out)
        per_rank = -(-n // ws)
        assert res["num_samples_measured"] == n and res["warmup_samples"] == warm
        assert res["samples_per_rank"] == per_rank and res["world_size"] == ws
        assert len(res["per_sample_times_ms"]) == min(per_rank, n)          # rank 0's measured samples only
        assert abs(res["throughput_samples_per_s"] - n / res["wall_clock_s"]) < 2e-2 * res["throughput_samples_per_s"]
        assert res["first_sample_time_s"] > 0 and res["avg_sample_time_s"] > 0
        assert res["mode"] == "data_parallel" and res["steps_per_gpu"] == res["total_steps"] == 3


def test_comparison_csv_schema_is_the_references():
    """ref scripts/benchmark_comparison.sh:47-71: header and one row per BENCHMARK_JSON line."""
    from vdpp_amd.modes import comparison

    assert ",".join(comparison.CSV_HEADER) == \
        "mode,gpu_count,total_steps,steps_per_gpu,num_samples,first_sample_s,avg_sample_s,throughput_sps"
    log = "noise\nBENCHMARK_JSON={\"steps_per_gpu\": 99}\nmore\nBENCHMARK_JSON=" + \
          '{"steps_per_gpu": 4, "first_sample_time_s": 8.7, "avg_sample_time_s": 8.5, "throughput_samples_per_s": 0.1178}\n'
    res = comparison.extract_json(log)                      # the LAST line wins (tail -1 in the reference)
    assert res["steps_per_gpu"] == 4
    assert comparison.csv_row("pipeline_parallel", 7, 28, 14, res) == ["pipeline_parallel", 7, 28, 4, 14, 8.7, 8.5, 0.1178]
    assert comparison.extract_json("nothing here") is None

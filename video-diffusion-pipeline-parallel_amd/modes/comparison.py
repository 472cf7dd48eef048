"""Pipeline-parallel vs data-parallel comparison driver: runs both benchmark modes for a list of GPU counts under
``torch.distributed.run`` and writes the reference's CSV.

Counterpart of ``/root/reference/scripts/benchmark_comparison.sh`` (``:47`` header, ``:50-71`` extraction of the
``BENCHMARK_JSON=`` line, ``:74-139`` loop over GPU counts), in Python and for one MI355X node:

    python -m vdpp_amd.modes.comparison --gpu-counts 1 2 4 8 --model svd --total-steps 25 --balanced

CSV columns (identical to the reference): ``mode,gpu_count,total_steps,steps_per_gpu,num_samples,first_sample_s,
avg_sample_s,throughput_sps`` with ``mode`` in ``pipeline_parallel`` / ``data_parallel``.
"""

from __future__ import annotations

import argparse
import csv
import json
import os
import subprocess
import sys
import time

CSV_HEADER = ["mode", "gpu_count", "total_steps", "steps_per_gpu", "num_samples", "first_sample_s", "avg_sample_s",
              "throughput_sps"]
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def extract_json(log_text: str) -> dict | None:
    """Last ``BENCHMARK_JSON=`` line of a run's output (ref benchmark_comparison.sh:55-60), or None."""
    found = None
    for line in log_text.splitlines():
        if line.startswith("BENCHMARK_JSON="):
            found = line[len("BENCHMARK_JSON="):]
    return json.loads(found) if found else None


def csv_row(mode: str, ngpus: int, total_steps: int, num_samples: int, result: dict) -> list:
    """One CSV row from a mode's JSON (ref benchmark_comparison.sh:62-69)."""
    return [mode, ngpus, total_steps, result["steps_per_gpu"], num_samples, result["first_sample_time_s"],
            result["avg_sample_time_s"], result["throughput_samples_per_s"]]


def _launch(module: str, ngpus: int, port: int, mode_args: list[str], log_path: str) -> str:
    entry = ("import sys; sys.path.insert(0, %r); import vdpp_amd; from vdpp_amd.modes import %s as m; m.main(sys.argv[1:])"
             % (_ROOT, module))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "--no-python", sys.executable, "-c", entry] + mode_args
    run = subprocess.run(cmd, capture_output=True, text=True)
    text = run.stdout + "\n" + run.stderr
    with open(log_path, "w") as fh:
        fh.write(text)
    return text


def main(argv=None) -> None:
    p = argparse.ArgumentParser(description="Pipeline parallel vs data parallel comparison (CSV like the reference)")
    p.add_argument("--gpu-counts", type=int, nargs="+", default=[1, 2, 4, 8])
    p.add_argument("--total-steps", type=int, default=28)
    p.add_argument("--num-samples", type=int, default=14)
    p.add_argument("--warmup-samples", type=int, default=7)
    p.add_argument("--model", type=str, default="svd", choices=["dummy", "svd"])
    p.add_argument("--latent-frames", type=int, default=14)
    p.add_argument("--latent-height", type=int, default=72)
    p.add_argument("--latent-width", type=int, default=128)
    p.add_argument("--hidden-channels", type=int, default=64)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--balanced", action="store_true", help="allow step counts that do not divide by the GPU count")
    p.add_argument("--results-dir", type=str, default="benchmark_results")
    p.add_argument("--base-port", type=int, default=29700)
    args = p.parse_args(argv)

    os.makedirs(args.results_dir, exist_ok=True)
    stamp = time.strftime("%Y%m%d_%H%M%S")
    csv_path = os.path.join(args.results_dir, f"comparison_{stamp}.csv")
    common = ["--total-steps", str(args.total_steps), "--num-samples", str(args.num_samples), "--warmup-samples",
              str(args.warmup_samples), "--model", args.model, "--latent-frames", str(args.latent_frames),
              "--latent-height", str(args.latent_height), "--latent-width", str(args.latent_width),
              "--hidden-channels", str(args.hidden_channels), "--seed", str(args.seed), "--log-level", "WARNING"]
    with open(csv_path, "w", newline="") as fh:
        out = csv.writer(fh)
        out.writerow(CSV_HEADER)
        port = args.base_port
        for ngpus in args.gpu_counts:
            for mode, module, extra in (("pipeline_parallel", "benchmark", ["--balanced"] if args.balanced else []),
                                        ("data_parallel", "benchmark_data_parallel", [])):
                port += 1
                log = os.path.join(args.results_dir, f"{'pp' if mode[0] == 'p' else 'dp'}_{ngpus}gpu_{stamp}.log")
                result = extract_json(_launch(module, ngpus, port, common + extra, log))
                if result is None:
                    print(f"[WARNING] BENCHMARK_JSON not found in {log}", file=sys.stderr)
                    continue
                out.writerow(csv_row(mode, ngpus, args.total_steps, args.num_samples, result))
                fh.flush()
                print(f"{mode} x{ngpus}: {result['throughput_samples_per_s']} samples/s, "
                      f"{result['avg_sample_time_s']} s/sample")
    print(csv_path)


if __name__ == "__main__":
    main()

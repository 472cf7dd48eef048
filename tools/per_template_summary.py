#!/usr/bin/env python3
"""Fraction of the fp16 MFMA peak PER KERNEL TEMPLATE from profiles/ alone.
usage: per_template_summary.py <rocprofv3 --stats kernel_stats.csv> <bench.py JSON line> <forwards profiled>
The bench line's roofline.per_template carries the algorithmic FLOPs each contraction kernel instantiation executes in
ONE UNet forward (names as rocprofv3 demangles them: sp_gemm_last_kernel); the stats CSV carries calls and total time
over `forwards` forwards.  A split-K contraction is two kernels (gemm_pp_kernel<256, 256, 128> + splitk_reduce_kernel):
their times are added."""
import csv, json, re, sys

PEAK = 2500.0
stats, bench, forwards = sys.argv[1], json.load(open(sys.argv[2])), float(sys.argv[3])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"spgemm::", "", n)
    m = re.match(r"(void )?([A-Za-z0-9_:<>, ]+?)\(", n)
    return m.group(2) if m else n


rows = {short(r["Name"]): r for r in csv.DictReader(open(stats))}
print(f"{'kernel instantiation':52s} {'launches/fwd':>12s} {'ms/fwd':>8s} {'TFLOP/fwd':>10s} {'TFLOP/s':>8s} {'of 2.5 PF':>9s}")
tot_ms = tot_fl = 0.0
# templates that run the same main kernel (e.g. "X" and "X + ln_part_finalize_kernel": the same gemm_pp instantiation with and
# without the follow-up that folds two-tile row statistics) share that kernel's CSV row: fold them into one line
merged = {}
for t in bench["roofline"]["per_template"]:
    parts = [p.strip() for p in t["kernel"].split("+")]
    m = merged.setdefault(parts[0], {"parts": [], "tflop": 0.0})
    m["tflop"] += t["tflop"]
    for p in parts:
        if p not in m["parts"]:
            m["parts"].append(p)
for t in merged.values():
    parts = t["parts"]
    t["kernel"] = " + ".join(parts)
    ns = calls = 0.0
    for p in parts:
        r = rows.get(p)
        if r is None:
            cand = [k for k in rows if k.startswith(p.split("<")[0]) and p in k]
            r = rows[cand[0]] if cand else None
        if r is not None:
            ns += float(r["TotalDurationNs"])
            calls = max(calls, float(r["Calls"]))
    ms = ns / 1e6 / forwards
    tf = t["tflop"] / (ms / 1e3) if ms else 0.0
    tot_ms += ms; tot_fl += t["tflop"]
    print(f"{t['kernel']:52s} {calls / forwards:12.1f} {ms:8.3f} {t['tflop']:10.4f} {tf:8.1f} {tf / PEAK:9.3f}")
print(f"{'all contraction kernels':52s} {'':12s} {tot_ms:8.3f} {tot_fl:10.4f} {tot_fl / (tot_ms / 1e3):8.1f} {tot_fl / (tot_ms / 1e3) / PEAK:9.3f}")
for name, key, bkey in (("attn_long_kernel", "attn_long_kernel", "roofline_attention"),
                        ("attn_spatial_kernel", "attn_spatial_kernel", "roofline_attention_short_rows")):
    r = [v for k, v in rows.items() if key in k]
    if bkey not in bench and bkey == "roofline_attention_short_rows" and "attn_spatial" in bench.get("roofline_attention", {}).get("kernel", ""):
        bkey = "roofline_attention"
    if r and bkey in bench and key.split("_kernel")[0] in bench[bkey]["kernel"]:
        ms = sum(float(x["TotalDurationNs"]) for x in r) / 1e6 / forwards
        ra = bench[bkey]
        flops = ra["achieved"] * 1e12 * ra["avg_launch_us"] * 1e-6 * ra["launches_per_forward"]
        print(f"{name:52s} {sum(float(x['Calls']) for x in r) / forwards:12.1f} {ms:8.3f} {flops / 1e12:10.4f} "
              f"{flops / 1e12 / (ms / 1e3):8.1f} {flops / 1e12 / (ms / 1e3) / PEAK:9.3f}")

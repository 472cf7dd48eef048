#!/usr/bin/env python3
"""Fold the three PMC passes of tools/pmc_forward.sh into a per-kernel table (second half of the dispatches =
the profiled step).  HBM bytes: FETCH_SIZE is in KB and reads HALF the bytes of wide coalesced streams on gfx950
(MI355X_MICROARCH.md, HBM section) -> doubled.  The factor was calibrated in round 3 on known byte counts in the GEMM
kernels' own access shapes (profiles/r03_fetch_size_calibration.txt, tools/dma_probe calib): 1.986 for LDS-DMA pieces
of 16 rows x 64 B (the K-step shape), 2.000 for 8 x 128 B, paired halves, contiguous KiB and register loads -- so x2
holds for every kernel family here.  WRITE_SIZE (KB) is exact for 16-byte streaming stores."""
import csv, glob, sys, collections, re
root = sys.argv[1]
def load(sub):
    cc = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(f"{root}/{sub}/**/*kernel_trace.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = list(csv.DictReader(open(cc)))
    return rows, dur
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"spgemm::", "", n)
    m = re.match(r"(void )?([A-Za-z0-9_:<>, ]+?)\(", n)
    return (m.group(2) if m else n)[:48]
tables = {}
for sub in ("fetch", "write", "mfma"):
    rows, dur = load(sub)
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    half = ids[len(ids) // 2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in rows:
        if int(r["Dispatch_Id"]) < half: continue
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen:
            seen.add(r["Dispatch_Id"]); agg[k]["ns"] += dur.get(r["Dispatch_Id"], 0); agg[k]["n"] += 1
    tables[sub] = agg
names = sorted(tables["fetch"], key=lambda k: -tables["mfma"][k]["ns"])
print(f"{'kernel':48s} {'calls':>5s} {'ms':>8s} {'rd GB':>8s} {'wr GB':>8s} {'HBM TB/s':>9s} {'MFMA busy':>9s} {'wait':>6s}")
for k in names:
    f, w, m = tables["fetch"][k], tables["write"][k], tables["mfma"][k]
    ms = m["ns"] / 1e6
    rd = 2 * f["FETCH_SIZE"] * 1024 / 1e9
    wr = w["WRITE_SIZE"] * 1024 / 1e9
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0          # sum over 8 XCDs
    busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc if cyc else 0.0   # per SIMD (1024 SIMDs)
    wait = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else 0.0
    tb = (rd + wr) / (f["ns"] / 1e6 / 1e3) / 1e3 if f["ns"] else 0.0
    if ms < 0.05: continue
    print(f"{k:48s} {int(m['n']):5d} {ms:8.2f} {rd:8.2f} {wr:8.2f} {tb:9.2f} {busy:9.3f} {wait:6.2f}")

# --- fold the implicit-GEMM rows for bench.py's roofline.traffic (argv[2] = output json, argv[3] = commit; optional)
if len(sys.argv) > 2:
    import json
    rd = wr = n = 0.0
    for k in names:
        if "gemm_pp_kernel" in k or "gemm_f16_kernel" in k or "gemm_ps_kernel" in k:
            rd += 2 * tables["fetch"][k]["FETCH_SIZE"] * 1024
            wr += tables["write"][k]["WRITE_SIZE"] * 1024
            n += tables["fetch"][k]["n"]
        elif "splitk_reduce" in k:          # second half of a split-K contraction: its bytes count, not its launch
            rd += 2 * tables["fetch"][k]["FETCH_SIZE"] * 1024
            wr += tables["write"][k]["WRITE_SIZE"] * 1024
    json.dump({"source": "tools/pmc_forward.sh + tools/pmc_forward_summary.py (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                         "passes over the two UNet forwards of tools/one_forward.py = second half of the dispatches)",
               "kernel": "implicit-GEMM kernels (gemm_pp_kernel<*>, gemm_ps_kernel<*>, gemm_f16_kernel<*>)",
               "commit": sys.argv[3] if len(sys.argv) > 3 else "unknown",
               "micro_batch": int(__import__("os").environ.get("BATCH", 2)),      # videos per UNet call in tools/one_forward.py
               "fabric_read_bytes": rd, "write_bytes": wr, "launches": int(n),
               "note": "read bytes = 2 x FETCH_SIZE (gfx950 half-count correction, calibrated on this kernel family's access "
                       "shapes: profiles/r03_fetch_size_calibration.txt), Infinity-Cache hits included"},
              open(sys.argv[2], "w"), indent=1)

#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE against KNOWN byte counts, per access shape.
Input: the rocprofv3 --pmc FETCH_SIZE (--kernel-trace, csv) output directory of `tools/dma_probe calib`.  Every probe
dispatch streams 256 workgroups x 240 steps x 32 KiB = 2,013,265,920 bytes that nothing has touched before (HBM
stream, 8 MiB per workgroup), in one of five access shapes, by LDS-DMA or into registers.
Output: bytes per unit of FETCH_SIZE (the counter is in KiB) = the factor to multiply the counter with."""
import collections, csv, glob, re, sys

root = sys.argv[1]
cc = glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)[0]
BYTES = 256 * 240 * 32768
SHAPES = {0: "16 rows x 64 B (a 32-channel fp16 K-step)", 1: "8 rows x 128 B (whole lines)", 2: "4 rows x 256 B",
          3: "1 KiB contiguous", 4: "16 rows x 64 B, the two halves of a line back to back (paired K-steps)"}
acc = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    m = re.search(r"probe<(\d), (\d)>", r["Kernel_Name"])
    if not m or r["Counter_Name"] != "FETCH_SIZE":
        continue
    acc[(int(m.group(1)), int(m.group(2)))].append(float(r["Counter_Value"]))
print(f"known bytes per dispatch: {BYTES}")
print(f"{'path':10s} {'access shape':72s} {'FETCH_SIZE (KiB)':>18s} {'bytes / (FETCH_SIZE x 1024)':>28s}")
for (shape, path), vals in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    v = sorted(vals)[len(vals) // 2]
    print(f"{'LDS-DMA' if path == 0 else 'registers':10s} {SHAPES[shape]:72s} {v:18.0f} {BYTES / (v * 1024):28.3f}")

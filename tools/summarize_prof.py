#!/usr/bin/env python3
"""Reduce a tools/profile.sh output directory to a short text summary (the file committed under profiles/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(outdir, sub, suffix):
    hits = glob.glob(os.path.join(outdir, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main(outdir):
    print("# profile summary: %s" % os.path.basename(os.path.normpath(outdir)))
    bj = os.path.join(outdir, "bench_trace.json")
    if os.path.exists(bj):
        for line in open(bj):
            line = line.strip()
            if line.startswith("{"):
                d = json.loads(line)
                print("bench (under rocprofv3 --kernel-trace): workload=%s mode=%s value=%.4g %s kernel_ms=%.4f roofline.frac=%.5f" % (
                    d["config"]["workload"], d["config"].get("mode"), d["value"], d["unit"], d["roofline"]["kernel_ms"],
                    d["roofline"]["frac"]))
    st = find(outdir, "trace", "kernel_stats.csv")
    if st:
        print("\n## rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
        for row in csv.DictReader(open(st)):
            print("%-70s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (row["Name"][:70], row["Calls"], row["AverageNs"],
                                                                         row["MinNs"], row["MaxNs"], row["Percentage"]))
    tr = find(outdir, "trace", "kernel_trace.csv")
    if tr:
        rows = [r for r in csv.DictReader(open(tr)) if "klatt" in r["Kernel_Name"]]
        if rows:
            r = rows[-1]
            print("dispatch: grid=%s wg=%s lds=%s scratch=%s vgpr=%s agpr=%s sgpr=%s" % (
                r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"],
                r["Accum_VGPR_Count"], r["SGPR_Count"]))
    for sub in ("pmc_insts", "pmc_waits", "pmc_fetch", "pmc_write"):
        f = find(outdir, sub, "counter_collection.csv")
        if not f:
            print("\n## %s: no counter file" % sub)
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "klatt" not in r["Kernel_Name"]:
                continue
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("\n## %s (per launch, mean over launches)" % sub)
        for k, cs in acc.items():
            print(k[:90])
            for c, v in sorted(cs.items()):
                print("    %-24s %.6g   (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main(sys.argv[1])

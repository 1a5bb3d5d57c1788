"""Reduce a tools/profile.sh output directory to a short text summary (the file committed under profiles/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def synthesis_kernel(name):
    """The kernels of a synthesis LAUNCH (what bench.py times): not the ones speechPlayer_batch_setUtterances runs once per batch."""
    once_per_batch = ("klatt_source_refs", "klatt_frame_facts", "klatt_verify_shared", "klatt_expand_frames")
    return "klatt" in name and not any(k in name for k in once_per_batch)


def find(outdir, sub, suffix):
    hits = glob.glob(os.path.join(outdir, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def pmc_entry(outdir):
    """Per-launch sums over the klatt kernels of one bench.py launch: VALU instructions, HBM bytes (FETCH_SIZE doubled per the
    gfx950 note of MI355X_MICROARCH.md; both counters are in KB), and the kernel-trace average durations."""
    def per_kernel(sub, counter):
        f = find(outdir, sub, "counter_collection.csv")
        acc = defaultdict(list)
        if f:
            for r in csv.DictReader(open(f)):
                if synthesis_kernel(r["Kernel_Name"]) and r["Counter_Name"] == counter:
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in acc.items()}
    valu, fetch, write = per_kernel("pmc_insts", "SQ_INSTS_VALU"), per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
    # the f64 share of the VALU instructions (their own pass: tools/profile.sh pmc_f64) and the clock the chip held (GRBM_GUI_ACTIVE is
    # summed over the 8 XCDs; divided by the kernel-trace duration of the same kernel)
    f64 = {}
    for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"):
        for k, v in per_kernel("pmc_f64", c).items():
            f64[k] = f64.get(k, 0.0) + v
    gui = per_kernel("pmc_insts", "GRBM_GUI_ACTIVE")
    ent = {"kernels": sorted(valu), "valu_insts_per_launch": sum(valu.values()), "valu_insts_by_kernel": valu,
           "valu_f64_insts_per_launch": sum(f64.values()) if f64 else None, "valu_f64_insts_by_kernel": f64 or None,
           "grbm_gui_active_by_kernel": gui or None,
           "fetch_size_kb": sum(fetch.values()), "write_size_kb": sum(write.values()),
           "hbm_bytes_per_launch": int((2.0 * sum(fetch.values()) + sum(write.values())) * 1024)}
    st = find(outdir, "trace", "kernel_stats.csv")
    if st:
        ent["kernel_avg_ns"] = {row["Name"]: float(row["AverageNs"]) for row in csv.DictReader(open(st)) if synthesis_kernel(row["Name"])}
        if gui:
            k = max(ent["kernel_avg_ns"], key=ent["kernel_avg_ns"].get)      # the dominant kernel
            if k in gui:
                ent["held_clock_hz"] = gui[k] / 8.0 / (ent["kernel_avg_ns"][k] * 1e-9)
    bj = os.path.join(outdir, "bench_trace.json")
    if os.path.exists(bj):
        for line in open(bj):
            if line.startswith("{"):
                d = json.loads(line)
                ent["algorithmic_bytes_per_launch"] = d["roofline"]["algorithmic_bytes_per_launch"]
                ent["bench_kernel_ms_under_profiler"] = d["roofline"]["kernel_ms"]
                ent["workload"] = d["config"]["workload"]
    return ent


def merge_json(outdir, key, path):
    """profiles/r2_pmc.json: one entry per workload, stamped with the digest of the engine sources it was measured on."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    sha = bench.engine_source_digest()
    data = {}
    if os.path.exists(path):
        data = json.load(open(path))
        if data.get("engine_sources_sha") != sha:
            data = {}                      # entries measured on other kernels do not mix with this one
    data["engine_sources_sha"] = sha
    data["_source"] = "tools/profile.sh: rocprofv3 --pmc passes (SQ_INSTS_VALU + GRBM_GUI_ACTIVE; SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64; FETCH_SIZE; WRITE_SIZE, each in its own run) + --kernel-trace --stats"
    data["_correction"] = "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); counters are in KB"
    data[key] = pmc_entry(outdir)
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


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
            dur = defaultdict(list)
            for r in rows:
                dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for k, v in sorted(dur.items()):
                v.sort()
                print("%-70s calls=%d median_ns=%d min_ns=%d max_ns=%d" % (k[:70], len(v), v[len(v) // 2], v[0], v[-1]))
            r = rows[-1]
            print("dispatch: grid=%s wg=%s lds=%s scratch=%s vgpr=%s agpr=%s sgpr=%s" % (
                r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"],
                r["Accum_VGPR_Count"], r["SGPR_Count"]))
    for sub in ("pmc_insts", "pmc_f64", "pmc_waits", "pmc_fetch", "pmc_write"):
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
    if len(sys.argv) >= 4:                 # summarize_prof.py <outdir> <workload key> <pmc json to merge into>
        merge_json(sys.argv[1], sys.argv[2], sys.argv[3])

#!/bin/bash
# L1 (TCP) counters of the flat synthesis kernel for two batches: cfg2 as benchmarked (the 64 lanes of a wavefront read the same
# track entries) and with jittered durations (64 different tracks): is the all-different case bound by the L1's gather throughput?
# usage (GPU box): bash tools/pmc_gather.sh > gpurun_out/r2_pmc_gather.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for name in cfg2 jittered; do
  # (a TA_* set -- TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES -- left rocprofv3 with
  # "incomplete dispatches" until the timeout, twice: not collected)
  for set in "TCP_PERF_SEL_TOTAL_READ TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES GRBM_GUI_ACTIVE"; do
    out=/tmp/pmc_gather_$name
    rm -rf $out
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out -- python3 "$ROOT/tools/track_probe.py" only=$name > /dev/null 2> /tmp/pmc_gather.err || { echo "$name: rocprofv3 failed"; tail -3 /tmp/pmc_gather.err; continue; }
    python3 - "$out" "$name" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "klatt_systolic" in k and "Lb1EEE" in k.replace(" ", "") or "true, false, true>" in k:
            rows["flat"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern, c in rows.items():
    print(sys.argv[2], kern, "  ".join("%s %.4g" % (n, sum(v) / len(v)) for n, v in sorted(c.items())), "(mean per launch, %d launches)" % len(next(iter(c.values()))))
PY
  done
done

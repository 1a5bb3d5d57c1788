#!/bin/bash
# HBM traffic of the flat synthesis kernel on a batch of tools/track_probe.py (FETCH_SIZE and WRITE_SIZE in passes of their own, as the
# MI355X guide prescribes; gfx950: FETCH_SIZE counts 32-byte units per 64 B fetched -> x2, both in KB).
# usage (GPU box): bash tools/pmc_hbm.sh jittered [cfg2 ...] > gpurun_out/...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ; do
    out=/tmp/pmc_hbm_${name}_$ctr
    rm -rf $out
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out -- python3 "$ROOT/tools/track_probe.py" only=$name > /dev/null 2> /tmp/pmc_hbm.err || { echo "$name $ctr: rocprofv3 failed"; tail -3 /tmp/pmc_hbm.err; continue; }
    python3 - "$out" "$name" "$ctr" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "klatt_systolic" in k or "klatt_tracks" in k:
            short = "tracks" if "klatt_tracks" in k else ("flat" if k.replace(" ", "").endswith("true,false,true>(klatt::KernelArgs)") or "Lb1ELb0ELb1EEE" in k else "other")
            rows[short].append(float(r["Counter_Value"]))
for kern, v in sorted(rows.items()):
    print(sys.argv[2], sys.argv[3], kern, "mean per launch %.4g over %d launches (max %.4g)" % (sum(v) / len(v), len(v), max(v)))
PY
  done
done

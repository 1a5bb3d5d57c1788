#!/bin/bash
# SQ counters of the direct kernel (klatt_direct) on a batch of tools/direct_ab.py (default all_different), both arithmetic modes.
# usage (GPU box): bash tools/direct_pmc.sh [workload] > gpurun_out/...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-all_different}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  out=/tmp/direct_pmc_out
  rm -rf $out
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $out -- python3 "$ROOT/tools/direct_ab.py" one $W > /dev/null 2> /tmp/direct_pmc.err || { echo "rocprofv3 failed for: $set"; tail -3 /tmp/direct_pmc.err; continue; }
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "klatt_direct" in k:
            rows["direct mode %s" % ("1" if ("<1," in k or "ILi1E" in k) else "0")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern, c in sorted(rows.items()):
    print(kern, "  ".join("%s %.4g" % (n, sum(v) / len(v)) for n, v in sorted(c.items())), "(mean per launch, %d launches)" % len(next(iter(c.values()))))
PY
done

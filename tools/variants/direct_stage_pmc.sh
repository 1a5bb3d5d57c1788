#!/bin/bash
# VALU / SALU instructions of each stage of the direct kernel on its own: variants built with one stage's body compiled in
# (tools/ab_probe.py build ds1=-DKLATT_DIRECT_STAGES=1 ... ds32=...; the other waves only take part in the barriers; PCM is garbage).
# usage (GPU box): bash tools/direct_stage_pmc.sh [workload]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-all_different}
cd /tmp && export TMPDIR=/tmp
for v in base ds1 ds64 ds2 ds4 ds8 ds16 ds32; do
  if [ $v != base ]; then export SPEECHPLAYER_LIB=$ROOT/nvspeechplayer_amd/lib/variants/libspeechPlayer_$v.so; else unset SPEECHPLAYER_LIB; fi
  out=/tmp/direct_stage_pmc_out
  rm -rf $out
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 "$ROOT/tools/direct_ab.py" one $W > /dev/null 2> /tmp/direct_stage_pmc.err || { echo "$v: rocprofv3 failed"; tail -3 /tmp/direct_stage_pmc.err; continue; }
  python3 - "$out" "$v" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "klatt_direct" in k:
            rows["mode %s" % ("1" if "<1," in k else "0")][r["Counter_Name"]].append(float(r["Counter_Value"]))
ROWS = 1515536384 / 64.0     # 64-sample rows of the batch (approximately: the jitter changes the lengths by a per cent)
for kern, c in sorted(rows.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print("%-5s %s: VALU %.4g (%.1f per row)  SALU %.4g (%.1f)  LDS %.4g (%.1f)  GRBM %.4g" % (sys.argv[2], kern, m["SQ_INSTS_VALU"], m["SQ_INSTS_VALU"] / ROWS, m["SQ_INSTS_SALU"], m["SQ_INSTS_SALU"] / ROWS, m["SQ_INSTS_LDS"], m["SQ_INSTS_LDS"] / ROWS, m["GRBM_GUI_ACTIVE"]))
PY
done

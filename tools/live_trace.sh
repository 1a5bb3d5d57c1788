#!/bin/bash
# rocprofv3 --kernel-trace --stats of the live-handle kernels: one handle pulled 8192 samples at a time (tools/single_stream_probe.py), and
# tools/live_unaligned.py at 256 handles (a wavefront per handle) and 8192 handles (64 per wavefront).  Run on the GPU box from the repo root;
# the summary lands in gpurun_out/r6_live_trace.txt.
R=$PWD
mkdir -p gpurun_out
summ() { python3 - "$1" "$2" <<'PY'
import csv, glob, sys
print("# " + sys.argv[2])
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "klatt" in r["Name"] or "ring" in r["Name"].lower() or "scatter" in r["Name"].lower():
            print("%-78s calls=%-5s avg_us=%9.1f min_us=%9.1f max_us=%9.1f" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
{
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lt1 /tmp/lt2 /tmp/lt3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt1 -- python3 $R/tools/single_stream_probe.py 50 > /dev/null 2> /tmp/lt1.err; summ /tmp/lt1 "tools/single_stream_probe.py 50: one live handle, 8192-sample pulls (the LONE instantiation: <MODE, true, 16, 1, true, true, false, true>; MODE 1: the MODE_FAST handle of the same extra)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt2 -- python3 $R/tools/live_unaligned.py 256 > /dev/null 2> /tmp/lt2.err; summ /tmp/lt2 "tools/live_unaligned.py 256: both policies, three cases each (LONE instantiation: a wavefront per handle; <.., false>: 64 handles per wavefront; the skewing pulls of single handles are LONE launches too)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt3 -- python3 $R/tools/live_unaligned.py 8192 > /dev/null 2> /tmp/lt3.err; summ /tmp/lt3 "tools/live_unaligned.py 8192: 64 handles per wavefront, three cases (in step / eight sentences / skewed) of 5 pulls each (+ 8192 skewing pulls of single handles: LONE)"
} > $R/gpurun_out/r6_live_trace.txt 2>&1
cat $R/gpurun_out/r6_live_trace.txt

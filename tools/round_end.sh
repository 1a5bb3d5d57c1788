#!/bin/bash
# What produced the round's numbers, in three GPU calls (`gpurun -- 'bash tools/round_end.sh a'`, then `... b`, `... c`):
#   a: the GPU test suite, the rocprofv3 profiles of the single-GPU configurations in both arithmetic modes (which also write the
#      PMC file bench.py reads)
#   b: the bench lines
#   c: the probes (tracks, mixed, steady, direct stages with a kernel trace of the all-different batch; compact forms, sparse wavefronts),
#      the live-handle bench
#   d: the N > 1 path on the one GPU this box has: bench.py --gpus 2 with the node-sized extras (two ranks share the device, gloo)
# Everything lands under gpurun_out/; the summaries are then copied into profiles/ (see profiles/README.md).
R=r6
mkdir -p gpurun_out
# (A/B variant libraries and their objects are scratch: they would travel with every snapshot -- 131 MB at the end of round 4)
rm -f nvspeechplayer_amd/build_tmp/libspeechPlayer_*.o nvspeechplayer_amd/lib/variants/*.so
case "${1:-a}" in
a)
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/${R}_gputests.log 2>&1; tail -3 gpurun_out/${R}_gputests.log
rm -f gpurun_out/${R}_pmc.json
PMC_KEY=cfg2 PMC_JSON=$PWD/gpurun_out/${R}_pmc.json timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_${R}_cfg2 --workload cfg2 > gpurun_out/prof_${R}_cfg2.log 2>&1
timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_${R}_cfg2_fast --workload cfg2 --mode 1 > gpurun_out/prof_${R}_cfg2_fast.log 2>&1
PMC_KEY=cfg1 PMC_JSON=$PWD/gpurun_out/${R}_pmc.json timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_${R}_cfg1 --workload cfg1 > gpurun_out/prof_${R}_cfg1.log 2>&1
tail -30 gpurun_out/prof_${R}_cfg2/summary.txt; tail -12 gpurun_out/prof_${R}_cfg2_fast/summary.txt; tail -12 gpurun_out/prof_${R}_cfg1/summary.txt
;;
b)
[ -f gpurun_out/${R}_pmc.json ] && cp gpurun_out/${R}_pmc.json profiles/${R}_pmc.json
timeout -k 10 500 python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err; echo bench rc=$?
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_steps20.json 2> gpurun_out/${R}_bench_steps20.err; echo bench rc=$?
timeout -k 10 400 python bench.py --mode 1 --steps 30 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_fast.json 2> /dev/null
timeout -k 10 400 python bench.py --workload cfg1 --steps 50 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_cfg1.json 2> /dev/null
timeout -k 10 400 python bench.py --workload cfg3 --steps 30 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_cfg3.json 2> /dev/null
timeout -k 10 600 python bench.py --workload cfg4 --steps 5 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_cfg4.json 2> /dev/null
python - <<'PY'
import json
for f in ("r6_bench", "r6_bench_steps20", "r6_bench_fast", "r6_bench_cfg1", "r6_bench_cfg3", "r6_bench_cfg4"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line:", e); continue
    print(f, "%.4g samples/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.4f" % d["roofline"]["frac"], d["roofline"].get("valu", {}).get("frac"))
    for k, v in d.items():
        if isinstance(v, dict) and "kernel_ms" in v:
            print("    %-34s %8.3f ms" % (k, v["kernel_ms"]))
        elif k in ("pipeline", "pipeline_from_ipa", "single_stream", "live_handles", "cfg0_cpu") and isinstance(v, dict):
            print("    %-34s %s" % (k, {a: b for a, b in v.items() if not isinstance(b, list) and (k == "live_handles" or not isinstance(b, dict))}))
    if "config" in d and "host" in d["config"]:
        print("    host:", d["config"]["host"])
PY
;;
d)
timeout -k 10 900 python bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/${R}_bench_2ranks_one_gpu.json 2> gpurun_out/${R}_bench_2ranks_one_gpu.err; echo rc=$?
tail -c 3000 gpurun_out/${R}_bench_2ranks_one_gpu.json
;;
c)
timeout -k 10 300 python tools/steady_probe.py > gpurun_out/${R}_steady_probe.txt 2>&1
timeout -k 10 300 python tools/mixed_probe.py 65536 > gpurun_out/${R}_mixed_probe.txt 2>&1
timeout -k 10 600 python tools/track_probe.py 65536 +distinct +cfg3 +cfg4 +unsorted > gpurun_out/${R}_track_probe.txt 2>&1
timeout -k 10 900 python tools/direct_probe.py time 65536 +legacy > gpurun_out/${R}_direct_probe.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ad_trace && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ad_trace -- python3 $OLDPWD/tools/direct_ab.py one all_different > /dev/null 2> /tmp/ad_trace.err; \
  python3 - <<'PY'
import csv, glob
print("# rocprofv3 --kernel-trace --stats of tools/direct_ab.py one all_different (65536 utterances, direct stages, both modes)")
for f in glob.glob("/tmp/ad_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "klatt" in r["Name"]:
            print("%-60s calls=%s avg_us=%.1f min_us=%.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
) > gpurun_out/${R}_all_different_trace.txt 2>&1
timeout -k 10 400 bash tools/direct_pmc.sh all_different > gpurun_out/${R}_direct_pmc.txt 2>&1
R=${R} bash tools/live_round3.sh > /dev/null 2>&1
timeout -k 10 300 python tools/live_large.py > gpurun_out/${R}_live_large.txt 2>&1
timeout -k 10 300 python tools/compact_probe.py 65536 0 > gpurun_out/${R}_compact_probe.txt 2>&1
timeout -k 10 300 python tools/lone_probe2.py > gpurun_out/${R}_sparse_wavefronts.txt 2>&1
timeout -k 10 300 python tools/lone_probe.py > gpurun_out/${R}_one_utterance.txt 2>&1
cat gpurun_out/${R}_compact_probe.txt gpurun_out/${R}_sparse_wavefronts.txt gpurun_out/${R}_one_utterance.txt
cat gpurun_out/${R}_steady_probe.txt gpurun_out/${R}_mixed_probe.txt gpurun_out/${R}_track_probe.txt gpurun_out/${R}_direct_probe.txt gpurun_out/${R}_all_different_trace.txt
grep -v "live. 1 handles" gpurun_out/${R}_live_bench.txt; cat gpurun_out/${R}_live_large.txt
;;
esac

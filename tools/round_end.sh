#!/bin/bash
# What produced the round's numbers, in two GPU calls (`gpurun -- 'bash tools/round_end.sh a'`, then `... b`):
#   a: the GPU test suite, the rocprofv3 profiles of the two single-GPU configurations (which also write the PMC file bench.py reads)
#   b: the bench lines, the probes, the live-handle bench
# Everything lands under gpurun_out/; the summaries are then copied into profiles/ (see profiles/README.md).
R=r3
mkdir -p gpurun_out
if [ "${1:-a}" = a ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${R}_gputests.log 2>&1; tail -3 gpurun_out/${R}_gputests.log
rm -f gpurun_out/${R}_pmc.json
PMC_KEY=cfg2 PMC_JSON=$PWD/gpurun_out/${R}_pmc.json timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_${R}_cfg2 --workload cfg2 > gpurun_out/prof_${R}_cfg2.log 2>&1
PMC_KEY=cfg1 PMC_JSON=$PWD/gpurun_out/${R}_pmc.json timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_${R}_cfg1 --workload cfg1 > gpurun_out/prof_${R}_cfg1.log 2>&1
tail -30 gpurun_out/prof_${R}_cfg2/summary.txt; tail -30 gpurun_out/prof_${R}_cfg1/summary.txt
else
[ -f gpurun_out/${R}_pmc.json ] && cp gpurun_out/${R}_pmc.json profiles/${R}_pmc.json
timeout -k 10 400 python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err; echo bench rc=$?
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_steps20.json 2> gpurun_out/${R}_bench_steps20.err; echo bench rc=$?
timeout -k 10 400 python bench.py --workload cfg3 --steps 30 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_cfg3.json 2> /dev/null
timeout -k 10 600 python bench.py --workload cfg4 --steps 5 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_cfg4.json 2> /dev/null
timeout -k 10 300 python tools/steady_probe.py > gpurun_out/${R}_steady_probe.txt 2>&1
timeout -k 10 300 python tools/mixed_probe.py 65536 > gpurun_out/${R}_mixed_probe.txt 2>&1
timeout -k 10 600 python tools/track_probe.py 65536 +distinct +cfg3 +cfg4 +unsorted > gpurun_out/${R}_track_probe.txt 2>&1
bash tools/live_round3.sh > /dev/null 2>&1
cat gpurun_out/${R}_steady_probe.txt gpurun_out/${R}_mixed_probe.txt gpurun_out/${R}_track_probe.txt; grep -v "live. 1 handles" gpurun_out/${R}_live_bench.txt
python - <<'PY'
import json
for f in ("r3_bench","r3_bench_steps20","r3_bench_cfg3","r3_bench_cfg4"):
    d=json.loads([l for l in open("gpurun_out/%s.json"%f) if l.startswith("{")][-1])
    print(f, "%.4g samples/s"%d["value"], "%.3f ms"%d["ms_per_step"], "frac %.4f"%d["roofline"]["frac"], d["roofline"].get("valu",{}).get("frac"), {k:(round(d[k]["kernel_ms"],3)) for k in ("mode_fast","tracks_off","rotated_frame_lists","jittered_durations","unsorted","cfg1","cfg1_recipe_at_batch_65536") if k in d})
PY
fi

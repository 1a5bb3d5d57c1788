#!/bin/bash
# What produced the round's numbers, in one GPU call (`gpurun -- 'bash tools/round_end.sh'`): the GPU test suite, the rocprofv3
# profiles of the two single-GPU configurations (which also write the PMC file bench.py reads), the bench lines, the probes.
# Everything lands under gpurun_out/; the summaries are then copied into profiles/ (see profiles/README.md).
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputests9.log 2>&1; tail -3 gpurun_out/r2_gputests9.log
rm -f gpurun_out/r2_pmc.json
PMC_KEY=cfg2 PMC_JSON=$PWD/gpurun_out/r2_pmc.json timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_r2_cfg2 --workload cfg2 > gpurun_out/prof_r2_cfg2.log 2>&1
PMC_KEY=cfg1 PMC_JSON=$PWD/gpurun_out/r2_pmc.json timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_r2_cfg1 --workload cfg1 > gpurun_out/prof_r2_cfg1.log 2>&1
cp gpurun_out/r2_pmc.json profiles/r2_pmc.json
timeout -k 10 400 python bench.py > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err; echo bench rc=$?
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench2_driver.json 2> gpurun_out/r2_bench2_driver.err; echo bench rc=$?
timeout -k 10 400 python bench.py --workload cfg3 --steps 30 --no-extras --no-cpu-baseline > gpurun_out/r2_bench_cfg3.json 2> /dev/null
timeout -k 10 600 python bench.py --workload cfg4 --steps 5 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r2_bench_cfg4.json 2> /dev/null
timeout -k 10 300 python tools/steady_probe.py > gpurun_out/r2_steady2.txt 2>&1
timeout -k 10 300 python tools/mixed_probe.py 65536 > gpurun_out/r2_mixed2.txt 2>&1
timeout -k 10 600 python tools/track_probe.py 65536 +distinct +cfg3 +cfg4 +unsorted > gpurun_out/r2_track.txt 2>&1
for n in 64 1024 8192; do timeout -k 10 300 python tools/live_bench.py $n; done > gpurun_out/r2_live3.txt 2>&1
cat gpurun_out/r2_steady2.txt gpurun_out/r2_mixed2.txt gpurun_out/r2_track.txt gpurun_out/r2_live3.txt
python - <<'PY'
import json
for f in ("r2_bench2","r2_bench2_driver","r2_bench_cfg3","r2_bench_cfg4"):
    d=json.loads([l for l in open("gpurun_out/%s.json"%f) if l.startswith("{")][-1])
    print(f, "%.4g samples/s"%d["value"], "%.3f ms"%d["ms_per_step"], "frac %.4f"%d["roofline"]["frac"], d["roofline"].get("valu",{}).get("frac"), {k:(round(d[k]["kernel_ms"],3)) for k in ("mode_fast","tracks_off","rotated_frame_lists","cfg1","cfg1_recipe_at_batch_65536") if k in d})
PY

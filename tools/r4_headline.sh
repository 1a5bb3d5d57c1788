#!/bin/bash
# Round 4, VERDICT item 2: what binds the headline (cfg2 on the tracked flat stages)?  Kernel-trace over >= 20 launches and the SQ counter
# passes in BOTH arithmetic modes, then the per-stage stamps of both.   gpurun -- 'bash tools/r4_headline.sh'
R=r4
mkdir -p gpurun_out
rm -f gpurun_out/${R}_pmc.json
PMC_KEY=cfg2 PMC_JSON=$PWD/gpurun_out/${R}_pmc.json timeout -k 10 500 bash tools/profile.sh gpurun_out/prof_${R}_cfg2 --workload cfg2 > gpurun_out/prof_${R}_cfg2.log 2>&1
timeout -k 10 500 bash tools/profile.sh gpurun_out/prof_${R}_cfg2_fast --workload cfg2 --mode 1 > gpurun_out/prof_${R}_cfg2_fast.log 2>&1
V=nvspeechplayer_amd/lib/variants/libspeechPlayer_dst.so
{ for m in 0 1; do SPEECHPLAYER_LIB=$V timeout -k 10 200 python tools/stamps.py cfg2 65536 $m -1; done; } > gpurun_out/${R}_stage_balance.txt 2>&1
tail -40 gpurun_out/prof_${R}_cfg2/summary.txt; tail -40 gpurun_out/prof_${R}_cfg2_fast/summary.txt; cat gpurun_out/${R}_stage_balance.txt

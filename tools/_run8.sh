mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputests5.log 2>&1; tail -4 gpurun_out/r2_gputests5.log
rm -f gpurun_out/r2_pmc.json
PMC_KEY=cfg2 PMC_JSON=$PWD/gpurun_out/r2_pmc.json timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_r2_cfg2 --workload cfg2 > gpurun_out/prof_r2_cfg2.log 2>&1
PMC_KEY=cfg1 PMC_JSON=$PWD/gpurun_out/r2_pmc.json timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_r2_cfg1 --workload cfg1 > gpurun_out/prof_r2_cfg1.log 2>&1
tail -50 gpurun_out/prof_r2_cfg2.log
cat gpurun_out/r2_pmc.json | head -60

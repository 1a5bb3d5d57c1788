mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputests6.log 2>&1; tail -4 gpurun_out/r2_gputests6.log
rm -f gpurun_out/r2_pmc.json
PMC_KEY=cfg2 PMC_JSON=$PWD/gpurun_out/r2_pmc.json timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_r2_cfg2 --workload cfg2 > gpurun_out/prof_r2_cfg2.log 2>&1
PMC_KEY=cfg1 PMC_JSON=$PWD/gpurun_out/r2_pmc.json timeout -k 10 600 bash tools/profile.sh gpurun_out/prof_r2_cfg1 --workload cfg1 > gpurun_out/prof_r2_cfg1.log 2>&1
cp gpurun_out/r2_pmc.json profiles/r2_pmc.json
timeout -k 10 400 python bench.py > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err; echo bench rc=$?
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench2_driver.json 2> gpurun_out/r2_bench2_driver.err; echo bench rc=$?
timeout -k 10 300 python tools/steady_probe.py > gpurun_out/r2_steady2.txt 2>&1
timeout -k 10 300 python tools/mixed_probe.py 65536 > gpurun_out/r2_mixed2.txt 2>&1
for n in 64 1024 8192; do timeout -k 10 300 python tools/live_bench.py $n; done > gpurun_out/r2_live3.txt 2>&1
cat gpurun_out/r2_steady2.txt gpurun_out/r2_mixed2.txt gpurun_out/r2_live3.txt
cut -c1-1500 gpurun_out/r2_bench2.json

mkdir -p gpurun_out
timeout -k 10 500 python tools/ab_probe.py run > gpurun_out/r2_ab1.txt 2>&1
cat gpurun_out/r2_ab1.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random or all_scenarios or nan or golden or cfg3_and_cfg4_recipes or streaming" > gpurun_out/r2_gputests2.log 2>&1; tail -5 gpurun_out/r2_gputests2.log

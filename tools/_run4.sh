mkdir -p gpurun_out
timeout -k 10 500 python tools/ab_probe.py run > gpurun_out/r2_ab2.txt 2>&1
cat gpurun_out/r2_ab2.txt

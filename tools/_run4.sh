mkdir -p gpurun_out
timeout -k 10 800 python tools/ab_probe.py run +cfg2 +rot +cfg2_16k +cfg4 +cfg3 > gpurun_out/r2_ab2.txt 2>&1
cat gpurun_out/r2_ab2.txt

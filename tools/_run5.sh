mkdir -p gpurun_out
timeout -k 10 300 python tools/steady_probe.py > gpurun_out/r2_steady1.txt 2>&1
cat gpurun_out/r2_steady1.txt

mkdir -p gpurun_out
for n in 64 1024 8192; do timeout -k 10 300 python tools/live_bench.py $n; done > gpurun_out/r2_live2.txt 2>&1
SPEECHPLAYER_LIVE_LAYOUT=0 timeout -k 10 300 python tools/live_bench.py 8192 >> gpurun_out/r2_live2.txt 2>&1
cat gpurun_out/r2_live2.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "live or streaming" > gpurun_out/r2_gputests4.log 2>&1; tail -3 gpurun_out/r2_gputests4.log

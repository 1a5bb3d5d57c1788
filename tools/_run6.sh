mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "live or streaming or threads or c_client or prototypeless or smoke" > gpurun_out/r2_gputests3.log 2>&1; tail -30 gpurun_out/r2_gputests3.log
for n in 1024 8192; do timeout -k 10 300 python tools/live_bench.py $n; SPEECHPLAYER_LIVE_LAYOUT=0 timeout -k 10 300 python tools/live_bench.py $n; done > gpurun_out/r2_live1.txt 2>&1
cat gpurun_out/r2_live1.txt

set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputests1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_gputests1.log
tail -5 gpurun_out/r2_gputests1.log
timeout -k 10 300 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err; echo "bench rc=$?"
SPEECHPLAYER_LIB=$PWD/nvspeechplayer_amd/lib/libspeechPlayer_stamps.so timeout -k 10 200 python tools/stamps.py cfg2 16384 > gpurun_out/r2_stamps1.txt 2>&1
SPEECHPLAYER_LIB=$PWD/nvspeechplayer_amd/lib/libspeechPlayer_stamps.so timeout -k 10 200 python tools/stamps.py cfg2 65536 >> gpurun_out/r2_stamps1.txt 2>&1
SPEECHPLAYER_LIB=$PWD/nvspeechplayer_amd/lib/libspeechPlayer_stamps.so timeout -k 10 200 python tools/stamps.py rotated 16384 >> gpurun_out/r2_stamps1.txt 2>&1
timeout -k 10 300 python tools/mixed_probe.py 65536 > gpurun_out/r2_mixed1.txt 2>&1
cat gpurun_out/r2_stamps1.txt gpurun_out/r2_mixed1.txt

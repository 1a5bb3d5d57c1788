#!/bin/bash
# Live handles on the GPU box: tests of the live path, then tools/live_bench.py at 64 / 1024 / 8192 handles with the host-side breakdown.
mkdir -p gpurun_out
{
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "live or stream or native or error or purge" 2>&1 | tail -5
for n in ${LIVE_SIZES:-64 1024 8192 16384 65536}; do
  SPEECHPLAYER_LIVE_TRACE=1 timeout -k 10 300 python tools/live_bench.py $n 2>&1 | grep -v "x 64 in" | awk '/speechPlayer\/live/ { if (++k % 3 == 0) print; next } { print }'
done
} > gpurun_out/${R:-r3}_live_bench.txt 2>&1
cat gpurun_out/${R:-r3}_live_bench.txt

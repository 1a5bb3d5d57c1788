#!/bin/bash
mkdir -p gpurun_out
V=nvspeechplayer_amd/lib/variants
{
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for n in base v2g; do echo $n; SPEECHPLAYER_LIB=$V/libspeechPlayer_$n.so timeout -k 10 200 python tools/steady_probe.py; done
} > gpurun_out/r3i.txt 2>&1
cat gpurun_out/r3i.txt

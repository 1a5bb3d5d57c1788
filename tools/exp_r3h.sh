#!/bin/bash
mkdir -p gpurun_out
V=nvspeechplayer_amd/lib/variants
{
timeout -k 10 400 python tools/ab_probe.py run x4 x8 +cfg2 +jit
SPEECHPLAYER_LIB=$V/libspeechPlayer_x4_st.so timeout -k 10 200 python tools/stamps.py jittered 65536 0 -1
SPEECHPLAYER_LIB=$V/libspeechPlayer_x8_st.so timeout -k 10 200 python tools/stamps.py cfg2 65536 0 -1
} > gpurun_out/r3h.txt 2>&1
cat gpurun_out/r3h.txt

#!/bin/bash
mkdir -p gpurun_out
V=nvspeechplayer_amd/lib/variants
{
timeout -k 10 300 python tools/ab_probe.py run v2a_e1 +cfg2 +jit
for n in 16384 65536; do
  SPEECHPLAYER_LIB=$V/libspeechPlayer_v2a_st.so timeout -k 10 200 python tools/stamps.py jittered $n 0 -1
done
SPEECHPLAYER_LIB=$V/libspeechPlayer_v2a_st.so timeout -k 10 200 python tools/stamps.py cfg2 65536 0 -1
} > gpurun_out/r3c_exp.txt 2>&1
cat gpurun_out/r3c_exp.txt

#!/bin/bash
mkdir -p gpurun_out
V=nvspeechplayer_amd/lib/variants
N=${1:-v2e}
{
timeout -k 10 400 python tools/ab_probe.py run base $N +cfg2 +rot +jit +cfg4 +cfg3
for w in "jittered 65536" "cfg2 65536"; do
  SPEECHPLAYER_LIB=$V/libspeechPlayer_${N}_st.so timeout -k 10 200 python tools/stamps.py $w 0 -1
done
} > gpurun_out/r3g_$N.txt 2>&1
cat gpurun_out/r3g_$N.txt

#!/bin/bash
mkdir -p gpurun_out
V=nvspeechplayer_amd/lib/variants
{
timeout -k 10 400 python tools/ab_probe.py run base v2b v2b_e1 +cfg2 +rot +jit +cfg4
for w in "jittered 65536" "cfg2 65536"; do
  SPEECHPLAYER_LIB=$V/libspeechPlayer_v2b_st.so timeout -k 10 200 python tools/stamps.py $w 0 -1
done
} > gpurun_out/r3d_exp.txt 2>&1
cat gpurun_out/r3d_exp.txt

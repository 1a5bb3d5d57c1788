#!/bin/bash
# A/B run of library variants on the GPU box (tools/ab_probe.py build NAME=-DFLAGS ... first): timings + digests, then per-stage stamps of NAME_st.
# usage: bash tools/ab_round3.sh NAME [workloads ...]      (writes gpurun_out/r3_ab_NAME.txt)
mkdir -p gpurun_out
V=nvspeechplayer_amd/lib/variants
N=${1:-v2}; shift
W=${*:-+cfg2 +rot +jit +cfg4 +cfg3}
{
timeout -k 10 500 python tools/ab_probe.py run base $N $W
if [ -f $V/libspeechPlayer_${N}_st.so ]; then
  for w in "jittered 65536" "cfg2 65536"; do
    SPEECHPLAYER_LIB=$V/libspeechPlayer_${N}_st.so timeout -k 10 200 python tools/stamps.py $w 0 -1
  done
fi
} > gpurun_out/r3_ab_$N.txt 2>&1
cat gpurun_out/r3_ab_$N.txt

#!/bin/bash
# round 3, first measurement: where does the sample-by-sample path of the flat stages spend its time?
mkdir -p gpurun_out
V=nvspeechplayer_amd/lib/variants
timeout -k 10 500 python tools/ab_probe.py run base exp1 exp3 +cfg2 +rot +jit > gpurun_out/r3a_exp.txt 2>&1
for near in 1024 65536; do
  echo "near $near" >> gpurun_out/r3a_exp.txt
  SPEECHPLAYER_EXP_NEAR=$near SPEECHPLAYER_LIB=$V/libspeechPlayer_base.so timeout -k 10 200 python tools/ab_probe.py one jit cfg2 >> gpurun_out/r3a_exp.txt 2>&1
  SPEECHPLAYER_EXP_NEAR=$near SPEECHPLAYER_LIB=$V/libspeechPlayer_exp3.so timeout -k 10 200 python tools/ab_probe.py one jit >> gpurun_out/r3a_exp.txt 2>&1
done
cat gpurun_out/r3a_exp.txt

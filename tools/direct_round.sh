#!/bin/bash
# One GPU round for the direct stages: correctness (legacy = direct = tracked, oracle), timings, per-stage stamps.
#   bash tools/direct_round.sh TAG [time args]        -> gpurun_out/r4_direct_TAG.txt
mkdir -p gpurun_out
T=${1:-x}; shift
{
timeout -k 10 400 python tools/direct_probe.py check || exit 1
timeout -k 10 400 python tools/direct_probe.py time "$@"
V=nvspeechplayer_amd/lib/variants/libspeechPlayer_dst.so
if [ -f $V ]; then
  for m in 0 1; do SPEECHPLAYER_LIB=$V timeout -k 10 200 python tools/direct_stamps.py all_different 65536 $m; done
fi
} > gpurun_out/r4_direct_$T.txt 2>&1
cat gpurun_out/r4_direct_$T.txt

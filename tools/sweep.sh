#!/bin/bash
# A/B builds of the engine (nvspeechplayer_amd/lib/variants/*.so) on the GPU box: bench.py per variant.
# usage: tools/sweep.sh "<bench args>" [variant.so ...]
ARGS="$1"; shift
for so in "$@"; do
  SPEECHPLAYER_LIB=$(realpath $so) timeout 300 python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s %s mode %d layout %d: kernel_ms %.3f value %.4g frac %.4f vgprs %d waves %d' % ('$(basename $so)', d['config']['workload'][:4], d['config']['mode'], d['config']['layout'], r['kernel_ms'], d['value'], r['frac'], r['vgprs'], r['wavefronts']))"
done
